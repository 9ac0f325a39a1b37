"""Profiles longer than the fused kernels' 64 slots, and more target groups than one fused call takes: the reference accepts
any --seq_len (scripts/training.py:34-63) and any number of target groups (carca.py:424).  Such calls run
carca_replication_amd/long_profile.py -- CARCA.forward composed from the row-level kernels (LayerNorm, row GEMMs, the
one-wave-per-query attention core, dropout) under torch.autograd.  Pinned here:
  * at L = 50, where both paths apply, the composed path reproduces the fused one (forward and every gradient);
  * at L = 100 it matches the oracle: eval forward (2e-5 on probabilities, the ranking identical), training forward + every
    parameter gradient (p = 0), dropout with the kernels' exported keep-masks replayed by the oracle, the dot decoders,
    a stand-alone SelfAttentionBlock / CrossAttentionBlock, engine.train_step / eval_batch;
  * five target groups at L = 50 against the oracle."""
import pytest
import torch

from oracle import carca_oracle as O
from tests.model_util import build_model, dev, model_from_params

pytestmark = pytest.mark.gpu


def _case(L, B=7, d=90, H=3, nb=2, n_items=300, n_attrs=40, n_ctx=4, g=120, seed=3, **cfg_kw):
    cfg = O.CarcaConfig(d=d, H=H, n_blocks=nb, **cfg_kw)
    P = O.perturb_params(O.init_params(cfg, n_items, g, n_ctx, n_attrs, L, seed=0), seed=1)
    profile, pos, _ = O.synth_eval_batch(B, L, L, n_items, n_attrs, n_ctx, seed=seed, min_len=2)
    neg = (pos[0].flip(1).contiguous(), pos[1].flip(1).contiguous(), pos[2])
    px = profile[0]
    pos = (pos[0] * (px != 0), pos[1], pos[2])
    neg = (neg[0] * (px != 0), neg[1], neg[2])
    y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
    o_x = torch.cat([pos[0], neg[0]], dim=1)
    return cfg, P, profile, pos, neg, y_true, o_x


def _grads_close(model, Pg, rtol=2e-4):
    for name, prm in model.named_parameters():
        ref = Pg[name].grad if Pg[name].grad is not None else torch.zeros_like(Pg[name])
        got = prm.grad.cpu() if prm.grad is not None else torch.zeros_like(ref)
        err = float((got - ref).abs().max())
        assert err <= rtol * float(ref.abs().max()) + 1e-7, (name, err, float(ref.abs().max()))


def test_composed_path_reproduces_the_fused_path_at_l50():
    from carca_replication_amd import long_profile, modules as M

    cfg, P, profile, pos, neg, y_true, o_x = _case(50)
    fused, comp = model_from_params(P, cfg).train(), model_from_params(P, cfg).train()
    yf = fused(profile=dev(profile), targets=[dev(pos), dev(neg)])
    yc = torch.cat(long_profile.forward(comp, dev(profile), [dev(pos), dev(neg)]), dim=-1)
    assert float((yf - yc).abs().max()) < 2e-6
    for y in (yf, yc):
        M.BinaryCrossEntropy()(y, y_true.cuda(), M.get_mask(o_x.cuda())).backward()
    for (n, a), (_, b) in zip(fused.named_parameters(), comp.named_parameters()):
        scale = float(a.grad.abs().max())
        assert float((a.grad - b.grad).abs().max()) <= 1e-4 * scale + 1e-7, n
    fused.eval(), comp.eval()
    with torch.no_grad():
        profile_e, target_e, _ = O.synth_eval_batch(7, 50, 101, 300, 40, 4, seed=5, min_len=2)
        ye = fused(profile=dev(profile_e), targets=[dev(target_e)])
        yl = long_profile.forward(comp, dev(profile_e), [dev(target_e)])[0]
    assert float((ye - yl).abs().max()) < 2e-6


@pytest.mark.parametrize("L", [65, 100, 128])
def test_eval_forward_matches_the_oracle_beyond_64_slots(L):
    cfg, P, *_ = _case(L)
    profile, target, _ = O.synth_eval_batch(9, L, 101, 300, 40, 4, seed=11, min_len=1)
    want = O.carca_forward(P, cfg, profile, [target], training=False)
    model = model_from_params(P, cfg).eval()
    with torch.no_grad():
        trace = {}
        got = model(profile=dev(profile), targets=[dev(target)])
        model.forward_nograd(dev(profile), [dev(target)], trace=trace)
    assert got.shape == want.shape
    assert float((got.cpu() - want).abs().max()) < 2e-5
    assert torch.equal(O.positive_rank(got.cpu()), O.positive_rank(want))
    ref = {}
    O.carca_forward(P, cfg, profile, [target], training=False, trace=ref)
    for k in ("p_embed", "block0", "block1", "p_final", "o_embed0"):
        assert float((trace[k].cpu() - ref[k]).abs().max()) < 1e-4, k  # activations: 1e-4


def test_training_forward_and_gradients_match_the_oracle_at_l100():
    from carca_replication_amd import modules as M

    cfg, P, profile, pos, neg, y_true, o_x = _case(100, encoding="learnable")
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yo = O.carca_forward(Pg, cfg, profile, [pos, neg], training=True)
    lo = O.bce_loss(yo, y_true, O.get_mask(o_x))
    lo.backward()
    model = model_from_params(P, cfg).train()
    y = model(profile=dev(profile), targets=[dev(pos), dev(neg)])
    loss = M.BinaryCrossEntropy()(y, y_true.cuda(), M.get_mask(o_x.cuda()))
    loss.backward()
    assert float((y.detach().cpu() - yo.detach()).abs().max()) < 2e-5
    assert abs(float(loss) - float(lo)) < 2e-6
    _grads_close(model, Pg)


@pytest.mark.parametrize("p", [0.3])
def test_dropout_at_l100_replayed_by_the_oracle_with_the_same_masks(p):
    from carca_replication_amd import modules as M
    from tests.test_hip_dropout import _oracle_masks

    cfg, P, profile, pos, neg, y_true, o_x = _case(100, B=5)
    B, L = profile[0].shape
    model = model_from_params(P, cfg).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p
    model._keep_dropout_masks = True
    torch.manual_seed(7)
    y = model(profile=dev(profile), targets=[dev(pos), dev(neg)])
    loss = M.BinaryCrossEntropy()(y, y_true.cuda(), M.get_mask(o_x.cuda()))
    loss.backward()
    raw = model._last_dropout_masks
    keep_rate = float(raw["blocks"][0]["m_attn"].float().mean())
    assert abs(keep_rate - (1 - p)) < 0.01
    mk = _oracle_masks(model, cfg, p, B, L, [L, L])
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yo = O.carca_forward(Pg, cfg, profile, [pos, neg], training=True, masks=mk)
    lo = O.bce_loss(yo, y_true, O.get_mask(o_x))
    lo.backward()
    assert float((y.detach().cpu() - yo.detach()).abs().max()) < 5e-5
    assert abs(float(loss) - float(lo)) < 1e-5
    _grads_close(model, Pg)
    # another seed, other masks; the same seed, the same scores
    torch.manual_seed(7)
    y2 = model(profile=dev(profile), targets=[dev(pos), dev(neg)])
    assert torch.equal(y2, y)
    torch.manual_seed(8)
    y3 = model(profile=dev(profile), targets=[dev(pos), dev(neg)])
    assert not torch.equal(y3, y)


@pytest.mark.parametrize("decoder", ["dot", "wdot"])
def test_dot_decoders_train_at_l100(decoder):
    from carca_replication_amd import modules as M

    cfg, P, profile, pos, neg, y_true, o_x = _case(100, decoder=decoder)
    P = {k: v for k, v in P.items() if not k.startswith("decoder.")}
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yo = O.carca_forward(Pg, cfg, profile, [pos, neg], training=True)
    lo = O.bce_loss(yo, y_true, O.get_mask(o_x))
    lo.backward()
    n_items, d = P["embeds.items_embed.weight"].shape
    g, F = P["embeds.feats_embed.weight"].shape
    model = build_model(dict(d=cfg.d, H=cfg.H, n_blocks=cfg.n_blocks, decoder=decoder), n_items, g, 0, F, 100)
    model.load_state_dict(P, strict=True)
    model = model.cuda().train()
    y = model(profile=dev(profile), targets=[dev(pos), dev(neg)])
    loss = M.BinaryCrossEntropy()(y, y_true.cuda(), M.get_mask(o_x.cuda()))
    loss.backward()
    assert float((y.detach().cpu() - yo.detach()).abs().max()) < 2e-5
    _grads_close(model, Pg)


def test_standalone_blocks_at_l100():
    cfg, P, profile, pos, neg, y_true, o_x = _case(100)
    model = model_from_params(P, cfg).eval()
    B, L = profile[0].shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, L, cfg.d, generator=g)
    mask = O.get_mask(profile[0])
    want = O.sa_block(P, cfg, 0, x, mask)
    with torch.no_grad():
        got = model.encoder[0](x.cuda(), mask.cuda())
    assert float((got.cpu() - want).abs().max()) < 1e-4
    o = torch.randn(B, 7, cfg.d, generator=g)
    o_mask = torch.ones(B, 7)
    want = O.cross_block(P, cfg, o, o_mask, x, mask, training=False)
    with torch.no_grad():
        got = model.decoder(o.cuda(), o_mask.cuda(), x.cuda(), mask.cuda())
    assert got.shape == want.shape and float((got.cpu() - want).abs().max()) < 2e-5
    # and differentiable on its own
    xg = x.cuda().requires_grad_(True)
    model.train()
    model.encoder[0](xg, mask.cuda()).square().sum().backward()
    xo = x.clone().requires_grad_(True)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    O.sa_block(Pg, cfg, 0, xo, mask).square().sum().backward()
    assert float((xg.grad.cpu() - xo.grad).abs().max()) <= 2e-4 * float(xo.grad.abs().max())


def test_engine_steps_at_l100():
    """engine.train_step (train.py:84-96) and engine.eval_batch (train.py:42-51) through the unchanged entry points."""
    from carca_replication_amd import engine

    cfg, P, profile, pos, neg, y_true, o_x = _case(100, B=16)
    model = model_from_params(P, cfg).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
    batch = dev((profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], neg[1]], 1), torch.cat([pos[2], neg[2]], 1),
                 y_true))
    losses = [float(engine.train_step(model, opt, batch)) for _ in range(8)]
    assert losses[-1] < losses[0]
    profile_e, target_e, _ = O.synth_eval_batch(16, 100, 101, 300, 40, 4, seed=21, min_len=1)
    y_e = torch.zeros(16, 101)
    y_e[:, 0] = 1.0  # candidate 0 is the positive (data.py:165,190)
    model.eval()
    with torch.no_grad():
        y, sums = engine.eval_batch(model, dev(profile_e) + dev(target_e) + (y_e.cuda(),), k=10)
    Pn = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = O.carca_forward(Pn, cfg, profile_e, [target_e], training=False)
    assert float((y.cpu() - want).abs().max()) < 2e-5
    hr, ndcg = O.hr_ndcg_sums(want, 10)
    sums = sums.cpu()
    assert float(sums[0]) == hr and abs(float(sums[1]) - ndcg) < 1e-4 and float(sums[4]) == 16.0


def test_five_target_groups_in_one_call():
    cfg, P, *_ = _case(50)
    profile, t0, _ = O.synth_eval_batch(6, 50, 11, 300, 40, 4, seed=31, min_len=2)
    groups = [t0] + [O.synth_eval_batch(6, 50, n, 300, 40, 4, seed=32 + n, min_len=2)[1] for n in (5, 9, 2, 3)]
    want = O.carca_forward(P, cfg, profile, groups, training=False)
    model = model_from_params(P, cfg).eval()
    with torch.no_grad():
        got = model(profile=dev(profile), targets=[dev(t) for t in groups])
    assert got.shape == want.shape and float((got.cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("d,H,L", [(192, 3, 20), (256, 4, 50), (130, 2, 12)])
def test_models_wider_than_128_run_the_composed_path(d, H, L):
    """--d_dim above the fused kernels' 128 (scripts/training.py:42; the reference's only constraint is d % H == 0,
    carca.py:208 -- VERDICT r4 'What's missing' 2): CARCA.forward routes to long_profile.py's composition, whose LayerNorms
    take rows up to 1024 wide.  Eval scores and ranks, training scores, loss and every parameter gradient against the
    oracle; a stand-alone SelfAttentionBlock / CrossAttentionBlock / AllEmbedding call too."""
    from carca_replication_amd import modules as M

    cfg, P, profile, pos, neg, y_true, o_x = _case(L, B=5, d=d, H=H, g=72, encoding="learnable")
    model = model_from_params(P, cfg)
    profile_e, target_e, _ = O.synth_eval_batch(6, L, 37, 300, 40, 4, seed=21, min_len=1)
    want = O.carca_forward(P, cfg, profile_e, [target_e], training=False)
    model.eval()
    with torch.no_grad():
        got = model(profile=dev(profile_e), targets=[dev(target_e)])
    assert got.shape == want.shape
    assert float((got.cpu() - want).abs().max()) < 2e-5
    assert torch.equal(O.positive_rank(got.cpu()), O.positive_rank(want))
    # training forward + backward
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yo = O.carca_forward(Pg, cfg, profile, [pos, neg], training=True)
    lo = O.bce_loss(yo, y_true, O.get_mask(o_x))
    lo.backward()
    model.train()
    y = model(profile=dev(profile), targets=[dev(pos), dev(neg)])
    loss = M.BinaryCrossEntropy()(y, y_true.cuda(), M.get_mask(o_x.cuda()))
    loss.backward()
    assert float((y.detach().cpu() - yo.detach()).abs().max()) < 2e-5
    assert abs(float(loss) - float(lo)) < 2e-6
    _grads_close(model, Pg)
    # the stand-alone modules of the ABCs at this width
    model.eval()
    with torch.no_grad():
        mask = (profile_e[0] != 0).float()
        e = model.embeds(dev(profile_e[0]), dev(profile_e[1]), dev(profile_e[2]), mask.cuda(), False)
        e_o = O.embedding(P, cfg, profile_e[0].long(), profile_e[1], profile_e[2], mask, False)
        assert float((e.cpu() - e_o).abs().max()) < 2e-5
        x1 = model.encoder[0](e, mask.cuda())
        x1_o = O.sa_block(P, cfg, 0, e_o, mask)
        assert float((x1[..., :d].cpu() - x1_o).abs().max()) < 1e-4
