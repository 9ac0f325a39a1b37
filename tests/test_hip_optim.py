"""carca_replication_amd.optim.Adam (one launch per step, carca_adam_step) against torch.optim.Adam, the optimizer
scripts/training.py:174 builds.  Same formula, fp32: trajectories agree to a few ulp per step (torch's fused / foreach
kernels contract multiply-adds differently), bound written below."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    a = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_matches_torch(wd):
    from carca_replication_amd.optim import Adam

    shapes = [(1,), (3, 5), (1024,), (1025,), (90, 540), (7, 1, 13), (300000,)]
    a, b = _pair(shapes, 3)
    ours = Adam(a, lr=1e-3, betas=(0.9, 0.98), weight_decay=wd)
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.98), weight_decay=wd)
    g = torch.Generator().manual_seed(4)
    for step in range(6):
        for p, q in zip(a, b):
            grad = (torch.randn(p.shape, generator=g) * (0.1 + step)).cuda()
            p.grad, q.grad = grad.clone(), grad.clone()
        ours.step()
        ref.step()
        for p, q in zip(a, b):
            # every step moves a weight by <= lr; the two implementations differ by rounding only
            assert float((p - q).abs().max()) <= 2e-6 * (step + 1), (step, tuple(p.shape))
    for p, q in zip(a, b):
        so, sr = ours.state[p], ref.state[q]
        assert float(so["step"]) == float(sr["step"]) == 6
        # gradients reach 5: rounding of the cancelling terms is ~1e-6 absolute on m, relative on v
        assert torch.allclose(so["exp_avg"], sr["exp_avg"], rtol=1e-5, atol=2e-6)
        assert torch.allclose(so["exp_avg_sq"], sr["exp_avg_sq"], rtol=1e-5, atol=1e-7)


def test_adam_state_dict_interchangeable_with_torch():
    from carca_replication_amd.optim import Adam

    a, b = _pair([(17, 9), (130,)], 5)
    ours = Adam(a, lr=2e-3, betas=(0.9, 0.98))
    for p in a:
        p.grad = torch.ones_like(p)
    ours.step()
    ours.step()
    ref = torch.optim.Adam(b, lr=2e-3, betas=(0.9, 0.98))
    ref.load_state_dict(copy.deepcopy(ours.state_dict()))          # ours -> torch (deep copy: load shares tensors)
    for p, q in zip(a, b):
        q.data.copy_(p.data)
        p.grad = torch.full_like(p, 0.5)
        q.grad = torch.full_like(q, 0.5)
    ours.step()
    ref.step()
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) <= 1e-6
    a2, _ = _pair([(17, 9), (130,)], 5)
    back = Adam(a2, lr=2e-3, betas=(0.9, 0.98))
    back.load_state_dict(copy.deepcopy(ref.state_dict()))          # torch -> ours: step counts carry over
    for p2, q in zip(a2, b):
        p2.data.copy_(q.data)
        p2.grad = torch.full_like(p2, -0.25)
        q.grad = torch.full_like(q, -0.25)
    back.step()
    ref.step()
    for p2, q in zip(a2, b):
        assert float(back.state[p2]["step"]) == 4
        assert float((p2 - q).abs().max()) <= 1e-6


def test_state_dict_reports_the_live_step_count_every_time():
    """A checkpoint taken mid-run must not cut the optimizer's state off its own step counter: state_dict() after 2, 4
    and 5 steps reports 2, 4 and 5 (the first version cloned the step tensor INTO the live state, and every later
    checkpoint carried the step count of the first one: a resumed run took its next step with a stale bias correction)."""
    from carca_replication_amd.optim import Adam

    a, _ = _pair([(17, 9), (130,)], 6)
    opt = Adam(a, lr=1e-3, betas=(0.9, 0.98))
    seen = []
    for n_steps in (2, 2, 1):
        for _ in range(n_steps):
            for p in a:
                p.grad = torch.ones_like(p)
            opt.step()
        sd = opt.state_dict()
        seen.append([float(st["step"]) for st in sd["state"].values()])
        assert all(float(opt.state[p]["step"]) == seen[-1][0] for p in a)
    assert seen == [[2.0, 2.0], [4.0, 4.0], [5.0, 5.0]]
    # the snapshots are copies: stepping on does not move them
    for p in a:
        p.grad = torch.ones_like(p)
    opt.step()
    assert [float(st["step"]) for st in sd["state"].values()] == [5.0, 5.0]


def test_adam_skips_parameters_without_gradient_and_rejects_bad_input():
    from carca_replication_amd.optim import Adam
    from carca_replication_amd.ops import CarcaHipError

    a, _ = _pair([(8,), (8,)], 6)
    before = a[1].detach().clone()
    opt = Adam(a, lr=1e-2)
    a[0].grad = torch.ones_like(a[0])
    opt.step()
    assert torch.equal(a[1], before) and not opt.state[a[1]]
    with pytest.raises(ValueError):
        Adam(a, lr=-1.0)
    cpu = [torch.nn.Parameter(torch.zeros(4))]
    cpu[0].grad = torch.ones(4)
    with pytest.raises(CarcaHipError):
        Adam(cpu).step()


def test_adam_step_invalidates_packed_weight_caches():
    """The kernel updates parameters through raw pointers; the modules repack their weights when a parameter's
    version changes, so the optimizer has to bump it.  After a step the forward must see the NEW attention weights:
    same outputs as a fresh model loaded with the updated state_dict."""
    import copy

    from carca_replication_amd.optim import Adam
    from tests.golden_util import load
    from tests.model_util import model_from_fixture

    fx = load("g2_d90h3")
    model = model_from_fixture(fx).cuda().train()
    g = lambda k: fx.ins[k].cuda()  # noqa: E731
    L = fx.ins["p_x"].shape[1]
    pos = tuple(g(k)[:, :L].contiguous() for k in ("o_x", "o_a", "o_c"))
    neg = tuple(g(k)[:, L:].contiguous() for k in ("o_x", "o_a", "o_c"))
    profile = (g("p_x"), g("p_a"), g("p_c"))
    opt = Adam(model.parameters(), lr=5e-2, betas=(0.9, 0.98))  # big steps: stale weights would be obvious
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        y = model(profile=profile, targets=[pos, neg])
        (y * y).sum().backward()
        opt.step()
    with torch.no_grad():
        model.eval()
        y_now = model(profile=profile, targets=[pos, neg])
        fresh = model_from_fixture(fx).cuda().eval()
        fresh.load_state_dict(copy.deepcopy(model.state_dict()))
        y_fresh = fresh(profile=profile, targets=[pos, neg])
    assert torch.equal(y_now, y_fresh)
    # and in train mode (the autograd path packs separately)
    model.train()
    fresh.train()
    y_t = model(profile=profile, targets=[pos, neg])
    y_ft = fresh(profile=profile, targets=[pos, neg])
    assert torch.equal(y_t, y_ft)


def test_fused_torch_adam_does_not_leave_stale_packed_weights():
    """torch.optim.Adam(fused=True) does not bump parameter versions; the packed-weight caches must not rely on them."""
    import copy

    from tests.golden_util import load
    from tests.model_util import model_from_fixture

    fx = load("g2_d90h3")
    model = model_from_fixture(fx).cuda().train()
    g = lambda k: fx.ins[k].cuda()  # noqa: E731
    L = fx.ins["p_x"].shape[1]
    pos = tuple(g(k)[:, :L].contiguous() for k in ("o_x", "o_a", "o_c"))
    neg = tuple(g(k)[:, L:].contiguous() for k in ("o_x", "o_a", "o_c"))
    profile = (g("p_x"), g("p_a"), g("p_c"))
    opt = torch.optim.Adam(model.parameters(), lr=5e-2, betas=(0.9, 0.98), fused=True)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        y = model(profile=profile, targets=[pos, neg])
        (y * y).sum().backward()
        opt.step()
    fresh = model_from_fixture(fx).cuda()
    fresh.load_state_dict(copy.deepcopy(model.state_dict()))
    for mode in (True, False):  # the training path packs per step; the inference path repacks once after training
        model.train(mode)
        fresh.train(mode)
        with torch.set_grad_enabled(mode):
            assert torch.equal(model(profile=profile, targets=[pos, neg]), fresh(profile=profile, targets=[pos, neg]))
    model.fold_embedding(True)
    fresh.fold_embedding(True)
    with torch.no_grad():
        assert torch.equal(model(profile=profile, targets=[pos, neg]), fresh(profile=profile, targets=[pos, neg]))


def test_sharded_step_reduces_the_flat_buffer_in_place_and_equals_the_plain_step():
    """The N > 1 train step on the HIP path, executed with a 1-rank process group (dist.FORCE_COLLECTIVES): global mask
    count -> forward / backward normalised by it -> the backward's flat gradient buffer all-reduced IN PLACE, its early
    range on a side stream gated by the event carca_embed_bwd records before its last launch -> one-launch Adam.
    A 1-rank sum is the identity, so parameters must equal the unsharded step's bit for bit -- while every pointer, view
    and stream hand-off of the sharded path has run.  Also through the hipGraph (GraphedTrainStep(sharded=True))."""
    import copy
    import os
    import socket

    import torch.distributed as dist

    from carca_replication_amd import dist as cdist
    from carca_replication_amd import engine
    from carca_replication_amd.optim import Adam
    from tests.test_hip_graph import _setup

    fresh, batch = _setup(0.0)
    other = tuple(t.roll(1, 0) for t in batch)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cdist.FORCE_COLLECTIVES = True
        model_p, opt_p = fresh()
        model_s = copy.deepcopy(model_p)
        opt_s = Adam(model_s.parameters(), lr=1e-3, betas=(0.9, 0.98))
        model_g = copy.deepcopy(model_p)
        opt_g = Adam(model_g.parameters(), lr=1e-3, betas=(0.9, 0.98))
        step_g = engine.GraphedTrainStep(model_g, opt_g, batch, sharded=True)
        for bt in (batch, other, batch):
            cdist.last_reduce = {}
            ls = engine.train_step(model_s, opt_s, bt, sharded=True)
            assert step_g.ev_early is not None, getattr(step_g, 'ev_early_note', '')  # (this runtime can add the external event-record node to a capture)
            assert cdist.last_reduce.get("path") == "flat-inplace" and cdist.last_reduce["early_overlapped"], cdist.last_reduce
            assert cdist.last_reduce["launches"] == 2  # early range + late range (12 MB model: one chunk each)
            cdist.last_reduce = {}
            lg = step_g(bt)
            # (the graph records the early-gradients event as an external node: the early range goes out under its last kernel)
            assert step_g.ev_early is not None, getattr(step_g, 'ev_early_note', '')  # (this runtime can add the external event-record node to a capture)
            assert cdist.last_reduce.get("path") == "flat-inplace" and cdist.last_reduce["early_overlapped"], cdist.last_reduce
            cdist.FORCE_COLLECTIVES = False
            lp = engine.train_step(model_p, opt_p, bt)
            cdist.FORCE_COLLECTIVES = True
            assert float(ls) == pytest.approx(float(lp), rel=1e-5) and float(lg) == pytest.approx(float(lp), rel=1e-5)
        info = cdist.flat_layout(model_s, list(model_s.parameters()))
        fw = model_s.embeds.feats_embed.weight
        assert info["late"][1] - info["late"][0] >= fw.numel() and fw.grad.storage_offset() >= info["late"][0]
        for (n, a), (_, b), (_, c) in zip(model_s.named_parameters(), model_p.named_parameters(), model_g.named_parameters()):
            if n.endswith("WK.bias"):  # true gradient 0: Adam turns round-off (fp32 atomics) into +-lr steps
                continue
            assert torch.allclose(a, b, rtol=1e-3, atol=2e-5), n
            assert torch.allclose(c, b, rtol=1e-3, atol=2e-5), n
    finally:
        cdist.FORCE_COLLECTIVES = False
        dist.destroy_process_group()


def test_touched_row_adam_is_bit_exact():
    """DESIGN section 5's claim, deterministically: IDENTICAL gradient tensors into optim.Adam + mark_rows (rows never
    touched are skipped) and into the same optimizer run densely over every row give bit-identical parameters, exp_avg
    and exp_avg_sq after three steps -- a row that was never touched has g = m = v = 0 and Adam's update of it is exactly
    0.  Against torch.optim.Adam over the dense gradient: untouched rows bitwise the initial ones with an all-zero state
    in both, touched rows within the rounding bound of test_adam_matches_torch (torch contracts multiply-adds differently)."""
    from carca_replication_amd.optim import Adam

    n_rows, dim, steps = 100_000, 128, 3
    gen = torch.Generator().manual_seed(11)
    w0 = torch.randn(n_rows, dim, generator=gen).cuda()
    small0 = torch.randn(33, 7, generator=gen).cuda()
    tabs = [torch.nn.Parameter(w0.clone()) for _ in range(3)]
    smalls = [torch.nn.Parameter(small0.clone()) for _ in range(3)]
    masked = Adam([tabs[0], smalls[0]], lr=1e-3, betas=(0.9, 0.98))
    dense = Adam([tabs[1], smalls[1]], lr=1e-3, betas=(0.9, 0.98))
    ref = torch.optim.Adam([tabs[2], smalls[2]], lr=1e-3, betas=(0.9, 0.98))
    touched = torch.zeros(n_rows, dtype=torch.bool, device="cuda")
    for step in range(steps):
        ids = torch.randint(1, n_rows, (128, 150), generator=gen).to(torch.int32).cuda()
        if step == 1:
            ids = ids[:, :40]  # (step 2 leaves most of step 1's rows without a gradient: their momentum still moves them)
        grad = torch.zeros(n_rows, dim, device="cuda")
        grad.index_add_(0, ids.reshape(-1).long(), torch.randn(ids.numel(), dim, generator=gen).cuda())
        gs = torch.randn(33, 7, generator=gen).cuda()
        touched[ids.reshape(-1).long()] = True
        for t, s_ in zip(tabs, smalls):
            t.grad, s_.grad = grad.clone(), gs.clone()
        assert masked.mark_rows(tabs[0], ids)
        masked.step()
        dense.step()
        ref.step()
    sm, sd, sr = masked.state[tabs[0]], dense.state[tabs[1]], ref.state[tabs[2]]
    assert "row_touched" in sm and "row_touched" not in sd
    assert torch.equal(sm["row_touched"].bool(), touched) and 0 < int(touched.sum()) < n_rows // 2
    assert torch.equal(tabs[0], tabs[1]) and torch.equal(smalls[0], smalls[1])
    assert torch.equal(sm["exp_avg"], sd["exp_avg"]) and torch.equal(sm["exp_avg_sq"], sd["exp_avg_sq"])
    for w, st in ((tabs[0], sm), (tabs[1], sd), (tabs[2], sr)):
        assert torch.equal(w[~touched], w0[~touched])
        assert int(st["exp_avg"][~touched].count_nonzero()) == 0 and int(st["exp_avg_sq"][~touched].count_nonzero()) == 0
    assert float((tabs[0] - tabs[2]).abs().max()) <= 2e-6 * steps
    # a parameter of no group is refused instead of raising out of the step (frozen table / another optimizer's)
    assert masked.mark_rows(torch.nn.Parameter(torch.zeros(8, 4, device="cuda")), ids) is False
