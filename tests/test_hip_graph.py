"""engine.GraphedTrainStep: forward + backward replayed from a hipGraph, optimizer launched behind it."""
import copy

import pytest
import torch

from oracle import carca_oracle as O
from tests.model_util import build_model

pytestmark = pytest.mark.gpu


def _setup(p_drop):
    from carca_replication_amd.optim import Adam
    from carca_replication_amd.synth import eval_batch

    cfg = O.CarcaConfig(d=90, H=3, n_blocks=2, encoding="learnable")
    n_items, n_attrs, n_ctx, g, L, B = 300, 40, 6, 64, 20, 9
    P = O.perturb_params(O.init_params(cfg, n_items, g, n_ctx, n_attrs, L, seed=0), seed=1)
    profile, pos, _ = eval_batch(B, L, L, n_items, n_attrs, n_ctx, seed=5)
    px = profile[0]
    o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
    o_a = torch.cat([pos[1], pos[1].flip(1)], dim=1)
    o_c = torch.cat([pos[2], pos[2]], dim=1)
    y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
    batch = tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, o_a, o_c, y_true))

    def fresh():
        m = build_model(dict(d=cfg.d, H=cfg.H, n_blocks=cfg.n_blocks, encoding=cfg.encoding), n_items, g, n_ctx, n_attrs, L,
                        p=p_drop)
        m.load_state_dict(copy.deepcopy(P), strict=True)
        m = m.cuda().train()
        return m, Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.98))

    return fresh, batch


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_graphed_step_is_the_eager_step(p_drop, monkeypatch):
    """Replay t of the captured step = eager train_step number t (with dropout: the eager step given seed + t): same
    kernels on the same numbers, so losses and gradients agree to the run-to-run noise of the fp32 atomics that some
    gradients are accumulated with (a wrong mask or a stale weight would show at 1e-2)."""
    from carca_replication_amd import engine, ops

    fresh, batch = _setup(p_drop)
    base = 123456789
    monkeypatch.setattr(ops, "new_dropout_seed", lambda: base)
    model_g, opt_g = fresh()
    step = engine.GraphedTrainStep(model_g, opt_g, batch)
    model_e, opt_e = fresh()
    graphed, eager = [], []
    for t in range(1, 4):
        graphed.append(float(step(batch)))
        monkeypatch.setattr(ops, "new_dropout_seed", lambda t=t: base + t)
        eager.append(float(engine.train_step(model_e, opt_e, batch)))
        if t == 1:  # gradients of the first step (later ones see parameters whose gradient is pure round-off, the key
            # biases, after Adam has turned that noise into +-lr steps)
            for (n, a), (_, b) in zip(model_g.named_parameters(), model_e.named_parameters()):
                assert torch.allclose(a.grad, b.grad, rtol=1e-3, atol=1e-6), n
    assert int(step.replays.item()) == (3 if p_drop > 0 else 0)  # (the counter node exists only when masks are drawn)
    assert graphed == pytest.approx(eager, rel=1e-4)
    assert len(set(graphed)) == 3  # (the steps do move the loss: nothing is replaying a frozen state)


def test_graphed_step_takes_new_batches():
    """A different batch of the same shape goes through the graph's input tensors."""
    from carca_replication_amd import engine

    fresh, batch = _setup(0.0)
    other = tuple(t.roll(1, 0) for t in batch)
    model_g, opt_g = fresh()
    step = engine.GraphedTrainStep(model_g, opt_g, batch)
    got = [step(batch).clone(), step(other).clone()]
    model_e, opt_e = fresh()
    want = [engine.train_step(model_e, opt_e, batch).clone(), engine.train_step(model_e, opt_e, other).clone()]
    assert [float(x) for x in got] == pytest.approx([float(x) for x in want], rel=1e-5)
    assert abs(float(got[1]) - float(got[0])) > 1e-4


def test_eager_step_between_replays_does_not_orphan_the_graphs_gradients():
    """train(graphed=True) sends an epoch's short last batch through the eager train_step, whose zero_grad(set_to_none)
    rebinds every p.grad: the replays after it must still feed the optimizer THEIR gradients (round-1 bug: the
    optimizer kept applying the eager batch's gradients)."""
    from carca_replication_amd import engine

    fresh, batch = _setup(0.0)
    other = tuple(t.roll(2, 0) for t in batch)
    short = tuple(t[:4].contiguous() for t in batch)
    model_g, opt_g = fresh()
    step = engine.GraphedTrainStep(model_g, opt_g, batch)
    model_e, opt_e = fresh()
    got, want = [], []
    for b, graphed in ((batch, True), (short, False), (other, True), (batch, True)):
        got.append(float(step(b) if graphed else engine.train_step(model_g, opt_g, b)))
        want.append(float(engine.train_step(model_e, opt_e, b)))
    assert got == pytest.approx(want, rel=1e-4)
    for (n, a), (_, b) in zip(model_g.named_parameters(), model_e.named_parameters()):
        if n.endswith("WK.bias"):  # true gradient 0: Adam turns round-off into +-lr steps
            continue
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-5), n


@pytest.mark.parametrize("split", [False, "graph"], ids=["one-stream", "two-streams"])
def test_graphed_step_with_a_big_item_table_captures_the_warm_up_gradient_cache(split, monkeypatch):
    """A model whose item table counts as big (BASELINE config 4: 512 MB) keeps ONE gradient buffer across steps and clears
    only the touched table rows (autograd._grad_buffers).  The captured pass -- which runs the backward on two streams and
    needs room for the second stream's buffers -- must find the buffer of the eager warm-up steps: a buffer created inside
    the capture would be zero-filled as a whole by every replay."""
    from carca_replication_amd import autograd, engine

    monkeypatch.setattr(autograd, "BIG_TABLE_BYTES", 1 << 10)
    monkeypatch.setattr(autograd, "SPLIT_EMBED_BWD", split)
    fresh, batch = _setup(0.0)
    other = tuple(t.roll(1, 0) for t in batch)
    model_g, opt_g = fresh()
    engine.train_step(model_g, opt_g, batch)
    flat0 = model_g.__dict__["_grad_cache"]["flat"]
    step = engine.GraphedTrainStep(model_g, opt_g, batch)
    assert model_g.__dict__["_grad_cache"]["flat"] is flat0
    model_e, opt_e = fresh()
    want = [float(engine.train_step(model_e, opt_e, b)) for b in (batch, other, batch)][1:]  # (the warm-up passes take no optimizer step)
    got = [float(step(b)) for b in (other, batch)]
    assert got == pytest.approx(want, rel=1e-4)
    for (n, a), (_, b) in zip(model_g.named_parameters(), model_e.named_parameters()):
        if n.endswith("WK.bias"):
            continue
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-5), n


def test_only_the_item_table_keeps_a_row_wise_cleared_gradient(monkeypatch):
    """The gradient cache clears table ROWS by the previous step's ids: right for nn.Embedding(n_items, d), wrong for any other
    matrix that happens to be large (feats_embed over a 40 k attribute vocabulary is 64 MB too).  With the threshold down at
    1 KB every 2-D weight of the model is 'large': two steps must still give the gradients of the default path."""
    from carca_replication_amd import autograd, engine

    fresh, batch = _setup(0.0)
    other = tuple(t.roll(1, 0) for t in batch)
    (m1, o1), (m2, o2) = fresh(), fresh()
    monkeypatch.setattr(autograd, "BIG_TABLE_BYTES", 1 << 10)
    for b in (batch, other):
        engine.train_step(m1, o1, b)
    assert "_grad_cache" in m1.__dict__ and len(m1.__dict__["_flat_grad"]["big"]) == 1
    monkeypatch.undo()
    for b in (batch, other):
        engine.train_step(m2, o2, b)
    assert "_grad_cache" not in m2.__dict__
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        if n.endswith("WK.bias"):  # (true gradient 0)
            continue
        scale = float(b.grad.abs().max())
        assert float((a.grad - b.grad).abs().max()) <= 1e-4 * scale + 1e-7, n
