"""Deterministic mode (ops.set_deterministic, include/carca_hip.h tuning key 8): no fp32 atomics in the backward pass --
every gradient accumulation goes through a 64-bit fixed-point shadow whose integer sums do not depend on the order in
which workgroups arrive.  Same inputs => same bits, so the trajectory tests that the default path can only hold
statistically (Adam turns a round-off-sized difference of a near-cancelling gradient into a +-lr step) are held with
torch.equal here, on EVERY tensor -- attention key biases (true gradient 0: pure round-off) included."""
import copy

import pytest
import torch

from tests.model_util import build_model

pytestmark = pytest.mark.gpu


@pytest.fixture
def det():
    from carca_replication_amd import ops

    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


def _all_equal(model_a, opt_a, model_b, opt_b, what):
    for (n, a), (_, b) in zip(model_a.named_parameters(), model_b.named_parameters()):
        assert torch.equal(a, b), f"{what}: parameter {n}: max |diff| {float((a - b).abs().max())}"
        sa, sb = opt_a.state[a], opt_b.state[b]
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(sa[key], sb[key]), f"{what}: {key} of {n}"


def _c2_like_batch(B, L, n_items, n_attrs, n_ctx, seed):
    from carca_replication_amd.synth import eval_batch

    profile, pos, _ = eval_batch(B, L, L, n_items, n_attrs, n_ctx, seed=seed)
    px = profile[0]
    o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
    return tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                    torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))


def test_train_steps_are_bit_reproducible_and_close_to_the_default_path(det):
    """Three train steps (fwd + bwd + Adam) of a C2-shaped model, twice from the same weights: bit-identical parameters,
    gradients and optimizer state.  B = 128 users x 150 rows: every split / stream-K flush / scatter duplicate of the
    backward pass is in play (the default path differs run to run here).  And the mode computes the SAME gradients as
    the default path up to fp32 round-off."""
    from carca_replication_amd import engine, ops
    from carca_replication_amd.optim import Adam

    B, L, d, g, H, n_attrs, n_ctx, n_items = 128, 50, 90, 450, 3, 512, 6, 3000
    torch.manual_seed(0)
    base = build_model(dict(d=d, H=H, n_blocks=2, encoding="learnable"), n_items, g, n_ctx, n_attrs, L).cuda().train()
    batches = [_c2_like_batch(B, L, n_items, n_attrs, n_ctx, s) for s in (1, 2, 1)]
    runs = []
    for _ in range(2):
        m = copy.deepcopy(base)
        o = Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.98))
        losses = [float(engine.train_step(m, o, bt)) for bt in batches]
        runs.append((m, o, losses))
    (ma, oa, la), (mb, ob, lb) = runs
    assert la == lb
    _all_equal(ma, oa, mb, ob, "run 1 vs run 2")
    for (n, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(a.grad, b.grad), n
    # against the default path (fp32 atomics): first-step gradients within round-off of each tensor's largest entry
    m1, m2 = copy.deepcopy(base), copy.deepcopy(base)
    engine.train_step(m1, torch.optim.SGD(m1.parameters(), lr=0.0), batches[0])
    ops.set_deterministic(False)
    engine.train_step(m2, torch.optim.SGD(m2.parameters(), lr=0.0), batches[0])
    ops.set_deterministic(True)
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        scale = float(b.grad.abs().max())
        assert float((a.grad - b.grad).abs().max()) <= 2e-6 * scale + 1e-9, n


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_graphed_step_equals_the_eager_step_bit_for_bit(det, p_drop, monkeypatch):
    """tests/test_hip_graph.py::test_graphed_step_is_the_eager_step held tight: replay t of the captured step and eager
    step t run the same kernels on the same numbers, so parameters and optimizer state are EQUAL after three steps."""
    from carca_replication_amd import engine, ops
    from tests.test_hip_graph import _setup

    fresh, batch = _setup(p_drop)
    base = 123456789
    monkeypatch.setattr(ops, "new_dropout_seed", lambda: base)
    model_g, opt_g = fresh()
    step = engine.GraphedTrainStep(model_g, opt_g, batch)
    model_e, opt_e = fresh()
    for t in range(1, 4):
        lg = float(step(batch))
        monkeypatch.setattr(ops, "new_dropout_seed", lambda t=t: base + t)
        le = float(engine.train_step(model_e, opt_e, batch))
        assert lg == le, (t, lg, le)
    _all_equal(model_g, opt_g, model_e, opt_e, "graph vs eager")


def test_sharded_step_equals_the_plain_step_bit_for_bit(det):
    """tests/test_hip_optim.py's one-rank sharded chain held tight: a 1-rank sum is the identity, so the sharded step
    (global mask count, flat buffer reduced in place through RCCL, one-launch Adam) and its hipGraph variant must leave
    EXACTLY the plain step's parameters."""
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
            assert cdist.last_reduce.get("path") == "flat-inplace"
            lg = step_g(bt)
            cdist.FORCE_COLLECTIVES = False
            lp = engine.train_step(model_p, opt_p, bt)
            cdist.FORCE_COLLECTIVES = True
            assert float(ls) == float(lp) == float(lg)
        _all_equal(model_s, opt_s, model_p, opt_p, "sharded vs plain")
        _all_equal(model_g, opt_g, model_p, opt_p, "sharded graph vs plain")
    finally:
        cdist.FORCE_COLLECTIVES = False
        dist.destroy_process_group()


def test_c4_touched_row_trajectory_equals_the_dense_one_bit_for_bit(det, monkeypatch):
    """tests/test_c4_scale.py held tight (BASELINE configs[3]: 1 M items, d = 128, g = 640, H = 4): the step that keeps one
    gradient buffer cleared row-wise and updates only the rows ever touched against the step that zero-fills the whole
    512 MB gradient and sweeps every row with the same optimizer -- equal parameters, exp_avg and exp_avg_sq, every
    tensor, every element."""
    from carca_replication_amd import autograd, engine
    from carca_replication_amd.optim import Adam

    B, L, d, g, H, n_attrs, n_ctx, n_items = 128, 50, 128, 640, 4, 256, 6, 1_000_001
    torch.manual_seed(0)
    model_a = build_model(dict(d=d, H=H, n_blocks=2), n_items, g, n_ctx, n_attrs, L).cuda().train()
    model_b = copy.deepcopy(model_a)
    opt_a = Adam(model_a.parameters(), lr=1e-3, betas=(0.9, 0.98))
    opt_b = Adam(model_b.parameters(), lr=1e-3, betas=(0.9, 0.98))
    monkeypatch.setattr(opt_b, "mark_rows", lambda *a, **k: False)  # ... and Adam over every row
    batches = [_c2_like_batch(B, L, n_items, n_attrs, n_ctx, s) for s in (7, 8, 7)]
    for bt in batches:
        la = engine.train_step(model_a, opt_a, bt)
        monkeypatch.setattr(autograd, "BIG_TABLE_BYTES", 1 << 62)  # the dense step: fresh zero fill of the whole table
        lb = engine.train_step(model_b, opt_b, bt)
        monkeypatch.setattr(autograd, "BIG_TABLE_BYTES", 64 * 2 ** 20)
        assert float(la) == float(lb)
    assert "_grad_cache" in model_a.__dict__ and "_grad_cache" not in model_b.__dict__
    assert "row_touched" in opt_a.state[model_a.embeds.items_embed.weight]
    assert "row_touched" not in opt_b.state[model_b.embeds.items_embed.weight]
    _all_equal(model_a, opt_a, model_b, opt_b, "touched-row vs dense")


def test_metric_sums_are_added_in_a_fixed_order(det):
    from carca_replication_amd import ops

    torch.manual_seed(3)
    y = torch.rand(4096, 101, device="cuda")
    a, _ = ops.rank_metrics(y, 10)
    vals = {tuple(ops.rank_metrics(y, 10)[0].tolist()) for _ in range(20)}
    assert vals == {tuple(a.tolist())}
    ops.set_deterministic(False)
    b, rank = ops.rank_metrics(y, 10, want_rank=True)
    ops.set_deterministic(True)
    assert float(a[0]) == float(b[0]) == float((rank < 10).sum()) and abs(float(a[1]) - float(b[1])) < 1e-3 * float(b[1])
