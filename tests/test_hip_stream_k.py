"""The hand-over inside the headline kernel (gemm_rows_sk_kernel, csrc/gemm.hip: AllEmbedding.feats_embed, carca.py:86, at
C2 size): the last column block's workgroup computes the last K steps of its four neighbours' tiles and hands the partial
tiles over through memory (sc1 stores, a flag; the taker polls, acquires at agent scope and adds them).

What must hold, and what these tests pin ("What's weak" 1-2):
  * replays of a hipGraph go through the same partial-tile memory every time, with the flags cleared once at capture: a
    replay must never read the PREVIOUS replay's partials (a 5 % error of the K sum) -- train step and eval forward,
    alternating two different batches, against eager launches of the kernel WITHOUT the hand-over (tuning variant 15);
  * a taker whose partial never arrives gives up after a bounded wait, says so in the library's error word, and both
    carca_poll_errors and the next stream-K launch fail loudly -- run once, deterministically, with one flag withheld and
    the bound cut to 2^10 sleeps (~0.5 ms), i.e. the fix of round 3's 200 s hang without a hang."""
import copy

import pytest
import torch

from oracle import carca_oracle as O
from tests.model_util import build_model, dev

pytestmark = pytest.mark.gpu

C2 = dict(n_items=3000, n_attrs=4096, n_ctx=6, g=450, L=50, N=101, B=128)  # (the table is smaller than C2's: not on this path)
CFG = O.CarcaConfig(d=90, H=3, n_blocks=2)


def _params():
    return O.perturb_params(O.init_params(CFG, C2["n_items"], C2["g"], C2["n_ctx"], C2["n_attrs"], C2["L"], seed=0), seed=1)


def _model(P):
    m = build_model(dict(d=CFG.d, H=CFG.H, n_blocks=CFG.n_blocks), C2["n_items"], C2["g"], C2["n_ctx"], C2["n_attrs"], C2["L"])
    m.load_state_dict(copy.deepcopy(P), strict=True)
    return m.cuda()


def _train_batch(seed):
    from carca_replication_amd.synth import eval_batch

    L = C2["L"]
    profile, pos, _ = eval_batch(C2["B"], L, L, C2["n_items"], C2["n_attrs"], C2["n_ctx"], seed=seed)
    px = profile[0]
    o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
    o_a = torch.cat([pos[1], pos[1].flip(1)], dim=1)
    o_c = torch.cat([pos[2], pos[2]], dim=1)
    y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
    return tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, o_a, o_c, y_true))


@pytest.fixture
def tuning():
    """carca_set_tuning with every key it touched put back to 0 afterwards, whatever the test does."""
    from carca_replication_amd import ops

    touched = set()

    def set_(key, value):
        touched.add(key)
        ops.set_tuning(key, value)

    yield set_
    for key in touched:
        ops.set_tuning(key, 0)
    torch.cuda.synchronize()
    try:
        ops.poll_errors()  # (leave no error word behind for the next test)
    except Exception:
        pass


def test_graph_replays_of_the_train_step_never_see_the_previous_replays_partials(tuning):
    """GraphedTrainStep at C2 size (forward feature GEMM = gemm_rows_sk_kernel, captured), 6 replays alternating two
    batches, deterministic mode.  Before every step both models receive the SAME in-place perturbation of every parameter
    (an optimizer with lr = 0 runs behind the replay: Adam would turn round-off into +-lr steps and hide or fake
    differences), so replay t must reproduce, to the round-off of the two kernels' different grouping of the K sum, the
    loss and every gradient of an eager step whose feature GEMM takes NO hand-over."""
    from carca_replication_amd import engine, ops
    from carca_replication_amd.optim import Adam

    ops.set_deterministic(True)
    try:
        P = _params()
        batches = [_train_batch(21), _train_batch(22)]
        model_g, model_e = _model(P).train(), _model(P).train()
        opt_g = Adam(model_g.parameters(), lr=0.0, betas=(0.9, 0.98))
        opt_e = Adam(model_e.parameters(), lr=0.0, betas=(0.9, 0.98))
        step = engine.GraphedTrainStep(model_g, opt_g, batches[0])
        assert step.library_bytes > 20e6  # (the stream-K partials alone are 30 MB: the capture took the hand-over kernel)
        gen = torch.Generator(device="cuda").manual_seed(5)
        losses = []
        for t in range(6):
            with torch.no_grad():
                for pg, pe in zip(model_g.parameters(), model_e.parameters()):
                    bump = 2e-3 * float(pe.abs().max()) * torch.randn(pe.shape, device="cuda", generator=gen)
                    pg.add_(bump)
                    pe.add_(bump)
            batch = batches[t % 2]
            loss_g = float(step(batch))
            tuning(0, 15)  # the one-tile-per-workgroup kernel: no partial tiles, no flags
            loss_e = float(engine.train_step(model_e, opt_e, batch))
            tuning(0, 0)
            losses.append(loss_g)
            assert loss_g == pytest.approx(loss_e, rel=1e-5), t
            for (n, a), (_, b) in zip(model_g.named_parameters(), model_e.named_parameters()):
                if n.endswith("WK.bias"):  # true gradient 0 (softmax is shift-invariant): what is there is round-off
                    continue
                # (a stale partial is a 5 % error of a 384 x 96 tile of q: gradients off by percents and the loss above by
                # ~1e-3.  The bound here leaves room for the one thing round-off can flip: a LeakyReLU pre-activation
                # within an ulp of 0, which moves single rows of a gradient by ~1e-4 of the tensor's largest entry)
                scale = float(b.grad.abs().max())
                assert float((a.grad - b.grad).abs().max()) <= 2e-3 * scale + 1e-12, (t, n)
        assert len(set(losses)) == 6  # (weights and batches really change from replay to replay)
        scope = step.scope
        step.close()
        assert scope != 0 and ops.capture_bytes(scope) == 0
    finally:
        ops.set_deterministic(False)


def test_graph_replays_of_the_eval_forward_never_see_the_previous_replays_partials(tuning):
    """The inference forward (one carca_forward call) captured once and replayed over two alternating batches."""
    from carca_replication_amd import ops
    from carca_replication_amd.synth import eval_batch

    model = _model(_params()).eval()
    batches = []
    for seed in (31, 32):
        profile, target, _ = eval_batch(C2["B"], C2["L"], C2["N"], C2["n_items"], C2["n_attrs"], C2["n_ctx"], seed=seed)
        batches.append((dev(profile), dev(target)))
    with torch.no_grad():
        tuning(0, 15)
        want = [model(profile=p, targets=[t]).clone() for p, t in batches]
        tuning(0, 0)
        assert float((want[0] - want[1]).abs().max()) > 1e-2
        static_p = tuple(t.clone() for t in batches[0][0])
        static_t = tuple(t.clone() for t in batches[0][1])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model(profile=static_p, targets=[static_t])
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            scope = ops.capture_scope()
            y = model(profile=static_p, targets=[static_t])
        assert scope != 0 and ops.capture_bytes(scope) > 20e6
        for it in range(8):
            p, t = batches[it % 2]
            for dst, src in zip(static_p + static_t, p + t):
                dst.copy_(src)
            graph.replay()
            assert float((y - want[it % 2]).abs().max()) < 2e-5, it
        torch.cuda.synchronize()
        del graph
        ops.capture_release(scope)
        assert ops.capture_bytes(scope) == 0


def test_a_taker_that_gives_up_is_reported_by_the_poll_and_by_the_next_launch(tuning):
    """Tuning key 13 withholds ONE giver's flag; key 12 cuts the taker's bound to 2^10 sleeps.  The launch ends (no hang)
    and the error word is set: carca_poll_errors raises, and -- second round -- so does the next stream-K launch, once;
    after that the kernel works as before.  (What the taker adds after giving up is whatever the partial's memory holds;
    here that happens to be the right tile -- only the flag was withheld -- so the output is not asserted on.)"""
    from carca_replication_amd import ops
    from carca_replication_amd._lib import CarcaHipError
    from carca_replication_amd.synth import eval_batch

    model = _model(_params()).eval()
    profile, target, _ = eval_batch(C2["B"], C2["L"], C2["N"], C2["n_items"], C2["n_attrs"], C2["n_ctx"], seed=41)
    p, t = dev(profile), dev(target)
    with torch.no_grad():
        want = model(profile=p, targets=[t]).clone()
        torch.cuda.synchronize()
        ops.poll_errors()  # clean
        for report in ("poll", "next launch"):
            tuning(12, 10)
            tuning(13, 2)  # (flag 1: the partial tile workgroup 1 hands to workgroup 0 -- profile rows of the first users)
            got = model(profile=p, targets=[t])
            torch.cuda.synchronize()
            tuning(12, 0)
            tuning(13, 0)
            assert bool(torch.isfinite(got).all())
            if report == "poll":
                with pytest.raises(CarcaHipError, match="gave up"):
                    ops.poll_errors()
            else:
                with pytest.raises(CarcaHipError, match="gave up"):
                    model(profile=p, targets=[t])
            ops.poll_errors()  # reported once, then cleared
            again = model(profile=p, targets=[t])
            torch.cuda.synchronize()
            ops.poll_errors()
            assert torch.equal(again, want)


def test_deterministic_mode_keeps_a_non_finite_gradient_visible():
    """grad_add's fixed-point shadow cannot hold NaN / Inf / |v| >= 2^27: such a contribution takes the fp32 atomic instead
     -- a diverging run must not read finite gradients."""
    from carca_replication_amd import ops

    ops.set_deterministic(True)
    try:
        n, d = 64, 8
        flat = torch.zeros(n * d, device="cuda")
        shadow = torch.zeros(n * d, dtype=torch.int64, device="cuda")
        ids = torch.arange(1, 5, dtype=torch.int32, device="cuda")
        dz = torch.ones(4, d, device="cuda")
        dz[1, 3] = float("inf")
        dz[2, 0] = float("nan")
        dz[3, 1] = 3e8
        ops.det_begin(flat, shadow)
        ops.embed_scatter(dz, ids, d, 1.0, flat.view(n, d))
        ops.det_flush(flat, shadow, 0, n * d)
        ops.det_begin(None, None)
        torch.cuda.synchronize()
        tab = flat.view(n, d)
        assert float(tab[1, 0]) == 1.0 and torch.isinf(tab[2, 3]) and torch.isnan(tab[3, 0]) and float(tab[4, 1]) == 3e8
    finally:
        ops.set_deterministic(False)


def _fp64_product(segs, ids_all, w, b, N):
    """The product itself in float64 on the GPU (torch.matmul, no kernel of this repo): ([a0 | a1] W^T + b) * (id != 0),
    carca.py:86,94 -- the reference the compacting kernel is judged against beside the every-row kernels."""
    outs = []
    for sg, ids in zip(segs, ids_all):
        a = torch.cat([sg["a0"], sg["a1"]], dim=1).double()
        y = a @ w.double().t() + b.double()
        outs.append(y * (ids != 0).double()[:, None])
    return outs


def _check_against_fp64(got, want64, ids_all, N, tol=6e-6):
    for x, y, ids in zip(got, want64, ids_all):
        scale = float(y.abs().max()) + 1e-30
        assert float((x[:, :N].double() - y).abs().max()) <= tol * scale + 1e-12  # (an fp32 sum over K = 2054 against fp64)
        if (ids == 0).any():
            assert float(x[ids == 0][:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("pattern", ["scattered", "left_padded", "one_segment_all_padding", "nothing_left_out", "all_padding"])
def test_the_compacting_kernel_on_awkward_id_patterns(tuning, pattern):
    """gemm_rows_skc_kernel (rows with id 0 left out, stretches over the kept rows' blocks) against the kernels that multiply
    every row (tuning variant 23), on id patterns the batches of the hot path never produce: zeros sprinkled anywhere, a
    segment that is nothing but padding, no padding at all, nothing but padding.  Kept rows agree to the round-off of another
    grouping of the K sum; left-out rows are exact zeros."""
    from carca_replication_amd import ops

    K0, K1, N = 2048, 6, 450
    rows = [6400, 5000, 3333]
    g = torch.Generator(device="cuda").manual_seed(3)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.03
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    segs, ids_all = [], []
    for si, r in enumerate(rows):
        a = torch.rand(r, K0, device="cuda", generator=g)
        c = torch.rand(r, K1, device="cuda", generator=g)
        ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
        if pattern == "scattered":
            ids[torch.rand(r, device="cuda", generator=g) < 0.37] = 0
        elif pattern == "left_padded":
            t = torch.arange(r, device="cuda") % 50
            npad = torch.randint(0, 48, ((r + 49) // 50,), device="cuda", generator=g).repeat_interleave(50)[:r]
            ids[t < npad] = 0
        elif pattern == "one_segment_all_padding" and si == 1:
            ids[:] = 0
        elif pattern == "all_padding":
            ids[:] = 0
        segs.append(dict(a0=a, a1=c, ids=ids))
        ids_all.append(ids)

    def run():
        return ops.gemm_rows(segs, w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=True, ncols_out=N)

    ops.gemm_rows_log(True)
    got = run()
    log = ops.gemm_rows_log()
    ops.gemm_rows_log(False)
    assert "gemm_rows_skc_kernel" in log
    tuning(0, 23)
    want = run()
    tuning(0, 0)
    torch.cuda.synchronize()
    ops.poll_errors()
    for x, y, ids in zip(got, want, ids_all):
        assert float(x[ids == 0][:, :N].abs().max() if (ids == 0).any() else 0.0) == 0.0
        scale = float(y[:, :N].abs().max()) + 1e-30
        assert float((x[:, :N] - y[:, :N]).abs().max()) <= 4e-6 * scale + 1e-12  # (two groupings of a K = 2054 fp32 sum)
    _check_against_fp64(got, _fp64_product(segs, ids_all, w, b, N), ids_all, N)
    again = run()
    for x, y in zip(got, again):
        assert torch.equal(x[:, :N], y[:, :N])  # the same bits from run to run


def test_the_compacting_kernel_at_b512_rows(tuning):
    """C3's row count (77,312 rows, 202 row blocks when nothing is left out): stretches of five and more row blocks per
    workgroup -- the launcher's bound on what the kernel's LDS lists hold."""
    from carca_replication_amd import ops

    K0, K1, N, B = 2048, 6, 450, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.03
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    for full in (True, False):
        segs, ids_all = [], []
        for T in (50, 101):
            r = B * T
            ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
            if not full and T == 50:
                t = torch.arange(r, device="cuda") % 50
                ids[t < torch.randint(0, 48, (B,), device="cuda", generator=g).repeat_interleave(50)] = 0
            segs.append(dict(a0=torch.rand(r, K0, device="cuda", generator=g), a1=torch.rand(r, K1, device="cuda", generator=g),
                             ids=ids))
            ids_all.append(ids)

        def run():
            return ops.gemm_rows(segs, w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=True, ncols_out=N)

        ops.gemm_rows_log(True)
        got = run()
        assert "gemm_rows_skc_kernel" in ops.gemm_rows_log()
        ops.gemm_rows_log(False)
        tuning(0, 23)
        want = run()
        tuning(0, 0)
        torch.cuda.synchronize()
        ops.poll_errors()
        for x, y, ids in zip(got, want, ids_all):
            scale = float(y[:, :N].abs().max())
            assert float((x[:, :N] - y[:, :N]).abs().max()) <= 4e-6 * scale
            if (ids == 0).any():
                assert float(x[ids == 0][:, :N].abs().max()) == 0.0
        _check_against_fp64(got, _fp64_product(segs, ids_all, w, b, N), ids_all, N)


@pytest.mark.parametrize("N", [640, 449, 434])
def test_the_compacting_kernel_with_other_narrow_column_blocks(tuning, N):
    """N = 640 = 6 x 96 + 64 (C4's g: the narrow block is exactly two MFMA column tiles, teams of six) and N = 449 (one VALU
    column) and N = 434 (a narrow block of 50 columns) through gemm_rows_skc_kernel, against the kernels that multiply every row."""
    from carca_replication_amd import ops

    K0, K1 = 2048, 6
    g = torch.Generator(device="cuda").manual_seed(9)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.03
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    segs, ids_all = [], []
    for r in (6400, 12928):
        ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
        if r == 6400:
            t = torch.arange(r, device="cuda") % 50
            ids[t < torch.randint(0, 48, (r // 50,), device="cuda", generator=g).repeat_interleave(50)] = 0
        segs.append(dict(a0=torch.rand(r, K0, device="cuda", generator=g), a1=torch.rand(r, K1, device="cuda", generator=g), ids=ids))
        ids_all.append(ids)
    ld = (N + 3) // 4 * 4

    def run():
        return ops.gemm_rows(segs, w[:, :K0], N, K0, ld, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=True, ncols_out=N)

    ops.gemm_rows_log(True)
    got = run()
    assert "gemm_rows_skc_kernel" in ops.gemm_rows_log()
    ops.gemm_rows_log(False)
    tuning(0, 23)
    want = run()
    tuning(0, 0)
    torch.cuda.synchronize()
    ops.poll_errors()
    for x, y, ids in zip(got, want, ids_all):
        scale = float(y[:, :N].abs().max())
        assert float((x[:, :N] - y[:, :N]).abs().max()) <= 4e-6 * scale
        if (ids == 0).any():
            assert float(x[ids == 0][:, :N].abs().max()) == 0.0
    _check_against_fp64(got, _fp64_product(segs, ids_all, w, b, N), ids_all, N)


@pytest.mark.parametrize("rows", [[12800, 8448, 8448, 8448], [6400, 4224, 4224, 4224], [6400, 6400, 5760, 64]])
def test_the_compacting_kernel_with_four_segments(tuning, rows):
    """CARCA_MAX_SEGS = 4 segments (a profile and three target groups, carca.py:424): ADVICE r4 -- the kernel's count of
    64-row chunks dropped the FOURTH segment's, so its prologue summed uninitialised LDS.  Row counts that the launcher sends
    to the one-block-per-CU kernels (whole rounds of 384-row blocks): 38,144 rows = 596 chunks take the path that reads the
    ids from memory (more than 27,648 ids), 19,072 rows the path that keeps them in LDS; the last case ends in a segment of
    one chunk.  Against the every-row kernels and the float64 product."""
    from carca_replication_amd import ops

    K0, K1, N = 2048, 6, 450
    g = torch.Generator(device="cuda").manual_seed(11)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.03
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    segs, ids_all = [], []
    for si, r in enumerate(rows):
        ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
        T = 50 if r % 50 == 0 else 101 if r % 101 == 0 else 64
        t = torch.arange(r, device="cuda") % T
        ids[t < torch.randint(0, T - 2, ((r + T - 1) // T,), device="cuda", generator=g).repeat_interleave(T)[:r]] = 0
        segs.append(dict(a0=torch.rand(r, K0, device="cuda", generator=g), a1=torch.rand(r, K1, device="cuda", generator=g), ids=ids))
        ids_all.append(ids)

    def run():
        return ops.gemm_rows(segs, w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=True, ncols_out=N)

    ops.gemm_rows_log(True)
    got = run()
    assert "gemm_rows_skc_kernel" in ops.gemm_rows_log()
    ops.gemm_rows_log(False)
    tuning(0, 23)
    want = run()
    tuning(0, 0)
    torch.cuda.synchronize()
    ops.poll_errors()
    for x, y in zip(got, want):
        scale = float(y[:, :N].abs().max())
        assert float((x[:, :N] - y[:, :N]).abs().max()) <= 4e-6 * scale
    _check_against_fp64(got, _fp64_product(segs, ids_all, w, b, N), ids_all, N)


@pytest.mark.parametrize("N", [450, 256])
def test_the_compacting_kernel_on_a_short_k(tuning, N):
    """K0 = 512 (16 K steps: BASELINE's C5 / the CLI's default n_attrs) through gemm_rows_skc_kernel -- admitted below its
    default bound of 64 steps by tuning key 19: stretches of about one row block, every piece a few steps long.  N = 256 is
    the CLI's g (a narrow block of exactly two MFMA column tiles, teams of two)."""
    from carca_replication_amd import ops

    K0, K1 = 512, 6
    g = torch.Generator(device="cuda").manual_seed(13)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.05
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    segs, ids_all = [], []
    for r in (12800, 25856):
        ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
        if r == 12800:
            t = torch.arange(r, device="cuda") % 50
            ids[t < torch.randint(0, 48, (r // 50,), device="cuda", generator=g).repeat_interleave(50)] = 0
        segs.append(dict(a0=torch.rand(r, K0, device="cuda", generator=g), a1=torch.rand(r, K1, device="cuda", generator=g), ids=ids))
        ids_all.append(ids)
    ld = (N + 3) // 4 * 4

    def run():
        return ops.gemm_rows(segs, w[:, :K0], N, K0, ld, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=True, ncols_out=N)

    tuning(19, 16)
    ops.gemm_rows_log(True)
    got = run()
    log = ops.gemm_rows_log()
    ops.gemm_rows_log(False)
    tuning(19, 0)
    assert "gemm_rows_skc_kernel" in log
    torch.cuda.synchronize()
    ops.poll_errors()
    _check_against_fp64(got, _fp64_product(segs, ids_all, w, b, N), ids_all, N)
