"""gemm_rows_cus_kernel (csrc/gemm_stream.hip): AllEmbedding.feats_embed (carca.py:86: q = [a ; c] W_f^T + b_f, rows with id 0
zeroed by carca.py:94) for a SHORT K with many tiles per CU -- BASELINE's C5 (n_attrs = 512, 1 + 1000 candidates) --
as one persistent workgroup per CU that streams the K steps of its tiles, the context columns as one 8-wide item.

Judged against the product itself in float64 (torch.matmul on the GPU, no kernel of this repo) and against
gemm_rows_cu_kernel (tuning variant 24 switches the streaming kernel off): same sums in the same order up to the context
columns' grouping.  Shapes: C5's own, ragged segment ends (rows that are no multiple of 384, one row, 383 / 385 rows), every
K1 the kernel takes (4..8), N with a narrow last block of 33..66 columns and with none (N = 288), four segments, alpha,
no bias, no row mask."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def tuning():
    from carca_replication_amd import ops

    touched = set()

    def set_(key, value):
        touched.add(key)
        ops.set_tuning(key, value)

    yield set_
    for key in touched:
        ops.set_tuning(key, 0)
    torch.cuda.synchronize()


def _make(rows_list, K0, K1, N, seed, pad=True, bias=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.05
    b = torch.randn(N, device="cuda", generator=g) * 0.01 if bias else None
    segs, ids_all = [], []
    for r in rows_list:
        ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
        if pad:
            ids[torch.rand(r, device="cuda", generator=g) < 0.2] = 0
        segs.append(dict(a0=torch.rand(r, K0, device="cuda", generator=g), a1=torch.rand(r, K1, device="cuda", generator=g), ids=ids))
        ids_all.append(ids)
    return w, b, segs, ids_all


def _fp64(segs, ids_all, w, b, mask_rows, alpha=1.0):
    outs = []
    for sg, ids in zip(segs, ids_all):
        a = torch.cat([sg["a0"], sg["a1"]], dim=1).double()
        y = alpha * (a @ w.double().t())
        if b is not None:
            y = y + b.double()
        outs.append(y * (ids != 0).double()[:, None] if mask_rows else y)
    return outs


def _run(segs, w, b, N, K0, K1, mask_rows=True, alpha=1.0):
    from carca_replication_amd import ops

    ld = (N + 3) // 4 * 4
    ops.gemm_rows_log(True)
    got = ops.gemm_rows(segs, w[:, :K0], N, K0, ld, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=mask_rows, ncols_out=N, alpha=alpha)
    log = ops.gemm_rows_log()
    ops.gemm_rows_log(False)
    torch.cuda.synchronize()
    return got, log


def _check(got, want, ids_all, N, mask_rows, tol=4e-6):
    for x, y, ids in zip(got, want, ids_all):
        scale = float(y.abs().max()) + 1e-30
        assert float((x[:, :N].double() - y).abs().max()) <= tol * scale + 1e-12
        if mask_rows and (ids == 0).any():
            assert float(x[ids == 0][:, :N].abs().max()) == 0.0


def test_c5_shape_streams_and_matches_float64_and_the_one_tile_kernel(tuning):
    """B (L + N) = 128 x (50 + 1001) rows in two segments, K0 = 512, K1 = 6, N = 450: the kernel choice must be the streaming
    kernel; every element within 4e-6 of the float64 product's largest, rows with id 0 exact zeros; against
    gemm_rows_cu_kernel within round-off of one differently grouped 6-term sum."""
    K0, K1, N = 512, 6, 450
    w, b, segs, ids_all = _make([6400, 128128], K0, K1, N, seed=3)
    got, log = _run(segs, w, b, N, K0, K1)
    assert "gemm_rows_cus_kernel<2>" in log, log
    _check(got, _fp64(segs, ids_all, w, b, True), ids_all, N, True)
    tuning(0, 24)
    ref, log2 = _run(segs, w, b, N, K0, K1)
    tuning(0, 0)
    assert "gemm_rows_cus_kernel" not in log2, log2
    for x, y in zip(got, ref):
        assert float((x[:, :N] - y[:, :N]).abs().max()) <= 2e-6 * float(y[:, :N].abs().max())


@pytest.mark.parametrize("rows_list", [[383, 150000], [385, 1, 140001], [70000, 384, 768, 70001], [200000]])
def test_ragged_segment_ends_and_up_to_four_segments(tuning, rows_list):
    K0, K1, N = 256, 6, 450
    w, b, segs, ids_all = _make(rows_list, K0, K1, N, seed=5 + len(rows_list))
    got, log = _run(segs, w, b, N, K0, K1)
    assert "gemm_rows_cus_kernel" in log, log
    _check(got, _fp64(segs, ids_all, w, b, True), ids_all, N, True)


@pytest.mark.parametrize("K1", [4, 5, 7, 8])
def test_every_context_width_the_kernel_takes(tuning, K1):
    """The context item is two 16-byte groups per row, the second one clamped to END at K1 and shifted when stored: K1 = 4 (the
    second group is all zeros), 5 and 7 (shift by 3 and 1), 8 (no shift).  The weights' context columns are a view into [N, K0 +
    K1] rows (what AllEmbedding hands over): the clamped loads must never leave a row."""
    K0, N = 128, 450
    w, b, segs, ids_all = _make([150000, 51], K0, K1, N, seed=20 + K1)
    # poison what lies behind the operands' rows: a group read past K1 would carry it into the sums
    for sg in segs:
        big = torch.full((sg["a1"].shape[0], K1 + 3), float("nan"), device="cuda")
        big[:, :K1] = sg["a1"]
        sg["a1"] = big[:, :K1]
    got, log = _run(segs, w, b, N, K0, K1)
    assert "gemm_rows_cus_kernel" in log, log
    _check(got, _fp64([dict(a0=s["a0"], a1=s["a1"].contiguous()) for s in segs], ids_all, w, b, True), ids_all, N, True)


@pytest.mark.parametrize("N,name", [(450, "<2>"), (449, "<1>"), (448, "<0>"), (434, "<0>"), (288, "<0>"), (417, "<0>")])
def test_narrow_last_column_blocks(tuning, N, name):
    """N = 4 x 96 + 66 / 65 / 64 / 50 / 33 (two MFMA column tiles + 2 / 1 / 0 VALU columns, the second tile partly past N) and
    N = 288 = 3 x 96 (no narrow block: every workgroup on full tiles)."""
    K0, K1 = 192, 6
    w, b, segs, ids_all = _make([90000, 60000], K0, K1, N, seed=40 + N)
    got, log = _run(segs, w, b, N, K0, K1)
    assert "gemm_rows_cus_kernel" + name in log, log
    _check(got, _fp64(segs, ids_all, w, b, True), ids_all, N, True)


def test_alpha_no_bias_no_row_mask(tuning):
    K0, K1, N = 512, 6, 450
    w, _, segs, ids_all = _make([134528], K0, K1, N, seed=77, bias=False)
    got, log = _run(segs, w, None, N, K0, K1, mask_rows=False, alpha=0.37)
    assert "gemm_rows_cus_kernel" in log, log
    _check(got, _fp64(segs, ids_all, w, None, False, alpha=0.37), ids_all, N, False)


def test_shapes_it_declines_stay_on_the_other_kernels(tuning):
    """One round of tiles (C2's row count), K1 outside 4..8, an odd number of K steps: the launcher says 'not mine'."""
    for rows_list, K0, K1 in (([19328], 512, 6), ([150000], 512, 3), ([150000], 96, 6)):
        w, b, segs, ids_all = _make(rows_list, K0, K1, 450, seed=90 + K1)
        got, log = _run(segs, w, b, 450, K0, K1)
        assert "gemm_rows_cus_kernel" not in log, log
        _check(got, _fp64(segs, ids_all, w, b, True), ids_all, 450, True)


# ---- gemm_rows_n96s_kernel: the narrow-output product (joint embedding, carca.py:89) over many rows per CU ---------------


def _joint_case(rows_list, K0, N, seed, lda_extra=0, a_off=0, table=True, pos_T=0, pad=True):
    """Rows as the joint embedding meets them: A is a column slice [a_off : a_off + K0] of a wider workspace (zq), the
    weights a column slice of joint_embed.weight, the item term gathered by id from a table, positions for the first segment."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    ldw = a_off + K0 + lda_extra
    w_full = (torch.rand(N, ldw, device="cuda", generator=g) * 2 - 1) * 0.05
    w = w_full[:, a_off:a_off + K0]
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    tab = torch.randn(1000, N + 6, device="cuda", generator=g)[:, :N] if table else None
    pos = torch.randn(pos_T, N, device="cuda", generator=g) if pos_T else None
    segs, ids_all = [], []
    for i, r in enumerate(rows_list):
        ids = torch.randint(1, 1000, (r,), device="cuda", dtype=torch.int32, generator=g)
        if pad:
            ids[torch.rand(r, device="cuda", generator=g) < 0.2] = 0
        big = torch.rand(r, ldw, device="cuda", generator=g)
        sg = dict(a0=big[:, a_off:a_off + K0], ids=ids)
        if pos_T and i == 0:
            sg.update(T=pos_T, add_pos=True)
        segs.append(sg)
        ids_all.append(ids)
    return w, b, tab, pos, segs, ids_all


def _joint_fp64(segs, ids_all, w, b, tab, pos, mask_rows=True, alpha=1.0):
    outs = []
    for sg, ids in zip(segs, ids_all):
        y = alpha * (sg["a0"].double() @ w.double().t()) + b.double()
        if tab is not None:
            y = y + tab.double()[ids.long()]
        if sg.get("add_pos"):
            T = sg["T"]
            y = y + pos.double()[torch.arange(y.shape[0], device="cuda") % T]
        outs.append(y * (ids != 0).double()[:, None] if mask_rows else y)
    return outs


def _joint_run(segs, w, b, tab, pos, N, K0, ld=96, mask_rows=True, alpha=1.0):
    from carca_replication_amd import ops

    ops.gemm_rows_log(True)
    got = ops.gemm_rows(segs, w, N, K0, ld, bias=b, pos=pos, mask_rows=mask_rows, ncols_out=ld, alpha=alpha, add_table=tab)
    log = ops.gemm_rows_log()
    ops.gemm_rows_log(False)
    torch.cuda.synchronize()
    return got, log


@pytest.mark.parametrize("rows_list,K0,a_off", [([6400, 128128], 450, 90), ([6400, 128128], 540, 0), ([110000], 450, 90),
                                               ([159, 70001, 17, 65000], 450, 90), ([120001], 256, 0), ([200001], 640, 128)])
def test_joint_product_streams_and_matches_float64(tuning, rows_list, K0, a_off):
    """C5's joint product and its like (inference: K = g = 450 out of zq's q columns, rows 8-byte aligned, K no multiple of 4, the
    item term from the table; training's order: K = d + g = 540), four segments with ragged ends (a share cut by a segment's
    end, segments shorter than a block, one row past a multiple of 16), K = 256 (four full stages) and 640 / N = 128-wide
    workspace.  Every element against the float64 product; columns N .. 95 of the output are zeros; and against the
    tiled kernels (tuning variant 26)."""
    N = 90
    w, b, tab, pos, segs, ids_all = _joint_case(rows_list, K0, N, seed=len(rows_list) * 7 + K0, a_off=a_off)
    got, log = _joint_run(segs, w, b, tab, pos, N, K0)
    assert "gemm_rows_n96s_kernel" in log, log
    want = _joint_fp64(segs, ids_all, w, b, tab, pos)
    _check(got, want, ids_all, N, True)
    for x in got:
        assert float(x[:, N:96].abs().max()) == 0.0
    tuning(0, 26)
    ref, log2 = _joint_run(segs, w, b, tab, pos, N, K0)
    tuning(0, 0)
    assert "gemm_rows_n96s_kernel" not in log2, log2
    for x, y in zip(got, ref):
        assert float((x[:, :N] - y[:, :N]).abs().max()) <= 2e-6 * float(y[:, :N].abs().max())


@pytest.mark.parametrize("N,ld", [(96, 96), (65, 68), (90, 92)])
def test_joint_product_widths_positions_alpha(tuning, N, ld):
    """N = 96 (no padding column), 65 (the last column tile holds one column), output rows narrower than 96 floats; positional
    rows on the first segment (T = 50: row % T), no table, alpha, no row mask."""
    K0 = 320
    w, b, tab, pos, segs, ids_all = _joint_case([50 * 1000, 70000], K0, N, seed=N, table=False, pos_T=50)
    got, log = _joint_run(segs, w, b, None, pos, N, K0, ld=ld, mask_rows=False, alpha=1.7)
    assert "gemm_rows_n96s_kernel" in log, log
    _check(got, _joint_fp64(segs, ids_all, w, b, None, pos, mask_rows=False, alpha=1.7), ids_all, N, False)


def test_joint_product_declines_small_batches(tuning):
    """C2's row count (19,328: one round of the one-block-per-CU kernel) and C3's (77,312: under 2.5 blocks per CU, measured
    slower) stay where they were; tuning variant 27 forces the kernel there and it is still right."""
    for rows_list in ([6400, 12928], [25600, 51712]):
        w, b, tab, pos, segs, ids_all = _joint_case(rows_list, 450, 90, seed=1, a_off=90)
        got, log = _joint_run(segs, w, b, tab, pos, 90, 450)
        assert "gemm_rows_n96s_kernel" not in log, log
        _check(got, _joint_fp64(segs, ids_all, w, b, tab, pos), ids_all, 90, True)
    tuning(0, 27)
    got, log = _joint_run(segs, w, b, tab, pos, 90, 450)
    tuning(0, 0)
    assert "gemm_rows_n96s_kernel" in log, log
    _check(got, _joint_fp64(segs, ids_all, w, b, tab, pos), ids_all, 90, True)
