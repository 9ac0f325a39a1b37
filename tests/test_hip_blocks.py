"""GPU unit tests of the dense building blocks (C ABI: carca_gemm_rows, carca_gemm_wgrad, ...).

These are floating-point kernels, so each is compared with a plain torch fp32/fp64 statement of the
same product on the same seeded inputs (tolerance: fp32 accumulation over K <= 4102 terms).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1)


def _close(got, want, tol=2e-5):
    scale = float(want.abs().max()) + 1e-12
    err = float((got.double().cpu() - want.double()).abs().max())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("rows,N,K0,K1", [(300, 90, 77, 0), (129, 450, 4096, 6), (64, 33, 5, 3), (1, 1, 1, 0),
                                          (257, 200, 96, 96)])
def test_gemm_rows_plain(rows, N, K0, K1):
    from carca_replication_amd import ops

    a0, a1 = _rand(rows, K0, seed=1), _rand(rows, max(K1, 1), seed=2)
    bt = _rand(N, K0 + K1, seed=3)
    bias = _rand(N, seed=4)
    want = a0.double() @ bt[:, :K0].double().T + bias.double()
    if K1:
        want = want + a1.double() @ bt[:, K0:].double().T
    btc = bt.cuda()
    out_ld = ((N + 3) // 4) * 4 + 4
    (got,) = ops.gemm_rows([dict(a0=a0.cuda(), a1=a1.cuda() if K1 else None)], btc[:, :K0], N, K0, out_ld,
                           bt1=btc[:, K0:] if K1 else None, K1=K1, bias=bias.cuda())
    _close(got[:, :N], want)
    assert float(got[:, N:].abs().max()) == 0.0  # pad columns are written as zeros


@pytest.fixture(params=[0, 1, 2, 4, 7], ids=["auto", "tile128x96", "one-block-per-cu", "no-buffer-loads", "cu-384x128"])
def gemm_variant(request):
    """Every carca_gemm_rows test runs under each kernel choice (tuning key 0); the forced ones take effect where the
    launcher's preconditions hold and otherwise fall through to the default, so all shapes stay valid."""
    from carca_replication_amd import _lib

    lib = _lib.load()
    lib.carca_set_tuning(0, request.param)
    yield request.param
    lib.carca_set_tuning(0, 0)


@pytest.mark.parametrize("rows,T,N,K0,K1", [((500, 777), 7, 200, 300, 6), ((384, 1), 1, 96, 64, 0),
                                            ((1000,), 50, 450, 4096, 6), ((40,), 1, 130, 65, 3)])
def test_gemm_rows_kernel_choices(gemm_variant, rows, T, N, K0, K1):
    """Ragged row blocks, a ragged K tail, a second k-source and unaligned strided rows under each kernel."""
    from carca_replication_amd import _lib, ops

    bt, bias = _rand(N, K0 + K1, seed=3), _rand(N, seed=4)
    btc = bt.cuda()
    segs, wants = [], []
    for i, r in enumerate(rows):
        big = _rand(r, K0 + 5, seed=10 + i)  # rows are a column slice of a wider matrix: lda0 > K0, 4-byte aligned only
        a0 = big[:, 1:1 + K0]
        a1 = _rand(r, max(K1, 1), seed=20 + i)
        want = a0.double() @ bt[:, :K0].double().T + bias.double()
        if K1:
            want = want + a1.double() @ bt[:, K0:].double().T
        wants.append(want)
        segs.append(dict(a0=big.cuda()[:, 1:1 + K0], a1=a1.cuda() if K1 else None, T=T))
    outs = ops.gemm_rows(segs, btc[:, :K0], N, K0, N, bt1=btc[:, K0:] if K1 else None, K1=K1, bias=bias.cuda())
    for got, want in zip(outs, wants):
        _close(got, want)


def test_gemm_rows_epilogue_and_segments():
    from carca_replication_amd import ops

    N, K, T = 90, 64, 7
    rows = [70, 131, 5]
    bt, bias, pos, colvec = _rand(N, K, seed=1), _rand(N, seed=2), _rand(T, N, seed=3), _rand(N, seed=4)
    segs, wants = [], []
    for i, r in enumerate(rows):
        a = _rand(r, K, seed=10 + i)
        ids = (torch.rand(r, generator=torch.Generator().manual_seed(20 + i)) > 0.3).int()
        add, gate, rs = _rand(r, N, seed=30 + i), _rand(r, N, seed=40 + i), _rand(r, seed=50 + i)
        v = a.double() @ bt.double().T + bias.double()
        if i == 0:
            v = v + pos.double()[torch.arange(r) % T]
        v = v + add.double() + rs.double()[:, None] * colvec.double()[None, :]
        v = v * torch.where(gate > 0, 1.0, 0.01).double()
        v = v * (ids != 0).double()[:, None]
        wants.append(v)
        segs.append(dict(a0=a.cuda(), ids=ids.cuda(), add=add.cuda(), gate=gate.cuda(), rowscale=rs.cuda(), T=T,
                         add_pos=(i == 0)))
    outs = ops.gemm_rows(segs, bt.cuda(), N, K, 96, bias=bias.cuda(), pos=pos.cuda(), colvec=colvec.cuda(),
                         gate_slope=0.01, mask_rows=True)
    for got, want in zip(outs, wants):
        _close(got[:, :N], want)


@pytest.mark.parametrize("rows,N,K", [([200], 96, 96), ([6400], 90, 540), ([31, 64, 1], 450, 130), ([300], 7, 6),
                                      ([1000, 500], 450, 1024)])
def test_gemm_wgrad(rows, N, K):
    from carca_replication_amd import ops

    segs, want_w, want_b = [], torch.zeros(N, K, dtype=torch.float64), torch.zeros(N, dtype=torch.float64)
    for i, r in enumerate(rows):
        dy, x = _rand(r, N + 3, seed=60 + i), _rand(r, K + 2, seed=70 + i)
        ids = (torch.rand(r, generator=torch.Generator().manual_seed(80 + i)) > 0.25).int()
        m = (ids != 0).double()[:, None]
        want_w += (dy[:, :N].double() * m).T @ x[:, :K].double()
        want_b += (dy[:, :N].double() * m).sum(0)
        segs.append(dict(dy=dy.cuda(), x=x.cuda(), ids=ids.cuda()))
    dw = torch.zeros(N, K + 5, device="cuda")
    db = torch.zeros(N, device="cuda")
    ops.gemm_wgrad(segs, N, K, dw, db, mask_rows=True)
    _close(dw[:, :K], want_w, tol=5e-5)
    _close(db, want_b, tol=5e-5)
    assert float(dw[:, K:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,T,N,K,K1,how", [
    ((700, 333), 7, 130, 700, 6, "dense"),      # ragged last chunks, ragged n / k blocks, ctx columns in the pad columns
    ((640,), 10, 96, 380, 3, "view"),           # K - k0 > 352: the ctx columns get a k block of their own
    ((512, 96), 1, 200, 384, 0, "gather"),      # attribute rows gathered from a table by item id, no second source
    ((950,), 50, 450, 1030, 8, "view"),
])
def test_gemm_wgrad_kernel_choices(gemm_variant, rows, T, N, K, K1, how):
    """The persistent one-block-per-CU kernel (forced by variant 2 whatever the size) against the tiled one and fp64:
    masked rows, [B, T, K] views, table gathers, both k-sources, bias gradient."""
    from carca_replication_amd import ops

    want_w = torch.zeros(N, K + K1, dtype=torch.float64)
    want_b = torch.zeros(N, dtype=torch.float64)
    table = _rand(57, K, seed=5)
    segs = []
    for i, r in enumerate(rows):
        r = (r // T) * T
        dy = _rand(r, N + 2, seed=60 + i)
        ids = torch.randint(0, 57, (r,), generator=torch.Generator().manual_seed(80 + i)).int()
        ids[::5] = 0
        m = (ids != 0).double()[:, None]
        x1 = _rand(r, max(K1, 1), seed=90 + i)
        sg = dict(dy=dy.cuda()[:, :N], ids=ids.cuda())
        if how == "gather":
            x = table[ids.long()]
            sg.update(x=table.cuda(), x_gather=True)
        elif how == "view":
            big = _rand(r // T, T + 2, K, seed=70 + i)
            x = big[:, 1:1 + T].reshape(r, K)
            sg.update(x=big.cuda()[:, 1:1 + T])
        else:
            x = _rand(r, K + 2, seed=70 + i)[:, :K]
            sg.update(x=_rand(r, K + 2, seed=70 + i).cuda()[:, :K])
        if K1:
            sg.update(x1=x1.cuda())
            want_w[:, K:] += (dy[:, :N].double() * m).T @ x1.double()
        want_w[:, :K] += (dy[:, :N].double() * m).T @ x.double()
        want_b += (dy[:, :N].double() * m).sum(0)
        segs.append(sg)
    dw = torch.zeros(N, K + K1 + 3, device="cuda")
    db = torch.zeros(N, device="cuda")
    ops.gemm_wgrad(segs, N, K, dw, db, K1=K1, mask_rows=True)
    _close(dw[:, :K + K1], want_w, tol=5e-5)
    _close(db, want_b, tol=5e-5)
    assert float(dw[:, K + K1:].abs().max()) == 0.0


def test_gemm_wgrad_group_equals_single_launches():
    """carca_gemm_wgrad_group: several independent products in one launch give what one launch each gives."""
    from carca_replication_amd import ops

    shapes = [(640, 96, 96), (1300, 90, 90), (257, 33, 130), (640, 96, 540)]
    ins = [(_rand(r, n + 2, seed=3 * i).cuda(), _rand(r, k + 1, seed=3 * i + 1).cuda()) for i, (r, n, k) in enumerate(shapes)]
    single, grouped = [], []
    wg = ops.WgradGroup()
    for (r, n, k), (dy, x) in zip(shapes, ins):
        dw1, db1 = torch.zeros(n, k, device="cuda"), torch.zeros(n, device="cuda")
        ops.gemm_wgrad([dict(dy=dy[:, :n], x=x[:, :k])], n, k, dw1, db1)
        single.append((dw1, db1))
        dw2, db2 = torch.zeros(n, k, device="cuda"), torch.zeros(n, device="cuda")
        wg.add([dict(dy=dy[:, :n], x=x[:, :k])], n, k, dw2, db2)
        grouped.append((dw2, db2))
    wg.launch()
    for (r, n, k), (dy, x), (dw1, db1), (dw2, db2) in zip(shapes, ins, single, grouped):
        want = dy[:, :n].double().T @ x[:, :k].double()
        _close(dw2, want.cpu(), tol=5e-5)
        _close(db2, dy[:, :n].double().sum(0).cpu(), tol=5e-5)
        _close(dw2, dw1.double().cpu(), tol=1e-5)


def test_pack_weights_fragment_order_layout():
    """CarcaPackDesc.frag16: element (rp, cp) of the padded matrix sits where lane (cp % 16 / 4, rp % 16) of the wave that
    loads tile (rp / 16, cp / 16) reads it -- checked against the index formula of include/carca_hip.h, with head padding."""
    from carca_replication_amd import ops

    d, H = 90, 3
    dpi, dhp, dpo = ops.padded_dims(d, H)
    dh = d // H
    w = torch.arange(d * d, dtype=torch.float32, device="cuda").view(d, d) + 1.0
    pw = ops.PackedWeights([ops.PackItem(w, dpo, dpi, row_heads=(dh, dhp), frag16=True),
                            ops.PackItem(w, dpo, dpi, row_heads=(dh, dhp))], "cuda")
    pw.pack()
    frag = pw.buf[pw.offsets[0]: pw.offsets[0] + dpo * dpi].cpu()
    plain = pw.view(1).cpu()  # row-major head-padded [dpo, dpi]
    rp = torch.arange(dpo).view(-1, 1).expand(dpo, dpi)
    cp = torch.arange(dpi).view(1, -1).expand(dpo, dpi)
    idx = (((rp // 16) * (dpi // 16) + cp // 16) * 64 + ((cp % 16) // 4) * 16 + rp % 16) * 4 + cp % 4
    assert torch.equal(frag[idx.reshape(-1)].view(dpo, dpi), plain)
    assert float(plain[dh, 0]) == 0.0 and float(plain[dhp, 0]) == float(w[dh, 0])  # pad rows are zero, heads shifted
    with pytest.raises(ops.CarcaHipError):
        pw.view(0)


def test_gemm_rows_group_equals_single_launches():
    """carca_gemm_rows_group: independent products in one launch give the bits of their own launches (narrow products
    share a kernel; a wide one in the same call falls back to its own launch)."""
    from carca_replication_amd import ops

    a = _rand(6400, 96, seed=1).cuda()
    b = _rand(6400, 96, seed=2).cuda()
    c = _rand(12800, 96, seed=3).cuda()
    wq, wk, wv = (_rand(96, 96, seed=s).cuda() for s in (4, 5, 6))
    big_w = _rand(540, 96, seed=7).cuda()
    ids = (torch.arange(12800) % 7 != 0).int().cuda()
    add = _rand(6400, 96, seed=8).cuda()
    calls = [dict(segs=[dict(a0=a, add=add)], bt0=wq, N=90, K0=96, out_ld=96),
             dict(segs=[dict(a0=a, a1=b)], bt0=wk, N=90, K0=96, out_ld=96, bt1=wv, K1=96),
             dict(segs=[dict(a0=c[:6400], ids=ids[:6400]), dict(a0=c[6400:], ids=ids[6400:])], bt0=wq, N=90, K0=96,
                  out_ld=96, mask_rows=True),
             dict(segs=[dict(a0=c)], bt0=big_w, N=540, K0=96, out_ld=540)]
    singles = [ops.gemm_rows(**{k: v for k, v in cl.items()}) for cl in calls]
    grouped = ops.gemm_rows_group(calls)
    for s_outs, g_outs in zip(singles, grouped):
        assert len(s_outs) == len(g_outs)
        for s_, g_ in zip(s_outs, g_outs):
            assert torch.equal(s_, g_)
    _close(grouped[0][0][:, :90], (a.cpu().double() @ wq.cpu().double().T + add.cpu().double())[:, :90])
    # five narrow products: more than one launch's worth of descriptors
    five = ops.gemm_rows_group([calls[0]] * 5)
    for o in five:
        assert torch.equal(o[0], singles[0][0])
