"""A/B the feats_embed weight-gradient GEMM (dW_f = dq^T [attrs|ctx], as large as the forward GEMM) at C2 train
shapes: tuning key 0 = 0 (persistent one-block-per-CU kernel) vs 5 (tiled kernel with row splits), interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402

R, g, d, n_attrs, n_ctx = 19200, 450, 90, 4096, 6
torch.manual_seed(0)
dzq = torch.randn(R, d + g, device="cuda")
attrs = torch.rand(R // 50, 50, n_attrs, device="cuda")
ctx = torch.rand(R // 50, 50, n_ctx, device="cuda")
lib = _lib.load()
flops = 2.0 * R * (n_attrs + n_ctx) * g
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "5"])]
res = {v: [] for v in variants}
want = (dzq[:, d:].double().T @ torch.cat([attrs.reshape(R, -1), ctx.reshape(R, -1)], 1).double())
want_b = dzq[:, d:].double().sum(0)
for rnd in range(4):
    for v in variants:
        lib.carca_set_tuning(0, v)
        evs = []
        for it in range(5):
            dw = torch.zeros(g, n_attrs + n_ctx, device="cuda")
            db = torch.zeros(g, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_wgrad([dict(dy=dzq[:, d:], x=attrs, x1=ctx)], g, n_attrs, dw, db, K1=n_ctx)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        err = float((dw.double() - want).abs().max() / want.abs().max())
        errb = float((db.double() - want_b).abs().max() / want_b.abs().max())
        assert err < 2e-5 and errb < 2e-5, (v, err, errb)
        if rnd:
            res[v] += [a.elapsed_time(b) for a, b in evs]
lib.carca_set_tuning(0, 0)
for v in variants:
    t = sorted(res[v])
    med = t[len(t) // 2]
    print(f"variant {v}: median {med*1e3:.1f} us ({flops/med/1e9:.1f} TF, {flops/med/1e9/157.3*100:.1f}%)  min {t[0]*1e3:.1f} us")
