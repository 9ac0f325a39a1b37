"""Time the feats_embed weight-gradient GEMM (dW_f = dq^T [attrs|ctx], as large as the forward GEMM) at C2
train shapes for several row-split targets (tuning key 2)."""
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
slots = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["512", "768", "1024"])]
res = {s: [] for s in slots}
for rnd in range(4):
    for sl in slots:
        lib.carca_set_tuning(2, sl)
        dw = torch.zeros(g, n_attrs + n_ctx, device="cuda")
        db = torch.zeros(g, device="cuda")
        evs = []
        for it in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_wgrad([dict(dy=dzq[:, d:], x=attrs, x1=ctx)], g, n_attrs, dw, db, K1=n_ctx)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        if rnd:
            res[sl] += [a.elapsed_time(b) for a, b in evs]
for sl in slots:
    t = sorted(res[sl])
    med = t[len(t) // 2]
    print(f"slots {sl}: median {med*1e3:.1f} us ({flops/med/1e9:.1f} TF, {flops/med/1e9/157.3*100:.1f}%)  min {t[0]*1e3:.1f} us")
