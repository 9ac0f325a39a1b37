"""Time the d x d weight-gradient products of one encoder block (rows = B*L = 6400, N = K = 96 padded) for several
row-split targets (tuning key 2) inside a stream of 12 such launches, the way the backward pass issues them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402

R, d = 6400, 90
torch.manual_seed(0)
dy = torch.randn(R, 96, device="cuda")
x = torch.randn(R, 96, device="cuda")
lib = _lib.load()
lib.carca_set_tuning(3, int(os.environ.get("PLAIN", "0")))
lib.carca_set_tuning(4, int(os.environ.get("MINCH", "0")))
slots = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16", "32", "64", "128", "256", "1024"])]
res = {s: [] for s in slots}
for rnd in range(5):
    for sl in slots:
        lib.carca_set_tuning(2, sl)
        dws = [torch.zeros(d, d, device="cuda") for _ in range(12)]
        dbs = [torch.zeros(d, device="cuda") for _ in range(12)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            ops.gemm_wgrad([dict(dy=dy, x=x)], d, d, dws[i], dbs[i])
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[sl].append(e0.elapsed_time(e1) / 12)
lib.carca_set_tuning(2, 0)
lib.carca_set_tuning(3, int(os.environ.get("PLAIN", "0")))
for sl in slots:
    t = sorted(res[sl])
    print(f"slots {sl:5d}: median {t[len(t)//2]*1e3:7.1f} us per launch")
