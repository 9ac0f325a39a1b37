"""Timing dissection of the split-precision feature GEMM (csrc/gemm_split.hip): python tools/split_probe.py
The product alone at C2 size (19,328 x (4096 + 6) x 450) under each mode, events around back-to-back launches after a
pre-heat, for the shipped kernel and for its DIAG variants (tuning key 15: timing experiments, wrong results):
1 no global loads / DMA, 4 no MFMAs, 8 no split arithmetic, 16 no barriers, 32 no B fragment reads (sums combine);
VARIANT=21 selects the register-staged kernel instead of the LDS-DMA one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import ops  # noqa: E402

rows, K0, K1, N = 19328, 4096, 6, 450
g = torch.Generator(device="cuda").manual_seed(3)
a = torch.rand(rows, K0, device="cuda", generator=g)
c = torch.rand(rows, K1, device="cuda", generator=g)
w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * 0.036
b = torch.zeros(N, device="cuda")
out = torch.empty(rows, 452, device="cuda")
planes = {}
if os.environ.get("VARIANT"):
    ops.set_tuning(0, int(os.environ["VARIANT"]))


def run():
    ops.gemm_rows([dict(a0=a, a1=c, out=out)], w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b)


def timed(reps=40):
    for _ in range(60):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


diags = [int(x) for x in os.environ.get("DIAGS", "0,64,128,192,1,4,68,5,8,13").split(",")]
print("fp32 path: %.1f us" % timed())
for mode_name, mode in (("bf16x3", 1), ("fp16x2", 2)):
    ops.set_feature_gemm_precision(mode_name)
    pl = ops.split_pack(w, K0, mode)
    ops.split_bind(w[:, :K0], pl, mode, K0)
    for d in diags:
        ops.set_tuning(15, d)
        print("%s diag %2d: %.1f us" % (mode_name, d, timed()), flush=True)
    ops.set_tuning(15, 0)
    ops.set_feature_gemm_precision("fp32")
