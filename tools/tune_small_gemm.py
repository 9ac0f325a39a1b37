"""A/B the narrow-output path of carca_gemm_rows (tuning key 0: 0 = 32-column blocks when the grid is small,
1 = always 128 x 96 blocks) on the joint-embedding GEMM and a d-wide backward product."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
lib = _lib.load()
torch.manual_seed(0)
cases = {"joint  rows=19328 K=540 N=90": (19328, 540, 90), "bwd    rows=6400  K=96  N=90": (6400, 96, 90),
         "bwd2   rows=19200 K=96  N=540": (19200, 96, 540)}
for name, (R, K, N) in cases.items():
    a = torch.randn(R, K, device="cuda"); bt = torch.randn(N, K, device="cuda"); bias = torch.randn(N, device="cuda")
    ref = None
    for v in (0, 2, 1, 0, 2, 1):  # 0 = 32-col blocks + 4-deep prefetch ring, 2 = 32-col blocks, 1 = 128x96 blocks
        lib.carca_set_tuning(0, v)
        ts = []
        for it in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            (out,) = ops.gemm_rows([dict(a0=a)], bt, N, K, ((N + 15) // 16) * 16, bias=bias)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ref = out.clone() if ref is None else ref
        assert float((out - ref).abs().max()) < 1e-3
        ts.sort()
        print(f"{name}  variant {v}: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us")
lib.carca_set_tuning(0, 0)
