"""Row GEMMs with a wide output and a SHORT K (dq = de W_jq: 19200 x 90 -> 450 in the backward pass; the feature GEMM of
a data set with few attributes): three K steps and then 35 MB of output -- which kernel writes it fastest?  Tuning key 0:
0 = the launcher's choice, 1 = 128 x 96 tiles (three blocks per CU), 2 = one 384 x 96 block per CU, 7 = 384 x 128."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
for rows, K, N in ((19200, 90, 450), (19328, 70, 450), (38656, 518, 256)):
    a = torch.randn(rows, K + 6, device="cuda")[:, :K]
    w = torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    segs = [dict(a0=a, T=1)]
    want = a.double() @ w.double().T + bias.double()
    for tune in (0, 1, 2, 7, 0, 1):
        lib.carca_set_tuning(0, tune)
        run = lambda: ops.gemm_rows(segs, w, N, K, N, bias=bias)[0]  # noqa: E731
        err = float((run().double() - want).abs().max())
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"rows={rows} K={K} N={N} tune={tune}: {e0.elapsed_time(e1) * 10:7.1f} us per launch (max err {err:.1e})", flush=True)
lib.carca_set_tuning(0, 0)
