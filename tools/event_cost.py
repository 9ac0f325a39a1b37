"""What do the dispatch-bound timing events of bench.py cost?  The eval step timed by the wall clock (sync on both sides)
with no events, with the feature GEMM's pair only, and with all four pairs (feature GEMM, scoring kernel, first
SelfAttentionBlock, joint GEMM), interleaved in one process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import ops  # noqa: E402

c = dict(bench.C2)
dev = torch.device("cuda", 0)
model = bench.build_model(c, dev)
_, _, profile, target = bench.build_inputs(c, 1234, dev)
K = int(os.environ.get("STEPS", "300"))
pool = [[ops.HipEvent() for _ in range(8)] for _ in range(K)]


def run(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        if kind != "none":
            hs = [e.handle for e in pool[i]]
            ops.set_fused_events(hs if kind == "all" else hs[:2] + [None] * 6)
        model(profile=profile, targets=[target])
        ops.set_fused_events(None)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


with torch.no_grad():
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(16):
            model(profile=profile, targets=[target])
        torch.cuda.synchronize()
    for rep in range(3):
        print("  ".join(f"{kind}: {run(kind):.4f} ms/step" for kind in ("none", "feat", "all", "none")), flush=True)
