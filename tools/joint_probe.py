"""Joint GEMM (AllEmbedding.joint_embed, carca.py:89) duration inside the eval pipeline under row-GEMM kernel choices
(tuning key 0): dispatch-bound events of carca_forward, C2 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import _lib, ops  # noqa: E402

c = dict(bench.C2)
dev = torch.device("cuda", 0)
model = bench.build_model(c, dev)
_, _, profile, target = bench.build_inputs(c, 1234, dev)
lib = _lib.load()
want = None
for variant in [int(v) for v in os.environ.get("VARIANTS", "0,9,10,0").split(",")]:
    lib.carca_set_tuning(0, variant % 100)
    lib.carca_set_tuning(15, variant // 100)  # (variant 100 * diag + v: timing experiments of the 80 x 96 kernel, wrong results)
    with torch.no_grad():
        for _ in range(30):
            y = model(profile=profile, targets=[target])
        pool = [[ops.HipEvent() for _ in range(8)] for _ in range(60)]
        for evs in pool:
            ops.set_fused_events([e.handle for e in evs])
            y = model(profile=profile, targets=[target])
            ops.set_fused_events(None)
        torch.cuda.synchronize()
    if want is None:
        want = y.clone()
    ms = sorted(e[6].elapsed_ms(e[7]) for e in pool)
    print(f"variant {variant}: joint GEMM avg {1e3 * sum(ms) / len(ms):.2f} us  median {1e3 * ms[len(ms) // 2]:.2f}  min {1e3 * ms[0]:.2f}"
          f"   max |dy| vs variant 0 {float((y - want).abs().max()):.2e}", flush=True)
lib.carca_set_tuning(0, 0)
lib.carca_set_tuning(15, 0)
