"""gemm_rows_sk_kernel: duration of the C2 feature-GEMM launch against the donated K steps (tuning key 4; 0 = the
launcher's own choice; variant 15 = the plain one-tile-per-workgroup kernel).  python tools/sk_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import _lib, ops  # noqa: E402

c = dict(bench.C2)
dev = torch.device("cuda", 0)
model = bench.build_model(c, dev)
_, _, profile, target = bench.build_inputs(c, 1234, dev)
lib = _lib.load()


def run(n=60):
    evs = [[ops.HipEvent() for _ in range(8)] for _ in range(n)]
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            for _ in range(8):
                model(profile=profile, targets=[target])
            torch.cuda.synchronize()
        for e in evs:
            hs = [x.handle for x in e]
            ops.set_fused_events(hs[:2] + [None] * 6)
            model(profile=profile, targets=[target])
            ops.set_fused_events(None)
        torch.cuda.synchronize()
    ms = sorted(e[0].elapsed_ms(e[1]) for e in evs)
    return ms[len(ms) // 2], ms[0]


lib.carca_set_tuning(0, 15)
print("plain kernel: median %.1f us, min %.1f" % tuple(1e3 * v for v in run()))
lib.carca_set_tuning(0, 8)  # (8: the item-row gather in its own launch -- as a passenger workgroup it alone takes ~455 us)
for don in [int(v) for v in os.environ.get("DONS", "1,2,3,4,5,6,7,8,10").split(",")]:
    lib.carca_set_tuning(11, don)
    print("don = %2d: median %.1f us, min %.1f" % ((don,) + tuple(1e3 * v for v in run())))
lib.carca_set_tuning(11, 0)
# (the kernel's timing experiments -- cheap tile alone, takers alone, givers alone: TUNING.md -- were removed from the kernel
# with their switch: a taker waiting for a partial nobody writes hangs the GPU)
lib.carca_set_tuning(11, 0)
lib.carca_set_tuning(0, 0)
