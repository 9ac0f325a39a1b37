"""End-to-end A/B of the C2 eval forward (ms per step, wall clock over 200 steps, interleaved, 4 rounds):
feature GEMM as gemm_rows_cu_kernel / gemm_rows_sk_kernel, item-row gather riding in its launch or in its own.
Tuning key 0: 15 = never the stream-K kernel, 8 = never let the gather ride; key 6... (see include/carca_hip.h)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import _lib  # noqa: E402

c = dict(bench.C2)
dev = torch.device("cuda", 0)
model = bench.build_model(c, dev)
_, _, profile, target = bench.build_inputs(c, 1234, dev)
lib = _lib.load()
CONF = {"cu + passenger": (15, 0), "sk, own gather launch": (0, 0), "sk, gather one row per wave (18)": (18, 0), "cu, own gather launch": (158, 0)}
res = {k: [] for k in CONF}
with torch.no_grad():
    for rnd in range(5):
        for name, (v0, _) in CONF.items():
            lib.carca_set_tuning(0, v0)
            for _ in range(30):
                model(profile=profile, targets=[target])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                model(profile=profile, targets=[target])
            torch.cuda.synchronize()
            if rnd:
                res[name].append((time.perf_counter() - t0) / 200)
lib.carca_set_tuning(0, 0)
for name, t in res.items():
    t = sorted(t)
    print("%-24s median %.4f ms/step  min %.4f  -> %.1f k users/s" % (name, 1e3 * t[len(t) // 2], 1e3 * t[0], c["B"] / t[len(t) // 2] / 1e3))
