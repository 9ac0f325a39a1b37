"""Eval throughput (users/s, one GPU) of the configurations SURVEY.md section 8d names besides C2 -- the parity-test cases of
BASELINE.json -- to see that kernel selection holds up away from the headline shape.  Synthetic data, random weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

CONFIGS = {
    "C2  B=128 d=90  g=450 H=3 n_attrs=4096 N=101": dict(B=128, L=50, N=101, d=90, g=450, H=3, nb=2, n_attrs=4096, n_ctx=6, n_items=12102),
    "C3  B=512 (same per-user dims)              ": dict(B=512, L=50, N=101, d=90, g=450, H=3, nb=2, n_attrs=4096, n_ctx=6, n_items=12102),
    "C4  B=128 d=128 g=640 H=4 n_attrs=4096      ": dict(B=128, L=50, N=101, d=128, g=640, H=4, nb=2, n_attrs=4096, n_ctx=6, n_items=12102),
    "C5  B=128 n_attrs=512 N=1001                ": dict(B=128, L=50, N=1001, d=90, g=450, H=3, nb=2, n_attrs=512, n_ctx=6, n_items=12102),
    "CLI B=256 d=64 g=256 H=2 3 blocks n_attrs=512": dict(B=256, L=50, N=101, d=64, g=256, H=2, nb=3, n_attrs=512, n_ctx=6, n_items=12102),
    "small n_attrs=64                             ": dict(B=128, L=50, N=101, d=90, g=450, H=3, nb=2, n_attrs=64, n_ctx=6, n_items=12102),
}
results = []
only = os.environ.get("ONLY")  # e.g. ONLY=C5 (prefix of the configuration's name) for a rocprofv3 run of one of them
for name, c in CONFIGS.items():
    if only and not name.startswith(only):
        continue
    torch.manual_seed(0)
    model = build_model(dict(d=c["d"], H=c["H"], n_blocks=c["nb"]), c["n_items"], c["g"], c["n_ctx"], c["n_attrs"], c["L"]).cuda().eval()
    profile, target, _ = synth_eval_batch(c["B"], c["L"], c["N"], c["n_items"], c["n_attrs"], c["n_ctx"], seed=1)
    profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:  # pre-heat as bench.py does: the clock needs tens of ms of load to settle
            for _ in range(8):
                model(profile=profile, targets=[target])
            torch.cuda.synchronize()
        steps = 100
        t0 = time.perf_counter()
        for _ in range(steps):
            model(profile=profile, targets=[target])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    F = c["n_attrs"] + c["n_ctx"]
    flop = c["B"] * ((c["L"] + c["N"]) * (2 * F * c["g"] + 2 * (c["d"] + c["g"]) * c["d"]) + c["nb"] * (10 * c["L"] * c["d"] ** 2 + 4 * c["L"] ** 2 * c["d"])
                     + 2 * c["N"] * c["d"] ** 2 + 4 * c["L"] * c["d"] ** 2 + 4 * c["N"] * c["L"] * c["d"] + 2 * c["N"] * c["d"])
    results.append(dict(config=" ".join(name.split()), ms_per_batch=dt * 1e3, users_per_s=c["B"] / dt,
                        algorithmic_tflops=flop / dt / 1e12, **c))
    print(f"{name}: {dt*1e3:7.3f} ms/batch  {c['B']/dt:10.0f} users/s  {flop/dt/1e12:6.1f} TFLOP/s (algorithmic)", flush=True)

if os.environ.get("JSON_OUT"):  # e.g. JSON_OUT=gpurun_out/r03_configs.json (copied to profiles/ by hand)
    import json

    with open(os.environ["JSON_OUT"], "w") as fh:
        json.dump(dict(what="eval forward, one MI355X, inputs resident in HBM, 0.3 s pre-heat + 100 steps (tools/bench_configs.py)",
                       peak_fp32_mfma_tflops=157.3, results=results), fh, indent=1)
