"""Eval throughput over a grid of (n_attrs, B, N) around the configurations of SURVEY 8d: a kernel-selection rule that
misfires away from the headline shape shows up as users/s falling when B grows or as a row far below its neighbours."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

L, d, g, H = 50, 90, 450, 3
for n_attrs in [int(a) for a in os.environ.get("ATTRS", "64,512,4096").split(",")]:
    for N in (101, 1001):
        for B in [int(b) for b in os.environ.get("BS", "16,64,128,256,512,1024").split(",")]:
            if B * (L + N) * n_attrs * 4 > 6e9:
                continue
            torch.manual_seed(0)
            model = build_model(dict(d=d, H=H, n_blocks=2), 12102, g, 6, n_attrs, L).cuda().eval()
            profile, target, _ = synth_eval_batch(B, L, N, 12102, n_attrs, 6, seed=1)
            profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
            with torch.no_grad():
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.15:
                    for _ in range(8):
                        model(profile=profile, targets=[target])
                    torch.cuda.synchronize()
                steps = 50
                t0 = time.perf_counter()
                for _ in range(steps):
                    model(profile=profile, targets=[target])
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            F = n_attrs + 6
            flop = B * ((L + N) * (2 * F * g + 2 * (d + g) * d) + 2 * (10 * L * d * d + 4 * L * L * d) + 2 * N * d * d + 4 * L * d * d
                        + 4 * N * L * d + 2 * N * d)
            print(f"n_attrs={n_attrs:5d} N={N:5d} B={B:5d}: {dt * 1e3:8.3f} ms/batch {B / dt:10.0f} users/s {flop / dt / 1e12:6.1f} TFLOP/s",
                  flush=True)
            del model, profile, target
