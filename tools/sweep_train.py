"""Train-step throughput (fwd + bwd + Adam, p = 0) over (n_attrs, B): the companion of sweep_throughput.py for the kernel
selection rules of the backward pass."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import engine  # noqa: E402
from carca_replication_amd.optim import Adam  # noqa: E402
from carca_replication_amd.synth import eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

L, d, g, H = 50, 90, 450, 3
for n_attrs in [int(a) for a in os.environ.get("ATTRS", "64,512,4096").split(",")]:
    for B in [int(b) for b in os.environ.get("BS", "16,64,128,256,512").split(",")]:
        torch.manual_seed(0)
        model = build_model(dict(d=d, H=H, n_blocks=2), 12102, g, 6, n_attrs, L).cuda().train()
        profile, pos, _ = eval_batch(B, L, L, 12102, n_attrs, 6, seed=4321)
        px = profile[0]
        o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
        o_a = torch.cat([pos[1], pos[1].flip(1)], dim=1)
        o_c = torch.cat([pos[2], pos[2]], dim=1)
        y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
        batch = tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, o_a, o_c, y_true))
        opt = Adam(model.parameters(), lr=1e-5, betas=(0.9, 0.98))
        for _ in range(8):
            engine.train_step(model, opt, batch)
        torch.cuda.synchronize()
        steps = 30
        t0 = time.perf_counter()
        for _ in range(steps):
            engine.train_step(model, opt, batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"n_attrs={n_attrs:5d} B={B:4d}: {dt * 1e3:7.3f} ms/step ({(t1 - t0) / steps * 1e3:6.3f} host)  {B / dt:9.0f} users/s", flush=True)
        del model, batch, opt
