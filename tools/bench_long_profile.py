"""Eval forward and train step at C2's dimensions with profiles of 100 slots (carca_replication_amd/long_profile.py: the
composed path beyond the fused kernels' 64 slots), beside the same model at L = 50 on the fused path."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import engine  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

for L in (50, 100):
    torch.manual_seed(0)
    model = build_model(dict(d=90, H=3, n_blocks=2), 12102, 450, 6, 4096, L).cuda().eval()
    profile, target, _ = synth_eval_batch(128, L, 101, 12102, 4096, 6, seed=1)
    profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
    with torch.no_grad():
        for _ in range(20):
            model(profile=profile, targets=[target])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            model(profile=profile, targets=[target])
        torch.cuda.synchronize()
    ev = (time.perf_counter() - t0) / 50 * 1e3
    model.train()
    prof, pos, _ = synth_eval_batch(128, L, L, 12102, 4096, 6, seed=2)
    px = prof[0]
    o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
    batch = tuple(t.cuda() for t in (prof[0], prof[1], prof[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                      torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
    for _ in range(5):
        engine.train_step(model, opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        engine.train_step(model, opt, batch)
    torch.cuda.synchronize()
    tr = (time.perf_counter() - t0) / 20 * 1e3
    print(f"L = {L:3d}: eval forward {ev:7.3f} ms per 128 users ({128 / ev:6.1f} k users/s), eager train step {tr:7.3f} ms")
