"""Where does the host time of one train step go?  cProfile over 40 steps (GPU queue drained every step so that
back-pressure does not distort the picture)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import engine  # noqa: E402
from carca_replication_amd.synth import eval_batch  # noqa: E402

c = bench.C2
model = bench.build_model(c, "cuda").train()
L = c["L"]
profile, pos, _ = eval_batch(c["B"], L, L, c["n_items"], c["n_attrs"], c["n_ctx"], seed=4321)
px = profile[0]
o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
batch = tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                  torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))
from carca_replication_amd.optim import Adam as _Adam
opt = _Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
for _ in range(5):
    engine.train_step(model, opt, batch)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)  # backward in THIS thread so that cProfile sees inside it
pr = cProfile.Profile()
N = 40
for _ in range(N):
    torch.cuda.synchronize()
    pr.enable()
    engine.train_step(model, opt, batch)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
print(f"(all times below are totals over {N} steps; divide by {N})")
st.print_stats(70)
st.sort_stats("tottime")
st.print_stats(30)
