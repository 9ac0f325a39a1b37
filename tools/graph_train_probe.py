"""Does a whole train step (forward + backward, the optimizer launched eagerly behind it) replay from a hipGraph?
Captures engine.train_step's forward/backward at C2 with torch.cuda.graph, replays it next to the eager loop from the same
initial weights and compares losses step by step; prints both step times.  FOLD=1: the folded-embedding training path."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import engine  # noqa: E402
from carca_replication_amd.optim import Adam  # noqa: E402
from carca_replication_amd.synth import eval_batch  # noqa: E402

c = dict(bench.C2)
c["n_attrs"] = int(os.environ.get("ATTRS", c["n_attrs"]))  # (ATTRS=512: a step whose kernels take ~0.6 ms)
c["B"] = int(os.environ.get("B", c["B"]))
dev = torch.device("cuda")
fold = bool(int(os.environ.get("FOLD", "0")))
L = c["L"]
profile, pos, _ = eval_batch(c["B"], L, L, c["n_items"], c["n_attrs"], c["n_ctx"], seed=4321)
px = profile[0]
o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
o_a = torch.cat([pos[1], pos[1].flip(1)], dim=1)
o_c = torch.cat([pos[2], pos[2]], dim=1)
y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
batch = tuple(t.to(dev) for t in (profile[0], profile[1], profile[2], o_x, o_a, o_c, y_true))


def fresh():
    torch.manual_seed(0)
    m = bench.build_model(c, dev)
    m.train()
    m.fold_embedding(fold, training=fold)
    return m, Adam(m.parameters(), lr=float(os.environ.get("LR", "1e-5")), betas=(0.9, 0.98))


STEPS = 30
model, opt = fresh()
eager = []
for _ in range(6):
    eager.append(engine.train_step(model, opt, batch))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    eager.append(engine.train_step(model, opt, batch))
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / STEPS
eager = [float(x) for x in eager]

model, opt = fresh()
step = engine.GraphedTrainStep(model, opt, batch)
graphed = []
for _ in range(6):
    graphed.append(step(step.inputs).clone())  # (the graph's own input tensors: the eager loop does not copy its batch either)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    graphed.append(step(step.inputs).clone())  # (the graph's own input tensors: the eager loop does not copy its batch either)
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / STEPS
graphed = [float(x) for x in graphed]
worst = max(abs(a - b) / max(abs(a), 1e-9) for a, b in zip(eager, graphed))
print(f"fold={fold} eager {t_eager * 1e3:.3f} ms/step ({c['B'] / t_eager:.0f} users/s)   graphed {t_graph * 1e3:.3f} ms/step "
      f"({c['B'] / t_graph:.0f} users/s)   worst relative loss difference over {len(eager)} steps: {worst:.2e}")
print("eager  ", " ".join(f"{x:.5f}" for x in eager[:8]))
print("graphed", " ".join(f"{x:.5f}" for x in graphed[:8]))
