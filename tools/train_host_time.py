"""How much of the train step is host (Python / ctypes issue) time?  Issue 30 steps without waiting, then wait."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import engine  # noqa: E402
from carca_replication_amd.synth import eval_batch  # noqa: E402

c = bench.C2
from carca_replication_amd import _lib  # noqa: E402
for kv in filter(None, os.environ.get("TUNE", "").split(",")):  # e.g. TUNE="2=256,5=16" (carca_set_tuning keys)
    _lib.load().carca_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
model = bench.build_model(c, "cuda").train()
if os.environ.get("FOLD"):  # the re-associated embedding in both directions (opt-in)
    model.fold_embedding(True, training=True)
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
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(30):
        engine.train_step(model, opt, batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e3 * (t1 - t0) / 30:.3f} ms/step   issue+drain {1e3 * (t2 - t0) / 30:.3f} ms/step")

one = []
for _ in range(10):  # one step issued into an EMPTY queue: pure host time, no back-pressure from the GPU
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.train_step(model, opt, batch)
    one.append(time.perf_counter() - t0)
torch.cuda.synchronize()
one.sort()
print(f"single step into an empty queue: host returns after {1e3 * one[len(one) // 2]:.3f} ms (median)")

if os.environ.get("PROFILE"):
    import cProfile
    import pstats

    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):  # the backward's Python code then runs on this thread
        pr.enable()
        for _ in range(20):
            engine.train_step(model, opt, batch)
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(70)
