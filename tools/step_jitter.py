"""Per-step GPU timeline of the bench loop: finds host-side hiccups inside the timed region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
c = dict(bench.C2)
dev = torch.device("cuda", 0)
model = bench.build_model(c, dev)
_, _, profile, target = bench.build_inputs(c, 1234, dev)
from carca_replication_amd import ops
N = int(os.environ.get("STEPS", "60"))
with torch.no_grad():
    for _ in range(10):
        model(profile=profile, targets=[target])
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    host = []
    evs[0].record()
    for i in range(N):
        t0 = time.perf_counter()
        ev = {"feat": (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)),
              "cross": (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))}
        ops.set_stage_events(ev)
        model(profile=profile, targets=[target])
        ops.set_stage_events(None)
        evs[i + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print("gpu ms per step:", " ".join(f"{g:.2f}" for g in gpu))
print("host ms per step:", " ".join(f"{h*1e3:.2f}" for h in host))
