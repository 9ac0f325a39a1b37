"""Interleaved A/B of the C2 eval forward on one GPU: settings given as KEY=VALUE[,KEY=VALUE] groups of carca_set_tuning keys
(and Z=0/1 for modules.USE_Z_TABLE), each measured `reps` times round-robin (pre-heated, 200 steps per measurement) with the
feature GEMM's and the joint GEMM's own durations from events bound to their dispatches.
    python tools/ab_eval.py "" "17=1,18=1" "17=9,18=9" "Z=0" """
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import modules as M  # noqa: E402
from carca_replication_amd import ops  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

B = int(os.environ.get("B", 128))
N = int(os.environ.get("N", 101))
NA = int(os.environ.get("NA", 4096))
D_, G_, H_, NB_ = (int(os.environ.get(k, v)) for k, v in (("D", 90), ("G", 450), ("H", 3), ("NB", 2)))  # other shapes: B N NA D G H NB
torch.manual_seed(0)
model = build_model(dict(d=D_, H=H_, n_blocks=NB_), 12102, G_, 6, NA, 50).cuda().eval()
profile, target, _ = synth_eval_batch(B, 50, N, 12102, NA, 6, seed=1)
profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
settings = sys.argv[1:] or [""]
reps = int(os.environ.get("REPS", 5))


def apply(spec, on):
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        if k == "Z":
            M.USE_Z_TABLE = bool(int(v)) if on else True
        else:
            ops.set_tuning(int(k), int(v) if on else 0)


def measure():
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            for _ in range(8):
                model(profile=profile, targets=[target])
            torch.cuda.synchronize()
        steps = 200
        t0 = time.perf_counter()
        for _ in range(steps):
            model(profile=profile, targets=[target])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        evs = [[ops.HipEvent() for _ in range(8)] for _ in range(16)]
        for e in evs:
            ops.set_fused_events([h.handle for h in e])
            model(profile=profile, targets=[target])
        ops.set_fused_events(None)
        torch.cuda.synchronize()
    avg = lambda i: sum(e[i].elapsed_ms(e[i + 1]) for e in evs) / len(evs)  # noqa: E731
    return dt * 1e3, avg(0), avg(6), avg(4), avg(2)


res = {s: [] for s in settings}
for r in range(reps):
    for s in settings:
        apply(s, True)
        res[s].append(measure())
        apply(s, False)
for s in settings:
    m = [sum(x[i] for x in res[s]) / len(res[s]) for i in range(5)]
    print(f"{s or 'default':24s} step {m[0]:.4f} ms  ({B / m[0] * 1e3:9.0f} users/s)  feature GEMM {m[1] * 1e3:7.1f} us  joint {m[2] * 1e3:6.1f} us  "
          f"SA block {m[3] * 1e3:5.1f} us  scoring {m[4] * 1e3:5.1f} us   steps: " + " ".join(f"{x[0]:.4f}" for x in res[s]), flush=True)
