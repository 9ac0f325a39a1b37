"""Wall-clock stamps (100 MHz) inside gemm_rows_skc_kernel at C2 with BASELINE's profile lengths: per workgroup, the time of
its prologue steps (chunk counts, running sums + clearing, row lists) and of each piece of its stretch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

torch.manual_seed(0)
# other shapes from the environment (B NA D G H NB; K19 = tuning key 19, the compacting kernel's lower bound on K steps)
B_, NA_, D_, G_, H_, NB_ = (int(os.environ.get(k, v)) for k, v in (("B", 128), ("NA", 4096), ("D", 90), ("G", 450), ("H", 3), ("NB", 2)))
if os.environ.get("K19"):
    ops.set_tuning(19, int(os.environ["K19"]))
model = build_model(dict(d=D_, H=H_, n_blocks=NB_), 12102, G_, 6, NA_, 50).cuda().eval()
profile, target, _ = synth_eval_batch(B_, 50, 101, 12102, NA_, 6, seed=1)
profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
lib = _lib.load()
buf = torch.zeros(65536 + 2 * 256 * 16, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(30):
        model(profile=profile, targets=[target])
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(buf.data_ptr())
    if os.environ.get("DIAG"):
        ops.set_tuning(15, int(os.environ["DIAG"]))
    # OV / OVL: the row-block ownership cost of the stretch plan in K steps (teams / lone workgroups; tuning keys 17 / 18)
    if os.environ.get("OV"):
        ops.set_tuning(17, int(os.environ["OV"]) + 1)
    if os.environ.get("OVL"):
        ops.set_tuning(18, int(os.environ["OVL"]) + 1)
    model(profile=profile, targets=[target])
    ops.set_tuning(15, 0)
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(None)
r = buf[65536:65536 + 4096].view(256, 16).cpu().double()
ws = buf[65536 + 4096:].view(256, 16).cpu().double()[:, :12]
print("a workgroup's waves start within %.2f us of each other (mean over workgroups; max %.2f); wave 0 first: %.0f %%" % (
    float((ws.max(1).values - ws.min(1).values).mean()) / 100, float((ws.max(1).values - ws.min(1).values).max()) / 100,
    100 * float((ws.argmin(1) == 0).float().mean())))
print("the gather's workgroup (last one): %.1f us" % ((float(r[255, 1]) - float(r[255, 0])) / 100.0))
act = r[:, 3] > 0
info = r[:, 10:13].clone().long()
r = torch.where(r > 0, r - r[:, :1], torch.full_like(r, -1.0))  # per workgroup, since its own start (the XCDs' clocks differ)
us = lambda x: x / 100.0  # noqa: E731
print("workgroups with a stretch:", int(act.sum()))
if not act.any():
    sys.exit(0)  # (DIAG=1: no ids read, nothing kept)
for k, name in ((0, "start"), (1, "after chunk counts"), (2, "after running sums / clearing"), (3, "after row lists")):
    x = us(r[act, k])
    print(f"{name:32s} mean {x.mean():7.2f} us   max {x.max():7.2f}")
fine = buf[65536 + 4096:].view(256, 16).cpu().double()[:, 12:16] - buf[65536:65536 + 4096].view(256, 16).cpu().double()[:, :1]
for k, name in ((0, "  prologue: arguments read"), (1, "  prologue: id loads issued"), (2, "  prologue: ids in LDS"), (3, "  prologue: chunks counted")):
    x = fine[act, k] / 100.0
    print(f"{name:32s} mean {x.mean():7.2f} us   max {x.max():7.2f}")
for piece in range(1):
    a, b = r[:, 4 + 2 * piece], r[:, 5 + 2 * piece]
    m = act & (b >= 0)
    if m.any():
        print(f"piece {piece}: workgroups {int(m.sum()):3d}  starts mean {us(a[m]).mean():7.2f}  ends mean {us(b[m]).mean():7.2f} max {us(b[m]).max():7.2f}  "
              f"length mean {((b[m] - a[m]) / 100).mean():7.2f} max {((b[m] - a[m]) / 100).max():7.2f}")
last = torch.stack([r[:, 5], r[:, 7], r[:, 9]]).max(dim=0).values
print("last piece ends: mean %.2f  min %.2f  max %.2f us" % (us(last[act]).mean(), us(last[act]).min(), us(last[act]).max()))
for w in (0, 1, 2, 3, 4, 5, 128, 250, 254):
    print(w, [round(float(us(v)), 1) if v >= 0 else None for v in r[w, :10]])

# us per K step by kind of piece: whole row blocks / given (first) pieces / taken (last) pieces, teams and lone workgroups
import collections
acc = collections.defaultdict(list)
raw = buf[65536:65536 + 4096].view(256, 16).cpu().double()
for wg in range(256):
    for pc in range(3):
        v = int(info[wg, pc])
        a, b = raw[wg, 4 + 2 * pc], raw[wg, 5 + 2 * pc]
        if v > 0 and b > a > 0:
            steps, kind = v >> 3, v & 7
            acc[("lone " if kind & 4 else "team ") + {0: "whole", 1: "given", 2: "taken"}[kind & 3]].append((float(b - a) / 100.0, steps))
for k, v in sorted(acc.items()):
    us_, st = sum(x for x, _ in v), sum(s_ for _, s_ in v)
    print(f"{k}: {len(v):3d} pieces, {st / len(v):6.1f} steps each, {us_ / st:.3f} us per step (ends included)")

# when the workgroups END (us since their own start), by what their stretch holds: the plan balances these (gemm.hip, step 3)
ends = collections.defaultdict(list)
for wg in range(256):
    if not bool(act[wg]):
        continue
    kinds = [int(info[wg, pc]) & 7 for pc in range(3) if int(info[wg, pc]) > 0]
    steps = sum(int(info[wg, pc]) >> 3 for pc in range(3) if int(info[wg, pc]) > 0)
    lone = any(k & 4 for k in kinds)
    owns = sum(1 for k in kinds if (k & 3) in (0, 2))
    gives = sum(1 for k in kinds if (k & 3) == 1)
    ends[("lone" if lone else "team") + f" gives {gives} owns {owns}"].append((float(last[wg]) / 100.0, steps))
for k, v in sorted(ends.items()):
    e = [x for x, _ in v]
    print(f"{k}: {len(v):3d} workgroups, {sum(s_ for _, s_ in v) / len(v):6.1f} steps, end mean {sum(e) / len(e):7.2f} us  min {min(e):7.2f}  max {max(e):7.2f}")
