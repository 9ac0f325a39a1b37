"""Wall-clock stamps (100 MHz) inside gemm_rows_n96s_kernel at C5 (or B / N / NA from the environment): per workgroup the time of
its prologue, of each block's pinned steps, its last (generic) steps and its epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

B, N, NA = (int(os.environ.get(k, v)) for k, v in (("B", 128), ("N", 1001), ("NA", 512)))
torch.manual_seed(0)
model = build_model(dict(d=90, H=3, n_blocks=2), 12102, 450, 6, NA, 50).cuda().eval()
profile, target, _ = synth_eval_batch(B, 50, N, 12102, NA, 6, seed=1)
profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
lib = _lib.load()
buf = torch.zeros(65536 + 256 * 64, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(30):
        model(profile=profile, targets=[target])
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(buf.data_ptr())
    if os.environ.get("DIAG"):
        ops.set_tuning(5, int(os.environ["DIAG"]))
    model(profile=profile, targets=[target])
    ops.set_tuning(5, 0)
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(None)
r = buf[65536:].view(256, 64).cpu().double()
act = r[:, 1] > 0
r = torch.where(r > 0, (r - r[:, :1]) / 100.0, torch.full_like(r, float("nan")))
r = r[act]
print("workgroups:", int(act.sum()))
mean = lambda x: float(x[~x.isnan()].mean()) if (~x.isnan()).any() else float("nan")  # noqa: E731
print("prologue done        %7.2f us" % mean(r[:, 1]))
for b in range(4):
    sb = 2 + 14 * b
    if torch.isnan(r[:, sb]).all():
        break
    print("block %d: start %7.2f | pinned steps done +%6.2f | last steps +%6.2f +%6.2f | epilogue +%6.2f | n = %d" % (
        b, mean(r[:, sb]), mean(r[:, sb + 1] - r[:, sb]), mean(r[:, sb + 2] - r[:, sb + 1]), mean(r[:, sb + 3] - r[:, sb + 2]),
        mean(r[:, sb + 13] - r[:, sb + 12]), int((~torch.isnan(r[:, sb])).sum())))
last = torch.nan_to_num(r, nan=0.0).max(1).values
print("workgroup end: mean %.2f us, max %.2f, min %.2f" % (float(last.mean()), float(last.max()), float(last.min())))
