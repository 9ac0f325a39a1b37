"""f4 measurement: the KNN baseline model (knn.py:8-21) at C2 shapes -- HBM-bound row dots.
Algorithmic bytes per launch = B*T*F*4 (target rows, streamed once) + B*F*4 (last profile rows) + B*T*4 (scores).
Several distinct batches are cycled so that no launch finds its rows in the 256 MB infinity cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import ops  # noqa: E402

B, L, T, F, n_items = 128, 50, 101, 4096, 12102
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.rand(n_items, F, device="cuda", generator=g)
NB = 8
batches = []
for i in range(NB):
    p_x = torch.randint(1, n_items, (B, L), device="cuda", generator=g, dtype=torch.int32)
    o_x = torch.randint(1, n_items, (B, T), device="cuda", generator=g, dtype=torch.int32)
    batches.append((p_x, o_x, table[p_x.long()], table[o_x.long()]))
algo = B * T * F * 4 + B * F * 4 + B * T * 4


def timed(fn, reps=40):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % NB)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


dt = timed(lambda i: ops.knn_score(batches[i][2], batches[i][3]))
print(f"dense  [B,T,F] operands: {dt * 1e6:7.1f} us/batch = {B / dt:11.0f} users/s, {algo / dt / 1e9:7.0f} GB/s "
      f"({algo / dt / 8e12 * 100:.0f} % of 8 TB/s)")
dt = timed(lambda i: ops.knn_score(None, None, batches[i][0], batches[i][1], table=table))
print(f"table  gather by id    : {dt * 1e6:7.1f} us/batch = {B / dt:11.0f} users/s, {algo / dt / 1e9:7.0f} GB/s algorithmic "
      f"(198 MB table: rows repeat across candidates and stay in cache)")
ref = (batches[0][2][:, -1:, :] * batches[0][3]).sum(-1)
torch.cuda.synchronize()
dt = timed(lambda i: (batches[i][2][:, -1:, :] * batches[i][3]).sum(-1))
print(f"ATen expression (knn.py:18) on the GPU, for scale: {dt * 1e6:7.1f} us/batch")
err = float((ops.knn_score(batches[0][2], batches[0][3]) - ref).abs().max())
print(f"max |hip - aten| = {err:.2e} on scores of magnitude {float(ref.abs().max()):.0f}")
