"""Workload for tools/attn_pmc.sh: the two per-user attention kernels (K2 SelfAttentionBlock, K4 cross-attention scoring)
alone, at the headline shape (C2: L=50, N=101, d=90, H=3) and BASELINE's profile lengths U{3..L}, for several batch sizes.
Each (kernel, batch) is launched REPS times back to back; the profiler's per-dispatch rows are told apart by grid size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

L, N, d, g, H = 50, 101, 90, 450, 3
REPS = int(os.environ.get("REPS", "40"))
torch.manual_seed(0)
model = build_model(dict(d=d, H=H, n_blocks=2), 500, g, 6, 64, L).eval().cuda()
dpi, _, _ = ops.padded_dims(d, H)
cw = model.decoder.weights_struct(torch.device("cuda"), model.norm)
sw = model.encoder[0].weights_struct(torch.device("cuda"))
if os.environ.get("TUNE7"):  # 2: the per-user scoring kernel at every batch size, 3: the persistent one (A/B)
    from carca_replication_amd import _lib
    _lib.load().carca_set_tuning(7, int(os.environ["TUNE7"]))
for B in [int(b) for b in os.environ.get("BS", "128,1024,4096").split(",")]:
    gen = torch.Generator(device="cuda").manual_seed(B)
    x = torch.zeros(B, L, dpi, device="cuda")
    x[..., :d] = torch.randn(B, L, d, device="cuda", generator=gen)
    o = torch.zeros(B, N, dpi, device="cuda")
    o[..., :d] = torch.randn(B, N, d, device="cuda", generator=gen)
    ln = torch.randint(3, L + 1, (B,), device="cuda", generator=gen)
    if os.environ.get("FULL"):
        ln[:] = L
    p_ids = ((torch.arange(L, device="cuda")[None, :] >= (L - ln)[:, None]).int() * 7).contiguous()
    o_ids = torch.randint(1, 5, (B, N), device="cuda", dtype=torch.int32, generator=gen)
    for _ in range(REPS):
        ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False)
    torch.cuda.synchronize()
    xz = x * (p_ids != 0)[..., None]  # pad rows are zeros, as the masked embedding hands them to the first block
    for _ in range(REPS):
        ops.sa_block_fwd(xz, p_ids, sw, d, H, True, pads_uniform=True)  # (what carca_forward launches in eval mode)
    torch.cuda.synchronize()
print("done")
