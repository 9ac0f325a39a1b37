"""The two per-user eval kernels at the HEADLINE batch (B = 128, two workgroups per user) with every profile at ONE length: what
the fourth slot tile (49..50 slots) costs sa_eval_kernel and cross_fold_kernel -- the batch's longest user sets each launch, and
BASELINE's draws always hold a 49- or 50-slot profile.  Back-to-back launches between two events (launch gaps included).
    python tools/b128_length_probe.py 50 48 34 32"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import ops  # noqa: E402

c = dict(bench.C2, n_attrs=64)
B = int(os.environ.get("B", 128))
device = torch.device("cuda:0")
model = bench.build_model(c, device)
L, N, d, H = c["L"], c["N"], c["d"], c["H"]
dpi, _, _ = ops.padded_dims(d, H)
cw = model.decoder.weights_struct(device, model.norm)
sw = model.encoder[0].weights_struct(device)
gen = torch.Generator(device=device).manual_seed(B)
x = torch.zeros(B, L, dpi, device=device)
x[..., :d] = torch.randn(B, L, d, device=device, generator=gen)
o = torch.zeros(B, N, dpi, device=device)
o[..., :d] = torch.randn(B, N, d, device=device, generator=gen)
o_ids = torch.randint(1, 5, (B, N), device=device, dtype=torch.int32, generator=gen)


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ln in [int(a) for a in sys.argv[1:]] or [50, 48]:
    p_ids = ((torch.arange(L, device=device)[None, :] >= (L - ln)).int() * 7).expand(B, L).contiguous()
    xm = x * (p_ids != 0)[..., None]  # (leading pad rows equal -- zeros -- as inside carca_forward: pads_uniform holds)
    t_sa = timed(lambda: ops.sa_block_fwd(xm, p_ids, sw, d, H, True, pads_uniform=True))
    t_k4 = timed(lambda: ops.cross_score_fwd(xm, p_ids, [(o, o_ids)], cw, d, H, True, False))
    print(f"B = {B}  every profile {ln:2d} slots: SelfAttentionBlock {t_sa:6.2f} us   scoring {t_k4:6.2f} us (back to back, gaps included)")
