"""The persistent scoring kernel (cross_stream_kernel) alone at B users, every profile at the SAME length, for a list of lengths:
what a key tile costs (48 -> 49 slots adds the fourth tile) -- the upper bound of what a cheap last key tile could return.
    B=4096 python tools/k4_length_probe.py 50 48 34 32 18 16"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

c = dict(bench.C2)
B = int(os.environ.get("B", 4096))
device = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(dict(d=c["d"], H=c["H"], n_blocks=2), c["n_items"], c["g"], c["n_ctx"], 64, c["L"]).cuda().eval()
L, N, d, H = c["L"], c["N"], c["d"], c["H"]
dpi, _, _ = ops.padded_dims(d, H)
cw = model.decoder.weights_struct(device, model.norm)
ca = bench.flops_per_user(c)["ca"]
gen = torch.Generator(device=device).manual_seed(B)
x = torch.zeros(B, L, dpi, device=device)
x[..., :d] = torch.randn(B, L, d, device=device, generator=gen)
o = torch.zeros(B, N, dpi, device=device)
o[..., :d] = torch.randn(B, N, d, device=device, generator=gen)
o_ids = torch.randint(1, 5, (B, N), device=device, dtype=torch.int32, generator=gen)
for ln in [int(a) for a in sys.argv[1:]] or [50, 48]:
    p_ids = ((torch.arange(L, device=device)[None, :] >= (L - ln)).int() * 7).expand(B, L).contiguous()
    run = lambda: ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False)  # noqa: E731
    for _ in range(8):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"B = {B}  every profile {ln:2d} slots: {ms * 1e3:7.1f} us  = {B * ca / (ms * 1e-3) / 1e12 / 157.3:.3f} of the fp32 MFMA peak on algorithmic flops (L = 50 counts)")
