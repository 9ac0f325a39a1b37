"""How the per-user attention kernels (K2 self-attention block, K4 cross-attention scoring) use the MFMA pipes as the
batch grows: back-to-back launches between two events, fraction = algorithmic flops / time / 157.3 TFLOP/s.
At B = 128 there is one workgroup pair per user and nothing to overlap with; at B >> #CUs several users share a CU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

L, N, d, g, H = 50, int(os.environ.get("N", "101")), 90, 450, 3
PEAK = 157.3e12
torch.manual_seed(0)
model = build_model(dict(d=d, H=H, n_blocks=2), 500, g, 6, 64, L).eval().cuda()
dpi, _, _ = ops.padded_dims(d, H)
cw = model.decoder.weights_struct(torch.device("cuda"), model.norm)
sw = model.encoder[0].weights_struct(torch.device("cuda"))
CA = 2 * N * d * d + 4 * L * d * d + 4 * N * L * d + 2 * N * d
SA = 10 * L * d * d + 4 * L * L * d


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


if os.environ.get("TUNE"):  # e.g. TUNE=1: the 16-wave scoring workgroups at every batch size (A/B against the 8-wave ones)
    from carca_replication_amd import _lib
    _lib.load().carca_set_tuning(1, int(os.environ["TUNE"]))
for B in [int(b) for b in os.environ.get("BS", "128,256,512,1024,2048,4096,8192").split(",")]:
    x = torch.zeros(B, L, dpi, device="cuda")
    x[..., :d] = torch.randn(B, L, d, device="cuda")
    o = torch.zeros(B, N, dpi, device="cuda")
    o[..., :d] = torch.randn(B, N, d, device="cuda")
    p_ids = torch.randint(0, 5, (B, L), device="cuda", dtype=torch.int32)
    o_ids = torch.randint(1, 5, (B, N), device="cuda", dtype=torch.int32)
    t4 = timed(lambda: ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False))
    t2 = timed(lambda: ops.sa_block_fwd(x, p_ids, sw, d, H, True))
    print(f"B={B:5d}  K4 {t4 * 1e6:8.1f} us  {B * CA / t4 / PEAK * 100:5.1f} % of fp32 MFMA peak   "
          f"K2 {t2 * 1e6:8.1f} us  {B * SA / t2 / PEAK * 100:5.1f} %", flush=True)
