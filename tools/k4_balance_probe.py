"""How much of cross_stream_kernel's time at B > #CUs is load imbalance between the persistent workgroups?
Same multiset of profile lengths (BASELINE's U{3..50}), three orders: as drawn (workgroup w gets users w, w + 256, ...: a
random mix), dealt (sorted by key tiles, then dealt round-robin: every workgroup gets the same mix) and adversarial
(sorted so that workgroup w gets similar lengths: the worst case).  python tools/k4_balance_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402


def make(d, H, L):
    torch.manual_seed(0)
    model = build_model(dict(d=d, H=H, n_blocks=1), 500, 64, 6, 64, L).eval().cuda()
    dpi, _, _ = ops.padded_dims(d, H)
    return model, dpi, model.decoder.weights_struct(torch.device("cuda"), model.norm)


def timed(fn, reps=30):
    """seconds per call: back-to-back launches between two events"""
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


d, H, L, N = 90, 3, 50, 101
model, dpi, cw = make(d, H, L)
CA = 2 * N * d * d + 4 * L * d * d + 4 * N * L * d + 2 * N * d
for B in (1024, 4096):
    gen = torch.Generator(device="cuda").manual_seed(B)
    x = torch.zeros(B, L, dpi, device="cuda")
    x[..., :d] = torch.randn(B, L, d, device="cuda", generator=gen)
    o = torch.zeros(B, N, dpi, device="cuda")
    o[..., :d] = torch.randn(B, N, d, device="cuda", generator=gen)
    o_ids = torch.randint(1, 5, (B, N), device="cuda", dtype=torch.int32, generator=gen)
    ln = torch.randint(3, L + 1, (B,), device="cuda", generator=gen)
    srt, _ = torch.sort(ln, descending=True)
    nwg = 256
    orders = {"as drawn": ln, "dealt (sorted, round-robin)": srt,
              "adversarial (workgroup w holds neighbours of the sorted list)": srt.view(nwg, B // nwg).t().reshape(-1)}
    for name, lens in orders.items():
        p_ids = ((torch.arange(L, device="cuda")[None, :] >= (L - lens)[:, None]).int() * 7).contiguous()
        t = timed(lambda: ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False), 40)
        print(f"B={B:5d} {name:64s} {1e6 * t:7.1f} us  {B * CA / t / 157.3e12:6.1%}")
