"""Phase stamps of the one-block-per-CU row GEMM (variant 1603): where a wave's cycles go at C2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402

B, L, N, d, g, n_attrs, n_ctx, n_items = 128, 50, 101, 90, 450, 4096, 6, 12102
torch.manual_seed(0)
mk = lambda *s: torch.rand(*s, device="cuda")  # noqa: E731
p_x = torch.randint(1, n_items, (B, L), device="cuda", dtype=torch.int32)
o_x = torch.randint(1, n_items, (B, N), device="cuda", dtype=torch.int32)
segs = [(p_x, mk(B, L, n_attrs), mk(B, L, n_ctx), False), (o_x, mk(B, N, n_attrs), mk(B, N, n_ctx), False)]
E, Wf, bf = mk(n_items, d) - 0.5, (mk(g, n_attrs + n_ctx) - 0.5) * 0.03, mk(g) - 0.5
Wj, bj = (mk(d, d + g) - 0.5) * 0.1, mk(d) - 0.5
lib = _lib.load()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["3"])]
for variant in variants:
    lib.carca_set_tuning(0, variant)
    print("== variant", variant)
    nblk, nw = 255, 12
    buf = torch.zeros(nblk * nw * 4 + 1024, dtype=torch.int64, device="cuda")
    for it in range(3):
        ops.embed_fwd(segs, E, Wf, bf, Wj, bj, None, 96)
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(20):  # sustained load: the clock settles
        if it == 19:
            ops.set_stage_events({"feat": (e0, e1)})
        ops.embed_fwd(segs, E, Wf, bf, Wj, bj, None, 96)
    ops.set_stage_events(None)
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(None)
    ms = e0.elapsed_time(e1)
    print("kernel %.1f us" % (ms * 1e3))
    r = buf[: nblk * nw * 4].view(nblk, nw, 4).double().cpu()
    tot = r[..., 0]
    print("implied shader clock if the loop were the whole kernel: %.2f GHz" % (float(buf[: nblk * nw * 4].view(nblk, nw, 4)[..., 0].double().max()) / (ms * 1e6)))
    print("loop cycles per wave: mean %.0f  min %.0f  max %.0f  (129 steps -> %.0f cyc/step)" % (tot.mean(), tot.min(), tot.max(), tot.mean() / 129))
    for i, name in ((1, "vmcnt(0) before LDS write"), (2, "s_barrier")):
        x = r[..., i]
        print("%-28s mean %.0f cyc/step (%.1f%% of loop)  max-wave %.0f" % (name, x.mean() / 129, 100 * x.sum() / tot.sum(), x.max() / 129))
    print("per-wave-slot mean barrier wait/step:", [round(float(r[:, w, 2].mean() / 129)) for w in range(nw)])
    print("per-wave-slot mean vmcnt wait/step:", [round(float(r[:, w, 1].mean() / 129)) for w in range(nw)])
