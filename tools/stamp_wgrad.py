"""Phase stamps of the persistent weight-gradient kernel (tuning variant 3) at C2 train shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402

R, g, d, n_attrs, n_ctx = 19200, 450, 90, 4096, 6
torch.manual_seed(0)
dzq = torch.randn(R, d + g, device="cuda")
attrs = torch.rand(R // 50, 50, n_attrs, device="cuda")
ctx = torch.rand(R // 50, 50, n_ctx, device="cuda")
lib = _lib.load()
nblk, nw = 256, 12
buf = torch.zeros(nblk * nw * 4 + 1024, dtype=torch.int64, device="cuda")
dw = torch.zeros(g, n_attrs + n_ctx, device="cuda")
db = torch.zeros(g, device="cuda")
lib.carca_set_tuning(0, 3)
lib.carca_set_debug_buffer(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(10):
    if it == 9:
        e0.record()
    ops.gemm_wgrad([dict(dy=dzq[:, d:], x=attrs, x1=ctx)], g, n_attrs, dw, db, K1=n_ctx)
e1.record()
torch.cuda.synchronize()
lib.carca_set_debug_buffer(None)
lib.carca_set_tuning(0, 0)
ms = e0.elapsed_time(e1)
r = buf[: nblk * nw * 4].view(nblk, nw, 4).double().cpu()
print("launch %.1f us; kernel cycles max %.0f -> %.2f GHz" % (ms * 1e3, r[..., 0].max(), r[..., 0].max() / (ms * 1e6)))
steps = 55 * 600 / 256
print("whole kernel: mean %.0f cycles; loop %.0f (%.0f per chunk, ideal 9216); vmcnt wait %.0f/chunk; barrier wait %.0f/chunk" % (
    r[..., 0].mean(), r[..., 1].mean(), r[..., 1].mean() / steps, r[..., 2].mean() / steps, r[..., 3].mean() / steps))
print("per-wave-slot barrier wait/chunk:", [round(float(r[:, w, 3].mean() / steps)) for w in range(nw)])
print("per-wave-slot vmcnt wait/chunk:", [round(float(r[:, w, 2].mean() / steps)) for w in range(nw)])
print("blocks: kernel cycles min %.0f max %.0f" % (r[:, 0, 0].min(), r[:, 0, 0].max()))
kc = r[:, 0, 0]
order = torch.argsort(kc)
print("kernel cycles by block id (sorted): fastest", [(int(i), int(kc[i])) for i in order[:8]], " slowest", [(int(i), int(kc[i])) for i in order[-8:]])
print("mean by block id mod 8 (XCD):", [int(kc[x::8].mean()) for x in range(8)])
print("mean by block id // 32:", [int(kc[32 * x: 32 * x + 32].mean()) for x in range(8)])
print("loop cycles by block: min %.0f mean %.0f max %.0f" % (r[:, 0, 1].min(), r[:, 0, 1].mean(), r[:, 0, 1].max()))
