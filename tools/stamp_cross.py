"""Phase anatomy of the cross-attention scoring kernel from in-kernel stamps (diagnostic; C2 eval shapes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

B, L, N, d, g, H = int(os.environ.get("B", "128")), 50, 101, 90, 450, 3
torch.manual_seed(0)
model = build_model(dict(d=d, H=H, n_blocks=2), 500, g, 6, 64, L).eval().cuda()
dpi, _, _ = ops.padded_dims(d, H)
x = torch.zeros(B, L, dpi, device="cuda")
x[..., :d] = torch.randn(B, L, d, device="cuda")
o = torch.zeros(B, N, dpi, device="cuda")
o[..., :d] = torch.randn(B, N, d, device="cuda")
p_ids = torch.randint(0, 5, (B, L), device="cuda", dtype=torch.int32)
o_ids = torch.randint(1, 5, (B, N), device="cuda", dtype=torch.int32)
cw = model.decoder.weights_struct(x.device, model.norm)
lib = _lib.load()
for _ in range(5):
    ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False)
buf = torch.zeros(2 * B * 16, dtype=torch.int64, device="cuda")
lib.carca_set_debug_buffer(buf.data_ptr())
ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False)
torch.cuda.synchronize()
lib.carca_set_debug_buffer(None)
st = buf.view(2 * B, 16)[:, :5].cpu().double()
st = st[st[:, 0] > 0]
dt = st[:, 1:] - st[:, :-1]
for i, n in enumerate(["A0 load", "A1 final LN", "B K/V", "C targets"]):
    print(f"  {n:12s} median {dt[:, i].median():9.0f}   max {dt[:, i].max():9.0f}")
print(f"  total        median {(st[:, 4] - st[:, 0]).median():9.0f}   workgroups {len(st)}")
