"""Phase anatomy of the SelfAttentionBlock kernels (eval: csrc/sa_eval.hip; TRAIN_KERNEL=1: csrc/sa_block.hip) from
in-kernel s_memtime stamps (diagnostic).  Part 0 / part 1 of a user (two workgroups per user at B <= #CUs / 2) apart."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

B, L, d, g, H = int(os.environ.get("B", "128")), 50, 90, 450, 3
torch.manual_seed(0)
model = build_model(dict(d=d, H=H, n_blocks=2), 500, g, 6, 64, L).eval().cuda()
dpi, _, _ = ops.padded_dims(d, H)
x = torch.zeros(B, L, dpi, device="cuda")
x[..., :d] = torch.randn(B, L, d, device="cuda")
ids = torch.randint(0, 5, (B, L), device="cuda", dtype=torch.int32)
if os.environ.get("FULL"):  # every slot real (the longest profile sets the launch's duration at B <= #CUs / 2)
    ids[:] = 3
sw = model.encoder[0].weights_struct(x.device)
lib = _lib.load()
if os.environ.get("TRAIN_KERNEL"):
    lib.carca_set_tuning(6, 1)
for _ in range(5):
    ops.sa_block_fwd(x, ids, sw, d, H, True)
buf = torch.zeros(2 * B * 16, dtype=torch.int64, device="cuda")
lib.carca_set_debug_buffer(buf.data_ptr())
ops.sa_block_fwd(x, ids, sw, d, H, True)
torch.cuda.synchronize()
lib.carca_set_debug_buffer(None)
st = buf.view(2 * B, 16)[:, :8].cpu().double()
st = st[st[:, 0] > 0]
dt = (st[:, 1:] - st[:, :-1])
names = ["A0 load x", "A1 LN1", "B K/V", "C1 attn", "C2 LN2", "C3 ffn1", "C4 ffn2"]
print("median cycles per phase over workgroups (s_memtime ticks = shader cycles); even / odd workgroups (parts 0 / 1):")
for i, n in enumerate(names):
    print(f"  {n:10s} {dt[:, i].median():9.0f}   max {dt[:, i].max():9.0f}    part 0 {dt[0::2, i].median():9.0f}   part 1 {dt[1::2, i].median():9.0f}")
print(f"  total      {(st[:, 7] - st[:, 0]).median():9.0f}   max {(st[:, 7] - st[:, 0]).max():9.0f}")
print(f"  start skew {(st[:, 0].max() - st[:, 0].min()):9.0f}   end skew {(st[:, 7].max() - st[:, 7].min()):9.0f}")
