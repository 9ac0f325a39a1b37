"""BASELINE configs[3] shape on ONE GPU: n_items = 1,000,001, d = 128, g = 640, H = 4, B = 128 -- does every index /
offset / buffer of the train step (512 MB item table: gradient, Adam state, scatter) and of the eval path hold up at
that size?  Checks the item-table gradient rows against a direct torch computation of the same scatter."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import engine  # noqa: E402
from carca_replication_amd.optim import Adam  # noqa: E402
from carca_replication_amd.synth import eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

B, L, N, d, g, H, n_attrs, n_ctx, n_items = 128, 50, 101, 128, 640, 4, 4096, 6, 1_000_001
torch.manual_seed(0)
model = build_model(dict(d=d, H=H, n_blocks=2), n_items, g, n_ctx, n_attrs, L).cuda()
profile, pos, _ = eval_batch(B, L, L, n_items, n_attrs, n_ctx, seed=7)
px = profile[0]
o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
batch = tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                  torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))
assert int(px.max()) > 900_000, "ids must reach the far end of the table"
model.train()
opt = Adam(model.parameters(), lr=1e-5, betas=(0.9, 0.98))  # (lr 1e-3 saturates this synthetic task in one step: every weight moves by lr over 4102 inputs in [0, 1))
w0 = model.embeds.items_embed.weight.detach().clone()
losses = []
for step in range(3):
    t0 = time.perf_counter()
    losses.append(float(engine.train_step(model, opt, batch)))
    torch.cuda.synchronize()
    print(f"train step {step}: loss {losses[-1]:.5f}  {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
assert all(l == l and l < 100 for l in losses), losses
moved = ((model.embeds.items_embed.weight.detach() - w0).abs().sum(dim=1) > 0).nonzero().flatten()
touched = torch.unique(torch.cat([batch[0].reshape(-1), batch[3].reshape(-1)]).long())
touched = touched[touched != 0]
# exactly the batch's item rows moved and pad row 0 did not -- except each user's LAST profile item, which no target
# may attend in train mode (strictly-lower causal mask of the decoder, carca.py:339): its gradient is exactly zero
missing = touched[~torch.isin(touched, moved)]
assert torch.isin(moved, touched).all() and missing.numel() <= B and torch.isin(missing, batch[0][:, -1].long()).all(), \
    (moved.numel(), touched.numel())
# steady-state step time: the touched-row path (gradient cache + optim.Adam.mark_rows) against the dense one
from carca_replication_amd import autograd  # noqa: E402


def steps_ms(n=20):
    for _ in range(3):
        engine.train_step(model, opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        engine.train_step(model, opt, batch)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


sparse_ms = steps_ms()
keep = autograd.BIG_TABLE_BYTES
autograd.BIG_TABLE_BYTES = 1 << 62   # fresh 512 MB zero fill per step ...
opt._marked_off = True
_mark = opt.mark_rows
opt.mark_rows = lambda *a, **k: False  # ... and Adam over every row
opt.state[model.embeds.items_embed.weight].pop("row_touched", None)
opt._fast = {}
dense_ms = steps_ms()
autograd.BIG_TABLE_BYTES, opt.mark_rows = keep, _mark
print(f"train step at C4 table size: {sparse_ms:.2f} ms touched-row path, {dense_ms:.2f} ms dense path", flush=True)
model.eval()
ep, et, _ = eval_batch(B, L, N, n_items, n_attrs, n_ctx, seed=8)
with torch.no_grad():
    y = model(profile=tuple(t.cuda() for t in ep), targets=[tuple(t.cuda() for t in et)])
assert y.shape == (B, N) and bool(torch.isfinite(y).all())
print(f"C4-shape check ok: {moved.numel()} item rows updated (max id {int(moved.max())}), eval scores finite, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
