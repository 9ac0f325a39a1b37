"""Phase stamps of the fused row chains of the SelfAttentionBlock backward (csrc/row_chain.hip) inside a real train step
at C2: tuning key 9 = 1 stamps the FFN-side chain, 2 the input-side chain (the last launch of the step wins the buffer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import _lib, engine  # noqa: E402
from carca_replication_amd.optim import Adam  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

c = dict(bench.C2)
torch.manual_seed(0)
model = build_model(dict(d=c["d"], H=c["H"], n_blocks=c["n_blocks"]), c["n_items"], c["g"], c["n_ctx"], c["n_attrs"], c["L"]).cuda().train()
L = c["L"]
profile, pos, _ = synth_eval_batch(c["B"], L, L, c["n_items"], c["n_attrs"], c["n_ctx"], seed=4321)
px = profile[0]
o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
batch = tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                  torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))
opt = Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
lib = _lib.load()
for _ in range(5):
    engine.train_step(model, opt, batch)
NAMES = ["tiles in LDS", "first product done", "its result in LDS", "barrier", "products done", "LayerNorm rounds done", "outputs stored", "end"]
DIAGS = [int(v) for v in os.environ.get("DIAGS", "0").split(",")]
for mode, what in [(m + 16 * dg, w + (f" [diag {dg}]" if dg else "")) for dg in DIAGS for m, w in ((1, "FFN side"), (2, "input side"))]:
    nwg = (c["B"] * L + 63) // 64
    nw = 2 * (96 // 16)  # waves per workgroup at dpi = 96
    buf = torch.zeros(nwg * 16 * 8 + 64, dtype=torch.int64, device="cuda")
    lib.carca_set_tuning(9, mode)
    lib.carca_set_debug_buffer(buf.data_ptr())
    engine.train_step(model, opt, batch)
    torch.cuda.synchronize()
    lib.carca_set_debug_buffer(None)
    lib.carca_set_tuning(9, 0)
    st = buf[: nwg * 16 * 8].view(nwg, 16, 8)[:, :nw].double().cpu()
    t0 = st[:, :, 0].min(dim=1, keepdim=True).values  # the workgroup's first wave to start
    rel = st - t0[:, :, None]
    print(what + ": cycles since the workgroup's first wave started; per phase: median over workgroups of the FASTEST / median / SLOWEST wave")
    for i in range(8):
        col = rel[:, :, i]
        if not (col > 0).any():
            continue
        print(f"   [{(['start'] + NAMES)[i]:24s}] {col.min(dim=1).values.median():8.0f} {col.median(dim=1).values.median():8.0f} {col.max(dim=1).values.median():8.0f}")
    w0 = rel[0]
    print("   workgroup 0, per wave (rows) x stamp (columns):")
    for w in range(nw):
        print("     wave %2d (SIMD %d): " % (w, w % 4) + " ".join(f"{w0[w, i]:7.0f}" for i in range(8)))
