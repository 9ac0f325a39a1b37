"""A/B the row-GEMM variants on the C2 feature GEMM, interleaved rounds in one process (guide rule 24)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402

B, L, N, d, g, n_attrs, n_ctx, n_items = 128, 50, 101, 90, 450, 4096, int(os.environ.get('NCTX', '6')), 12102
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "2"])]  # variant + 100 * ablate mask
torch.manual_seed(0)
dev = "cuda"
mk = lambda *s: torch.rand(*s, device=dev)  # noqa: E731
p_x = torch.randint(1, n_items, (B, L), device=dev, dtype=torch.int32)
o_x = torch.randint(1, n_items, (B, N), device=dev, dtype=torch.int32)
segs = [(p_x, mk(B, L, n_attrs), mk(B, L, n_ctx), False), (o_x, mk(B, N, n_attrs), mk(B, N, n_ctx), False)]
E, Wf, bf = mk(n_items, d) - 0.5, (mk(g, n_attrs + n_ctx) - 0.5) * 0.03, mk(g) - 0.5
Wj, bj = (mk(d, d + g) - 0.5) * 0.1, mk(d) - 0.5
lib = _lib.load()
flops = 2.0 * B * (L + N) * (n_attrs + n_ctx) * g
ref = None
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        lib.carca_set_tuning(0, v)
        evs = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.set_stage_events({"feat": (e0, e1)})
            outs, zq = ops.embed_fwd(segs, E, Wf, bf, Wj, bj, None, 96)
            evs.append((e0, e1))
        ops.set_stage_events(None)
        torch.cuda.synchronize()
        if rnd > 0:
            res[v] += [a.elapsed_time(b) for a, b in evs]
        if ref is None:
            ref = outs[1].clone()
        else:
            err = float((outs[1] - ref).abs().max())
            assert v >= 100 or err < 1e-3, (v, err)
for v in variants:
    t = sorted(res[v])
    med, mn = t[len(t) // 2], t[0]
    print(f"variant {v}: median {med*1e3:.1f} us ({flops/med/1e9:.1f} TF, {flops/med/1e9/157.3*100:.1f}%)  min {mn*1e3:.1f} us")
