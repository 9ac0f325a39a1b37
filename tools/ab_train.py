"""A/B the train step (fwd + bwd + fused Adam at C2, B = 128) under tuning settings, interleaved in ONE process
(boxes differ by a few percent, so separate runs cannot resolve small effects).
usage: ab_train.py "key=value[,key=value...]" "..."      e.g.  ab_train.py "0=0" "0=6" "5=-1"  (empty string = defaults)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import _lib, engine  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402

c = dict(bench.C2) if hasattr(bench, "C2") else dict(B=128, L=50, N=101, d=90, g=450, H=3, n_blocks=2, n_attrs=4096, n_ctx=6, n_items=12102)
settings = sys.argv[1:] or ["", "0=6"]
from tests.model_util import build_model  # noqa: E402

torch.manual_seed(0)
model = build_model(dict(d=c["d"], H=c["H"], n_blocks=c["n_blocks"]), c["n_items"], c["g"], c["n_ctx"], c["n_attrs"], c["L"]).cuda().train()
L = c["L"]
profile, pos, _ = synth_eval_batch(c["B"], L, L, c["n_items"], c["n_attrs"], c["n_ctx"], seed=4321)
px = profile[0]
o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
batch = tuple(t.cuda() for t in (profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                  torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))
from carca_replication_amd.optim import Adam as _Adam  # noqa: E402
opt = _Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
if os.environ.get("FOLD") == "1":  # the opt-in re-associated embedding (CARCA.fold_embedding(True, training=True))
    model.fold_embedding(True, training=True)
lib = _lib.load()
res = {s: [] for s in settings}
for rnd in range(4):
    for s in settings:
        for k in range(8):  # (keys 8+ are modes, not variants)
            lib.carca_set_tuning(k, 0)
        for kv in filter(None, s.split(",")):
            k, v = kv.split("=")
            lib.carca_set_tuning(int(k), int(v))
        for _ in range(3):
            engine.train_step(model, opt, batch, sharded=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            engine.train_step(model, opt, batch, sharded=False)
        torch.cuda.synchronize()
        if rnd:
            res[s].append((time.perf_counter() - t0) / 20)
for s in settings:
    t = sorted(res[s])
    print(f"{s or 'defaults':24s} median {t[len(t)//2]*1e3:.3f} ms/step  min {t[0]*1e3:.3f}")
