"""A/B of the GRAPHED train step (engine.GraphedTrainStep at C2, B = 128) under settings of the backward's two-stream
split (autograd.SPLIT_EMBED_BWD / SPLIT_SIDE_CUS / SPLIT_MAIN_TARGET_USERS), interleaved in one process.
usage: ab_train_graph.py "split=0" "split=1,side=128,frac=0.0" "tail=0" "tail=1,tfrac=0.3,tmain=128,table=1" ...   (+ "t<key>=<value>" = tuning keys)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import _lib, autograd, engine  # noqa: E402
from carca_replication_amd.optim import Adam  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

c = dict(bench.C2)
settings = sys.argv[1:] or ["split=0", "split=1"]
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
res = {s: [] for s in settings}
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for s in settings:
        autograd.SPLIT_EMBED_BWD, autograd.SPLIT_SIDE_CUS, autograd.SPLIT_MAIN_TARGET_USERS = "graph", 128, 0.25
        autograd.SPLIT_MAIN_CUS = 256
        autograd.SPLIT_TAIL_ON_SIDE, autograd.SPLIT_TABLE_STREAM = False, False
        autograd.SPLIT_TAIL_MAIN_CUS, autograd.SPLIT_TAIL_MAIN_TARGET_USERS = 128, 0.35
        for k in range(8):
            lib.carca_set_tuning(k, 0)
        for kv in filter(None, s.split(",")):
            k, v = kv.split("=")
            if k == "split":
                autograd.SPLIT_EMBED_BWD = "graph" if v == "graph" else bool(int(v))
            elif k == "side":
                autograd.SPLIT_SIDE_CUS = int(v)
            elif k == "main":
                autograd.SPLIT_MAIN_CUS = int(v)
            elif k == "frac":
                autograd.SPLIT_MAIN_TARGET_USERS = float(v)
            elif k == "tail":       # round 5: the weight gradients that feed nothing on the second stream's tail
                autograd.SPLIT_TAIL_ON_SIDE = bool(int(v))
            elif k == "table":      # ... and the row tables on a third stream
                autograd.SPLIT_TABLE_STREAM = bool(int(v))
            elif k == "tmain":
                autograd.SPLIT_TAIL_MAIN_CUS = int(v)
            elif k == "tfrac":
                autograd.SPLIT_TAIL_MAIN_TARGET_USERS = float(v)
            elif k.startswith("t"):
                lib.carca_set_tuning(int(k[1:]), int(v))
        step = engine.GraphedTrainStep(model, opt, batch)
        for _ in range(40):
            step(step.inputs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            loss = step(step.inputs)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[s].append(e0.elapsed_time(e1) / 100)
        del step
print("last loss %.6f" % float(loss))
for s in settings:
    t = sorted(res[s])
    print(f"{s:32s} median {t[len(t)//2]:.4f} ms/step  min {t[0]:.4f}")
