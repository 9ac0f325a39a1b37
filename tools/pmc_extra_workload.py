"""Workloads for the PMC passes that profiles/ did not hold yet: python tools/pmc_extra_workload.py MODE
  table  the C2 eval forward over ids-only batches: attribute rows gathered by id INSIDE the feature GEMM
         (AllEmbedding.register_attr_table; the "fused embedding gather" of the north star)
  train  the C2 train step (fwd + bwd + Adam): gemm_wgrad_cu_kernel (dW of feats_embed) and its neighbours
  knn    the KNN baseline's scoring kernel (knn.py:8-21) over 8 distinct C2-shaped batches (212 MB each: beyond the MALL)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from carca_replication_amd import engine, ops  # noqa: E402
from carca_replication_amd.optim import Adam  # noqa: E402

mode = sys.argv[1]
steps = int(os.environ.get("STEPS", "30"))
c = dict(bench.C2)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if mode == "table":
    model = bench.build_model(c, dev)
    _, _, profile, target = bench.build_inputs(c, 1234, dev)
    import numpy as np

    tab = torch.from_numpy(np.random.default_rng(99).random((c["n_items"], c["n_attrs"]), dtype=np.float32))
    tab[0] = 0
    model.embeds.register_attr_table(tab.to(dev))
    pf, tg = (profile[0], None, profile[2]), (target[0], None, target[2])
    with torch.no_grad():
        for _ in range(steps + 10):
            model(profile=pf, targets=[tg])
elif mode == "train":
    from carca_replication_amd.synth import eval_batch

    model = bench.build_model(c, dev).train()
    L = c["L"]
    profile, pos, _ = eval_batch(c["B"], L, L, c["n_items"], c["n_attrs"], c["n_ctx"], seed=4321)
    px = profile[0]
    o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
    batch = tuple(t.to(dev) for t in (profile[0], profile[1], profile[2], o_x, torch.cat([pos[1], pos[1].flip(1)], 1),
                                      torch.cat([pos[2], pos[2]], 1), torch.cat([(px != 0).int(), torch.zeros_like(px)], 1)))
    opt = Adam(model.parameters(), lr=1e-5, betas=(0.9, 0.98))
    for _ in range(steps + 6):
        engine.train_step(model, opt, batch)
elif mode == "knn":
    B, L, T, F = 128, 50, 101, 4096
    gen = torch.Generator(device="cuda").manual_seed(0)
    sets = [(torch.rand(B, L, F, device="cuda", generator=gen), torch.rand(B, T, F, device="cuda", generator=gen)) for _ in range(8)]
    for i in range(steps + 8):
        p_a, o_a = sets[i % 8]
        ops.knn_score(p_a, o_a)
else:
    raise SystemExit("mode: table | train | knn")
torch.cuda.synchronize()
