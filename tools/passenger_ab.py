"""A/B of the item-row gather riding in the stream-K feature GEMM's launch (gather_rows_dma, csrc/gemm.hip) against its own
launch (tuning variant 19), interleaved on one box: ms per eval forward at C2 and at n_attrs = 2048 (the shortest product the
stream-K kernel takes: the passenger must still end before the tiles do)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import ops  # noqa: E402
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

for n_attrs in (4096, 2048):
    torch.manual_seed(0)
    model = build_model(dict(d=90, H=3, n_blocks=2), 12102, 450, 6, n_attrs, 50).cuda().eval()
    profile, target, _ = synth_eval_batch(128, 50, 101, 12102, n_attrs, 6, seed=1)
    profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
    res = {0: [], 19: []}
    with torch.no_grad():
        ops.gemm_rows_log(True)
        y0 = model(profile=profile, targets=[target]).clone()
        print(n_attrs, ops.gemm_rows_log())
        ops.gemm_rows_log(False)
        ops.set_tuning(0, 19)
        y1 = model(profile=profile, targets=[target]).clone()
        ops.set_tuning(0, 0)
        assert torch.equal(y0, y1)
        for rep in range(6):
            for v in (0, 19):
                ops.set_tuning(0, v)
                for _ in range(60):
                    model(profile=profile, targets=[target])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    model(profile=profile, targets=[target])
                torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t0) / 200 * 1e3)
        ops.set_tuning(0, 0)
    print(f"n_attrs={n_attrs}: riding {min(res[0]):.4f} ms (median {sorted(res[0])[3]:.4f}), own launch {min(res[19]):.4f} ms "
          f"(median {sorted(res[19])[3]:.4f})", flush=True)
