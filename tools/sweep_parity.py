"""Ad-hoc parity sweep: random model / batch shapes through the HIP forward (+ backward for a subset) against the oracle.
Not part of the test suite (takes a minute); run it after touching kernel-selection logic."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import carca_oracle as O  # noqa: E402
from tests.model_util import dev, model_from_params  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    d, H = [(64, 1), (64, 2), (64, 4), (90, 1), (90, 2), (90, 3), (128, 2), (128, 4)][int(rng.integers(8))]
    cfg = O.CarcaConfig(d=d, H=H, n_blocks=int(rng.integers(1, 4)), encoding=["identity", "learnable"][int(rng.integers(2))],
                        embedding=["all", "all", "attrctx", "id"][int(rng.integers(4))],
                        decoder=["ca", "ca", "dot"][int(rng.integers(3))])
    L = int(rng.integers(1, 65))
    N = int(rng.integers(1, 230))
    B = int([1, 2, 7, 64, 128, 129, 300, 520, 700][int(rng.integers(9))])  # (>= 512: the 8-wave scoring workgroups)
    n_attrs = int([7, 64, 513, 2048][int(rng.integers(4))])
    n_ctx, g, n_items = int(rng.integers(1, 9)), int([32, 250, 450][int(rng.integers(3))]), int(rng.integers(50, 400))
    n_items = max(n_items, L + N + 10)
    if B * (L + N) * n_attrs > 3e8:
        B = 7
    P = O.perturb_params(O.init_params(cfg, n_items, g, n_ctx, n_attrs, max(L, 1), seed=it), seed=it + 1)
    profile, target, _ = O.synth_eval_batch(B, L, N, n_items, n_attrs, n_ctx, seed=it, min_len=min(3, L))
    want = O.carca_forward(P, cfg, profile, [target], training=False)
    model = model_from_params(P, cfg).eval() if cfg.embedding == "all" and cfg.decoder == "ca" else None
    if model is None:
        from tests.model_util import build_model
        c = dict(d=d, H=H, n_blocks=cfg.n_blocks, encoding=cfg.encoding, embedding=cfg.embedding, decoder=cfg.decoder)
        model = build_model(c, n_items, g, n_ctx, n_attrs, L)
        model.load_state_dict(P, strict=True)
        model = model.cuda().eval()
    with torch.no_grad():
        got = model(profile=dev(profile), targets=[dev(target)]).cpu()
    err = float((got.reshape(want.shape) - want).abs().max())
    worst = max(worst, err)
    flag = "" if err < 1e-4 else "   <-- FAIL"
    print(f"{it:3d} d={d} H={H} blocks={cfg.n_blocks} {cfg.embedding}/{cfg.decoder}/{cfg.encoding} B={B} L={L} N={N} "
          f"n_attrs={n_attrs} n_ctx={n_ctx} g={g}: max err {err:.2e}{flag}")
print("worst", worst)
