"""Ad-hoc gradient sweep: random model / batch shapes, train-mode forward + backward on the GPU (plain path, the opt-in
re-associated embedding, both workgroup-per-user layouts) against torch.autograd over the CPU oracle.
Not part of the test suite; run it after touching the backward path:  python tools/sweep_backward.py [seed] [count]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib  # noqa: E402
from carca_replication_amd import modules as M  # noqa: E402
from oracle import carca_oracle as O  # noqa: E402
from tests.model_util import dev, model_from_params  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
lib = _lib.load()
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    d, H = [(64, 1), (64, 2), (64, 4), (90, 1), (90, 2), (90, 3), (128, 2), (128, 4)][int(rng.integers(8))]
    cfg = O.CarcaConfig(d=d, H=H, n_blocks=int(rng.integers(1, 4)), encoding=["identity", "learnable"][int(rng.integers(2))])
    L = int(rng.integers(2, 65))
    B = int([1, 2, 7, 33, 130][int(rng.integers(5))])
    n_attrs = int([7, 64, 513][int(rng.integers(3))])
    n_ctx, g = int(rng.integers(1, 9)), int([32, 250, 450][int(rng.integers(3))])
    n_items = max(int(rng.integers(50, 400)), 2 * L + 10)
    P = O.perturb_params(O.init_params(cfg, n_items, g, n_ctx, n_attrs, L, seed=it), seed=it + 1)
    profile, pos, _ = O.synth_eval_batch(B, L, L, n_items, n_attrs, n_ctx, seed=it, min_len=1)
    px = profile[0]
    neg = (pos[0].flip(1).contiguous() * (px != 0), pos[1].flip(1).contiguous(), pos[2])
    pos = (pos[0] * (px != 0), pos[1], pos[2])
    y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
    o_x = torch.cat([pos[0], neg[0]], dim=1)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    kink = [1.0]
    _lrelu = torch.nn.functional.leaky_relu

    def spy(x, slope):
        kink[0] = min(kink[0], float(x.detach().abs()[px != 0].min()))
        return _lrelu(x, slope)

    torch.nn.functional.leaky_relu = spy
    loss = O.bce_loss(O.carca_forward(Pg, cfg, profile, [pos, neg], training=True), y_true, O.get_mask(o_x))
    loss.backward()
    torch.nn.functional.leaky_relu = _lrelu
    ref = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    errs = []
    for fold, one_wg in ((False, 0), (True, 0), (False, 1)):
        model = model_from_params(P, cfg).train()
        model.fold_embedding(fold, training=fold)
        lib.carca_set_tuning(1, one_wg)
        lg = M.BinaryCrossEntropy()(model(profile=dev(profile), targets=[dev(pos), dev(neg)]), y_true.cuda(),
                                    M.get_mask(o_x.cuda()))
        lg.backward()
        lib.carca_set_tuning(1, 0)
        e = abs(float(lg) - float(loss))
        for name, prm in model.named_parameters():
            scale = float(ref[name].abs().max())
            # (a tensor whose true gradient is 0 -- key biases: softmax ignores them -- holds round-off only: absolute floor)
            en = max(float((prm.grad.cpu() - ref[name]).abs().max()) - 1e-7, 0.0) / max(scale, 1e-5)
            if en > 1e-4:
                print(f"      {name}: err {float((prm.grad.cpu() - ref[name]).abs().max()):.2e} scale {scale:.2e}")
            e = max(e, en)
        errs.append(e)
    flag = "" if max(errs) < 1e-4 else "   <-- FAIL"
    if flag and kink[0] < 1e-5:
        # a LeakyReLU pre-activation (valid row) within fp32 round-off of 0: another summation order lands on the other
        # side and the slope flips (1 vs 0.01) -- a discontinuity of the function, not a kernel error
        flag = f"   <-- skipped: smallest |ffn_1 pre-activation| on a valid row is {kink[0]:.1e}"
        errs = [0.0, 0.0, 0.0]
    worst = max(worst, max(errs))
    print(f"{it:3d} d={d} H={H} blocks={cfg.n_blocks} {cfg.encoding} B={B} L={L} n_attrs={n_attrs} n_ctx={n_ctx} g={g}: "
          f"worst relative gradient error plain {errs[0]:.1e}  folded {errs[1]:.1e}  one-wg {errs[2]:.1e}{flag}", flush=True)
print("worst", worst)
