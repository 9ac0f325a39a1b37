"""Is the CPU restatement (oracle/carca_oracle.py, bench.py's `cpu_baseline`, kind "port") a fair stand-in for the
reference's own CPU path?  Times both on the same C2 tensors in THIS container (the reference checkout does not exist
on the GPU box): reference = src/carca.py CARCA.forward in eval mode under no_grad, exactly what train.py:44 calls.
Run from the repo root: PYTHONDONTWRITEBYTECODE=1 python tools/cpu_port_vs_reference.py"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
from oracle import carca_oracle as O  # noqa: E402

threads = int(os.environ.get("THREADS", os.cpu_count()))
torch.set_num_threads(threads)
B, L, N, d, g, H, n_attrs, n_ctx, n_items = 128, 50, 101, 90, 450, 3, 4096, 6, 12102
cfg = O.CarcaConfig(d=d, H=H, n_blocks=2)
P = O.init_params(cfg, n_items, g, n_ctx, n_attrs, L, seed=0)
profile, target, _ = O.synth_eval_batch(B, L, N, n_items, n_attrs, n_ctx, seed=1234)


def timed(fn, n=12):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


with torch.no_grad():
    t_port = timed(lambda: O.carca_forward(P, cfg, profile, [target], training=False))
    y_port = O.carca_forward(P, cfg, profile, [target], training=False)

sys.path.remove(ROOT)
for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
    del sys.modules[k]
sys.path.insert(0, "/root/reference")
from src import carca as R  # noqa: E402  (the reference)

torch.manual_seed(0)
enc = R.IdentityEncoding()
model = R.CARCA(d=d, p=0.0, emb=R.AllEmbedding(n_items, d, g, n_ctx, n_attrs, enc),
                enc=torch.nn.ModuleList([R.SelfAttentionBlock(d, H, 0.0, True) for _ in range(2)]),
                dec=R.CrossAttentionBlock(d, H, 0.0, True)).eval()
model.load_state_dict({k: v for k, v in P.items()}, strict=True)
with torch.no_grad():
    t_ref = timed(lambda: model(profile=profile, targets=[target]))
    y_ref = model(profile=profile, targets=[target])
print(f"threads {threads}: reference {B / t_ref:8.0f} users/s ({t_ref * 1e3:.1f} ms/batch)   port {B / t_port:8.0f} users/s "
      f"({t_port * 1e3:.1f} ms/batch)   port/reference = {t_ref / t_port:.2f}   max |y_port - y_ref| = "
      f"{float((y_port.reshape(-1) - y_ref.reshape(-1)).abs().max()):.1e}")
