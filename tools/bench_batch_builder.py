"""f1 measurement: evaluation batches (128 users x 50 history slots x 1 + 100 candidates, C2 shapes) built on the
device (DeviceInteractions.eval_batch, ids + context) vs on the host (data.get_test_sequences, one core)."""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import data as D  # noqa: E402
from carca_replication_amd.device_data import DeviceInteractions  # noqa: E402

n_users, n_items, n_ctx, n_attrs, L, N, B = 4096, 12102, 6, 4096, 50, 100, 128
rng = np.random.default_rng(0)
profiles, ctx = {}, {}
for u in range(n_users):
    items = [int(v) for v in rng.integers(1, n_items, size=int(rng.integers(5, 80)))]
    profiles[u] = items
    for it in set(items):
        ctx[(u, it)] = rng.random(n_ctx, dtype=np.float32)
attrs = rng.random((n_items, n_attrs), dtype=np.float32)
log = DeviceInteractions(profiles, ctx, n_items)
users = log.valid_users("test")
batches = [users[i: i + B] for i in range(0, users.numel() - B + 1, B)]
for it in range(3):
    log.eval_batch(batches[0], L, N, "test", seed=it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it, ub in enumerate(batches):
    log.eval_batch(ub, L, N, "test", seed=it)
torch.cuda.synchronize()
dt_dev = (time.perf_counter() - t0) / len(batches)

def host(with_attrs, nb):
    random.seed(0)
    t0 = time.perf_counter()
    for ub in batches[:nb]:
        rows = [D.get_test_sequences(u, profiles[u], L, N, attrs, ctx, "test", True, with_attrs=with_attrs)
                for u in ub.cpu().tolist()]
        [np.stack(c) for c in zip(*rows)]  # default collate
    return (time.perf_counter() - t0) / nb

dt_ids, dt_dense = host(False, 4), host(True, 2)
print(f"device (ids + ctx, one launch):        {dt_dev * 1e6:9.1f} us/batch  = {B / dt_dev:12.0f} users/s")
print(f"host, ids + ctx (vectorised, 1 core):  {dt_ids * 1e6:9.1f} us/batch  = {B / dt_ids:12.0f} users/s")
print(f"host, dense attrs like the reference:  {dt_dense * 1e6:9.1f} us/batch  = {B / dt_dense:12.0f} users/s "
      f"(+ {B * (L + 1 + N) * n_attrs * 4 / 1e6:.0f} MB over PCIe per batch)")
