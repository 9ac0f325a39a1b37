"""Synthetic Beauty-shaped batches for benchmarks and tools: pure numpy -> torch, no model code.

lengths ~ U{min_len..L} left-padded with 0 (data.py:113,173); ids ~ U{1..n_items-1}; candidate 0 is the positive,
1..N-1 are distinct negatives outside the profile (data.py:77-87); p_a = attrs[p_x], o_a = attrs[o_x]; every candidate
carries the positive's context (data.py:185).  (oracle/carca_oracle.py keeps its own copy for the tests.)"""
from __future__ import annotations

import numpy as np
import torch


def eval_batch(B: int, L: int, N: int, n_items: int, n_attrs: int, n_ctx: int, seed: int = 1234, min_len: int = 3):
    """-> ((p_x [B,L] i32, p_a [B,L,n_attrs], p_c [B,L,n_ctx]), (o_x [B,N] i32, o_a, o_c), attrs table) on the CPU."""
    if n_items - 1 < L + N:
        raise ValueError(f"{n_items - 1} item ids cannot give {N - 1} distinct negatives outside a profile of {L} items")
    rng = np.random.default_rng(seed)
    attrs = rng.random((n_items, n_attrs), dtype=np.float32)
    attrs[0] = 0.0  # pad row (data.py:33-34)
    p_x = np.zeros((B, L), dtype=np.int32)
    o_x = np.zeros((B, N), dtype=np.int32)
    for u in range(B):
        ell = int(rng.integers(min(min_len, L), L + 1))
        p_x[u, L - ell:] = rng.integers(1, n_items, size=ell)
        o_x[u, 0] = rng.integers(1, n_items)
        seen = set(p_x[u].tolist()) | {int(o_x[u, 0])}
        negs = []
        while len(negs) < N - 1:
            cand = int(rng.integers(1, n_items))
            if cand not in seen:
                seen.add(cand)
                negs.append(cand)
        o_x[u, 1:] = negs
    p_c = rng.random((B, L, n_ctx), dtype=np.float32) * (p_x != 0)[..., None]
    o_c = np.repeat(rng.random((B, 1, n_ctx), dtype=np.float32), N, axis=1)
    t = torch.from_numpy
    return (t(p_x), t(attrs[p_x]), t(p_c)), (t(o_x), t(attrs[o_x]), t(o_c)), t(attrs)
