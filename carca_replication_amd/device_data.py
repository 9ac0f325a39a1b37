"""Batch construction on the device (reference: src/data.py:53-192).

`DeviceInteractions` keeps the interaction log in HBM (CSR over users: item ids + the context row of every
interaction) and builds whole evaluation / training batches with one kernel launch each
(csrc/batch_build.hip: leave-one-out windows, left padding, negative sampling, context assignment).  Batches are
ids + context only; pair it with `AllEmbedding.register_attr_table(load_attrs(...))` so that the model gathers the
attribute rows itself.  The deterministic parts equal `data.get_train_sequences / get_test_sequences` (and so the
reference, fixture G5) exactly; the negatives have the reference's distribution but come from a counter-based hash,
reproducible per seed.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib, ops
from ._lib import CarcaHipError

_MODES = ("train", "val", "test")


def split_constants(mode: str, test: bool) -> Tuple[int, int]:
    """(held_out, floor) of pad_profile for a split (data.py:53-74)."""
    if mode not in _MODES:
        raise ValueError(f"Invalid mode: {mode}")
    held_out = {"train": 2 if test else 1, "val": 1 if test else 0, "test": 0}[mode]
    return held_out, _MODES.index(mode) + 1


class DeviceInteractions:
    def __init__(self, profiles: Dict[int, Sequence[int]], ctx: Dict[Tuple[int, int], np.ndarray], n_items: int,
                 device: str = "cuda"):
        """profiles / ctx exactly as data.load_profiles / data.load_ctx return them; n_items = attrs.shape[0] (pad row
        included), the exclusive upper bound of the negatives (data.py:83)."""
        self.user_ids = list(profiles.keys())
        lens = np.array([len(profiles[u]) for u in self.user_ids], dtype=np.int64)
        offs = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        hist = np.zeros(int(offs[-1]), dtype=np.int32)
        n_ctx = int(next(iter(ctx.values())).shape[0]) if ctx else 0
        hctx = np.zeros((int(offs[-1]), n_ctx), dtype=np.float32)
        for i, u in enumerate(self.user_ids):
            items = profiles[u]
            hist[offs[i]: offs[i + 1]] = items
            for j, it in enumerate(items):
                if n_ctx:
                    hctx[offs[i] + j] = ctx[(u, int(it))]
        self.lens = lens
        self.n_items, self.n_ctx = int(n_items), n_ctx
        self.max_len = int(lens.max()) if len(lens) else 0
        self.hist = torch.from_numpy(hist).to(device)
        self.offs = torch.from_numpy(offs).to(device)
        self.hctx = torch.from_numpy(hctx).to(device)
        self.device = self.hist.device

    @classmethod
    def from_arrays(cls, lens: np.ndarray, hist: np.ndarray, hctx: np.ndarray, n_items: int, device: str = "cuda",
                    user_ids: Sequence[int] = None) -> "DeviceInteractions":
        """The same log from its CSR arrays (bulk / synthetic construction without the per-interaction dict walk):
        lens [U] interactions per user, hist [sum lens] item ids in interaction order, hctx [sum lens, n_ctx]."""
        self = cls.__new__(cls)
        lens = np.asarray(lens, dtype=np.int64)
        offs = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        hist, hctx = np.asarray(hist, dtype=np.int32), np.asarray(hctx, dtype=np.float32)
        if hist.shape != (int(offs[-1]),) or hctx.ndim != 2 or hctx.shape[0] != hist.shape[0]:
            raise ValueError("from_arrays: hist must hold sum(lens) ids and hctx one context row per id")
        self.user_ids = list(range(len(lens))) if user_ids is None else list(user_ids)
        self.lens, self.n_items, self.n_ctx = lens, int(n_items), int(hctx.shape[1])
        self.max_len = int(lens.max()) if len(lens) else 0
        self.hist = torch.from_numpy(hist).to(device)
        self.offs = torch.from_numpy(offs).to(device)
        self.hctx = torch.from_numpy(np.ascontiguousarray(hctx)).to(device)
        self.device = self.hist.device
        return self

    def valid_users(self, mode: str, test: bool = True) -> Tensor:
        """Indices (into this log) of the users that have a window in the split (CARCADataset.valid_user_ids)."""
        _, floor = split_constants(mode, test)
        return torch.from_numpy(np.nonzero(self.lens > floor)[0].astype(np.int32)).to(self.device)

    def _check(self, users: Tensor, n_neg: int) -> Tensor:
        if users.device != self.device:
            raise CarcaHipError("user indices must live on the interaction log's device")
        if self.n_items - 1 - self.max_len < n_neg:
            raise CarcaHipError(f"cannot draw {n_neg} distinct negatives outside a history of up to {self.max_len} items "
                                f"from {self.n_items - 1} ids")
        return users.to(torch.int32).contiguous()

    def eval_batch(self, users: Tensor, L: int, N: int, mode: str, test: bool = True, seed: int = 0):
        """-> (p_x [B,L] i32, p_c [B,L,n_ctx], o_x [B,1+N] i32, o_c [B,1+N,n_ctx], y_true [B,1+N] i32)."""
        lib = _lib.load()
        users = self._check(users, N)
        B, nc, dev = users.numel(), self.n_ctx, self.device
        held_out, floor = split_constants(mode, test)
        p_x = torch.empty(B, L, dtype=torch.int32, device=dev)
        p_c = torch.empty(B, L, nc, dtype=torch.float32, device=dev)
        o_x = torch.empty(B, 1 + N, dtype=torch.int32, device=dev)
        o_c = torch.empty(B, 1 + N, nc, dtype=torch.float32, device=dev)
        y = torch.empty(B, 1 + N, dtype=torch.int32, device=dev)
        _lib.check(lib.carca_build_eval_batch(self.hist.data_ptr(), self.offs.data_ptr(), self.hctx.data_ptr(),
                                              users.data_ptr(), len(self.lens), B, L, N, nc, self.n_items, held_out, floor,
                                              int(seed) & (2 ** 64 - 1), p_x.data_ptr(), p_c.data_ptr(), o_x.data_ptr(),
                                              o_c.data_ptr(), y.data_ptr(), ops._stream()), "build_eval_batch")
        return p_x, p_c, o_x, o_c, y

    def train_batch(self, users: Tensor, L: int, test: bool = True, seed: int = 0):
        """-> (p_x [B,L], p_c [B,L,n_ctx], o_x [B,2L] = positives | negatives, o_c [B,2L,n_ctx], y_true [B,2L])."""
        lib = _lib.load()
        users = self._check(users, L)
        B, nc, dev = users.numel(), self.n_ctx, self.device
        held_out, floor = split_constants("train", test)
        p_x = torch.empty(B, L, dtype=torch.int32, device=dev)
        p_c = torch.empty(B, L, nc, dtype=torch.float32, device=dev)
        o_x = torch.empty(B, 2 * L, dtype=torch.int32, device=dev)
        o_c = torch.empty(B, 2 * L, nc, dtype=torch.float32, device=dev)
        y = torch.empty(B, 2 * L, dtype=torch.int32, device=dev)
        _lib.check(lib.carca_build_train_batch(self.hist.data_ptr(), self.offs.data_ptr(), self.hctx.data_ptr(),
                                               users.data_ptr(), len(self.lens), B, L, nc, self.n_items, held_out, floor,
                                               int(seed) & (2 ** 64 - 1), p_x.data_ptr(), p_c.data_ptr(),
                                               o_x.data_ptr(), o_c.data_ptr(), y.data_ptr(), ops._stream()),
                   "build_train_batch")
        return p_x, p_c, o_x, o_c, y


class DeviceLoader:
    """The reference's `DataLoader(CARCADataset(...))` (scripts/training.py:118-163, data.py:211-248) with the dataset
    resident in HBM: iterating yields the batches `train()` / `evaluate()` take -- the reference's 7-tuple
    (p_x, p_a, p_c, o_x, o_a, o_c, y_true) with p_a = o_a = None, ids + context only -- built on the device by
    `DeviceInteractions`.  Pair it with `model.embeds.register_attr_table(load_attrs(...))`: the model gathers the
    attribute rows by id inside its feature GEMM and no dense [B, T, n_attrs] tensor exists anywhere (2.5 MB per user at
    BASELINE config 2).

        log = DeviceInteractions(profiles, ctx, n_items=attrs.shape[0])
        train_loader = DeviceLoader(log, "train", batch_size=128, profile_seq_len=50, shuffle=True)
        val_loader = DeviceLoader(log, "val", batch_size=128, profile_seq_len=50, target_seq_len=100)

    One kernel launch builds `chunk_batches` batches at once (ids + context of a user are ~4 KB: the launch is sized
    for the chip, not for one batch) and the iterator hands out views of that chunk.  shuffle draws a fresh permutation
    of the split's users per epoch from `seed` (torch's device generator); negatives are re-drawn per epoch and chunk
    from the same seed (counter-based hash, reproducible).  `len()` = batches per epoch, short last batch included
    (drop_last=False as in the reference)."""

    def __init__(self, log: DeviceInteractions, mode: str, batch_size: int, profile_seq_len: int,
                 target_seq_len: int = 100, test: bool = True, shuffle: bool = False, seed: int = 0,
                 users: Tensor = None, chunk_batches: int = 32, drop_last: bool = False, order: str = "log"):
        """order="length" (evaluation, shuffle=False): users sorted by the length of their history, longest first.  The
        metrics are sums over users, so the order is free -- and a batch of more users than the GPU has CUs runs the
        scoring kernel as persistent workgroups that take users w, w + #CUs, ...: with neighbours of similar length every
        workgroup gets the same mix (a user's cost is its key tiles; +8 % at B = 1024, tools/k4_balance_probe.py)."""
        split_constants(mode, test)  # (validates mode)
        self.log, self.mode, self.test = log, mode, test
        self.batch_size, self.L, self.N = int(batch_size), int(profile_seq_len), int(target_seq_len)
        self.shuffle, self.seed, self.drop_last = bool(shuffle), int(seed), bool(drop_last)
        self.users = log.valid_users(mode, test) if users is None else users.to(log.device, torch.int32)
        if order not in ("log", "length"):
            raise ValueError(f"order: 'log' or 'length', got {order!r}")
        if order == "length":
            if shuffle:
                raise ValueError("order='length' is for unshuffled (evaluation) loaders")
            lens = torch.from_numpy(log.lens).to(log.device)[self.users.long()]
            self.users = self.users[torch.argsort(lens, descending=True, stable=True)]
        self.chunk_batches = max(1, int(chunk_batches))
        self.epoch = 0
        self._gen = torch.Generator(device=log.device)

    def __len__(self) -> int:
        n = self.users.numel()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    @property
    def n_users(self) -> int:
        n = self.users.numel()
        return n // self.batch_size * self.batch_size if self.drop_last else n

    def __iter__(self):
        users = self.users
        if self.shuffle:
            self._gen.manual_seed(self.seed * 1_000_003 + self.epoch)
            users = users[torch.randperm(users.numel(), device=users.device, generator=self._gen)]
        epoch, self.epoch = self.epoch, self.epoch + 1
        B, n = self.batch_size, self.n_users
        step = B * self.chunk_batches
        for ci, lo in enumerate(range(0, n, step)):
            chunk = users[lo: min(lo + step, n)]
            seed = (self.seed * 0x9E3779B97F4A7C15 + epoch * 0xD1B54A32D192ED03 + ci) & (2 ** 64 - 1)
            if self.mode == "train":
                p_x, p_c, o_x, o_c, y = self.log.train_batch(chunk, self.L, self.test, seed)
            else:
                p_x, p_c, o_x, o_c, y = self.log.eval_batch(chunk, self.L, self.N, self.mode, self.test, seed)
            for b in range(0, chunk.numel(), B):
                s = slice(b, b + B)
                yield p_x[s], None, p_c[s], o_x[s], None, o_c[s], y[s]
