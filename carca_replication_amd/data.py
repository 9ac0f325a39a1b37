"""Host-side batch construction with the reference's contract (src/data.py of r-papso/carca-replication).

Same public names, arguments, return tuples, dtypes and left-padding as the reference so that
`scripts/training.py:112-163` can build its three datasets from this module unchanged; the windows of
`pad_profile` and the seeded sequences are pinned by fixture G5 (tests/test_data_host.py).

What is different: the per-slot Python loop that copies one attribute row at a time (data.py:112-132,
167-187; the true end-to-end bottleneck) is replaced by one numpy gather per
tensor, and `CARCADataset(..., with_attrs=False)` yields ids + context only, for models that keep the
attribute table on the device (`AllEmbedding.register_attr_table`): the 2.5 MB/user dense attribute tensor is
then never built on the host nor shipped over PCIe.
"""
from __future__ import annotations

import pickle
import random
from collections import defaultdict
from typing import Dict, List, Sequence, Tuple

import numpy as np
from torch.utils.data import Dataset

DATA_PATH = "../../data"

_MODES = ("train", "val", "test")


def set_datapath(path: str) -> None:
    global DATA_PATH
    DATA_PATH = path


def load_ctx(ctx_file: str) -> Dict[Tuple[int, int], np.ndarray]:
    """(user, item) -> float32 context vector (data.py:17-25)."""
    with open(f"{DATA_PATH}/{ctx_file}", "rb") as fh:
        raw = pickle.load(fh)
    return {key: np.asarray(val, dtype=np.float32) for key, val in raw.items()}


def load_attrs(attr_file: str) -> np.ndarray:
    """Item attribute matrix with a zero row prepended for the <pad> item 0 (data.py:28-35)."""
    with open(f"{DATA_PATH}/{attr_file}", "rb") as fh:
        attrs = np.asarray(pickle.load(fh), dtype=np.float32)
    return np.vstack([np.zeros((1, attrs.shape[1]), dtype=np.float32), attrs])


def load_profiles(profile_file: str):
    """Lines "user item ..." in interaction order -> (user ids, item ids, {user: [items]}) (data.py:38-50)."""
    users, items = set(), set()
    profiles = defaultdict(list)
    with open(f"{DATA_PATH}/{profile_file}", "r") as fh:
        for line in fh:
            fields = line.strip().split(" ")
            u, it = int(fields[0]), int(fields[1])
            users.add(u)
            items.add(it)
            profiles[u].append(it)
    return list(users), list(items), profiles


def pad_profile(profile: Sequence[int], max_len: int, mode: str, test: bool) -> List[int]:
    """Indices of the profile entries a split may see: the leave-one-out window of data.py:53-74.

    The last index is the entry to predict, the ones before it the (at most max_len) history.  train holds out
    the last 1 (2 when a test split exists) entries, val the last 0 (1), test none; a profile shorter than
    1 / 2 / 3 entries more than that yields no window.
    """
    if mode not in _MODES:
        raise ValueError(f"Invalid mode: {mode}")
    n = len(profile)
    held_out = {"train": 2 if test else 1, "val": 1 if test else 0, "test": 0}[mode]
    floor = _MODES.index(mode) + 1  # shortest profile that still gives this split something: > 1 / > 2 / > 3
    if n <= floor:
        return []
    stop = max(floor, n - held_out)
    start = max(0, n - held_out - max_len - 1)
    return list(range(start, stop))


def sample_negatives(profile: Sequence[int], n_items: int, n: int) -> List[int]:
    """n distinct item ids in [1, n_items-1] that are not in the profile, drawn with python's `random`
    exactly like data.py:77-87 (same call sequence, so a seeded run reproduces the reference's draws)."""
    seen = set(profile)
    picked = set()
    while len(picked) < n:
        cand = random.randint(1, n_items - 1)
        if cand not in picked and cand not in seen:
            picked.add(cand)
    return list(picked)


def _ctx_rows(ctx, user_id: int, items: np.ndarray, width: int) -> np.ndarray:
    out = np.zeros((len(items), width), dtype=np.float32)
    for i, it in enumerate(items):
        out[i] = ctx[(user_id, int(it))]
    return out


def get_train_sequences(user_id, profile, seq_len, attrs, ctx, test, with_attrs: bool = True):
    """One training sample (data.py:90-137): left-padded history p_*, and [positives | negatives] o_* where the
    positive of slot t is the item that followed slot t's item and the negative shares its context."""
    a_len, c_len = attrs.shape[1], next(iter(ctx.values())).shape[0]
    window = pad_profile(profile, seq_len, "train", test)
    negs = sample_negatives(profile, attrs.shape[0], len(window))
    prof = np.asarray(profile, dtype=np.int64)
    n = max(len(window) - 1, 0)  # history slots that have a successor
    hist = prof[window[:-1]] if n else np.zeros(0, np.int64)
    nxt = prof[window[1:]] if n else np.zeros(0, np.int64)
    # the reference walks the window backwards and hands negatives out in that order (data.py:112-117)
    neg = np.asarray(negs[:n][::-1], dtype=np.int64) if n else np.zeros(0, np.int64)

    p_x = np.zeros(seq_len, dtype=np.int32)
    o_x = np.zeros(2 * seq_len, dtype=np.int32)
    p_c = np.zeros((seq_len, c_len), dtype=np.float32)
    o_c = np.zeros((2 * seq_len, c_len), dtype=np.float32)
    lo = seq_len - n
    p_x[lo:] = hist
    o_x[lo:seq_len] = nxt
    o_x[seq_len + lo:] = neg
    if n:
        p_c[lo:] = _ctx_rows(ctx, user_id, hist, c_len)
        nxt_c = _ctx_rows(ctx, user_id, nxt, c_len)
        o_c[lo:seq_len] = nxt_c
        o_c[seq_len + lo:] = nxt_c  # negatives carry the positive's context (data.py:130)
    y_true = np.zeros(2 * seq_len, dtype=np.int32)
    y_true[np.where(p_x > 0)] = 1
    if not with_attrs:
        return p_x, p_c, o_x, o_c, y_true
    return p_x, attrs[p_x], p_c, o_x, attrs[o_x], o_c, y_true


def get_test_sequences(user_id, profile, profile_seq_len, target_seq_len, attrs, ctx, mode, test,
                       with_attrs: bool = True):
    """One evaluation sample (data.py:140-192): history, then candidate 0 = the held-out item followed by
    target_seq_len sampled negatives that all carry the held-out item's context."""
    a_len, c_len = attrs.shape[1], next(iter(ctx.values())).shape[0]
    window = pad_profile(profile, profile_seq_len, mode, test)
    negs = sample_negatives(profile, attrs.shape[0], target_seq_len)
    prof = np.asarray(profile, dtype=np.int64)
    held = int(prof[window[-1]])
    hist = prof[window[:-1]]
    n = len(hist)

    p_x = np.zeros(profile_seq_len, dtype=np.int32)
    p_c = np.zeros((profile_seq_len, c_len), dtype=np.float32)
    if n:
        p_x[profile_seq_len - n:] = hist
        p_c[profile_seq_len - n:] = _ctx_rows(ctx, user_id, hist, c_len)
    o_x = np.zeros(target_seq_len + 1, dtype=np.int32)
    o_x[0] = held
    o_x[1:] = negs
    o_c = np.repeat(ctx[(user_id, held)][None, :].astype(np.float32), target_seq_len + 1, axis=0)
    y_true = np.zeros(target_seq_len + 1, dtype=np.int32)
    y_true[0] = 1
    if not with_attrs:
        return p_x, p_c, o_x, o_c, y_true
    return p_x, attrs[p_x], p_c, o_x, attrs[o_x], o_c, y_true


def get_sequences(user_id, profile, profile_seq_len, target_seq_len, attrs, ctx, mode, test, with_attrs: bool = True):
    if mode == "train":
        return get_train_sequences(user_id, profile, profile_seq_len, attrs, ctx, test, with_attrs)
    return get_test_sequences(user_id, profile, profile_seq_len, target_seq_len, attrs, ctx, mode, test, with_attrs)


class CARCADataset(Dataset):
    """Map-style dataset over the users that have a window in this split (data.py:211-248)."""

    def __init__(self, user_ids, item_ids, profiles, attrs, ctx, profile_seq_len, target_seq_len, mode,
                 test: bool = True, with_attrs: bool = True):
        super().__init__()
        self.user_ids = self.valid_user_ids(profiles, profile_seq_len, mode, test)
        self.item_ids = item_ids
        self.profiles = profiles
        self.attrs = attrs
        self.ctx = ctx
        self.profile_seq_len = profile_seq_len
        self.target_seq_len = target_seq_len
        self.mode = mode
        self.test = test
        self.with_attrs = with_attrs

    def __len__(self) -> int:
        return len(self.user_ids)

    def __getitem__(self, idx):
        uid = self.user_ids[idx]
        return get_sequences(uid, self.profiles[uid], self.profile_seq_len, self.target_seq_len, self.attrs, self.ctx,
                             self.mode, self.test, self.with_attrs)

    def valid_user_ids(self, profiles, seq_len, mode, test):
        return [u for u, prof in profiles.items() if len(pad_profile(prof, seq_len, mode, test)) > 0]
