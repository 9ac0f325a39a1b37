"""Host-side batch construction (carca_replication_amd/data.py) against fixture G5, captured from the reference's
src/data.py: the pad_profile truth table and seeded train / test sequences."""
import random

import numpy as np
import pytest

from carca_replication_amd import data as D
from tests.golden_util import load


def test_pad_profile_truth_table():
    fx = load("g5_data")
    for key, want in fx.outs.items():
        if not key.startswith("pad/"):
            continue
        _, mode, test, n = key.split("/")
        got = D.pad_profile(list(range(100, 100 + int(n))), 5, mode, bool(int(test)))
        assert got == [int(v) for v in np.asarray(want)], key
    with pytest.raises(ValueError):
        D.pad_profile([1, 2, 3], 5, "dev", True)


def _case(fx):
    attrs = np.asarray(fx.ins["attrs"])
    profile = [int(v) for v in np.asarray(fx.ins["profile"])]
    user = int(np.asarray(fx.ins["user"]))
    items, vals = np.asarray(fx.ins["ctx_items"]), np.asarray(fx.ins["ctx_vals"])
    ctx = {(user, int(it)): vals[i].astype(np.float32) for i, it in enumerate(items)}
    return attrs, profile, user, ctx


def test_seeded_train_and_test_sequences_equal_reference():
    fx = load("g5_data")
    attrs, profile, user, ctx = _case(fx)
    L = int(fx.dim["L"])
    names = ("p_x", "p_a", "p_c", "o_x", "o_a", "o_c", "y_true")
    random.seed(123)
    got = D.get_train_sequences(user, profile, L, attrs, ctx, True)
    for nm, g in zip(names, got):
        want = np.asarray(fx.outs["train/" + nm])
        assert g.dtype == want.dtype and g.shape == want.shape and np.array_equal(g, want), nm
    random.seed(124)
    got = D.get_test_sequences(user, profile, L, 10, attrs, ctx, "test", True)
    for nm, g in zip(names, got):
        want = np.asarray(fx.outs["test/" + nm])
        assert g.dtype == want.dtype and g.shape == want.shape and np.array_equal(g, want), nm
    # ids-only variant: same ids / ctx / labels, no attribute tensors
    random.seed(124)
    p_x, p_c, o_x, o_c, y = D.get_test_sequences(user, profile, L, 10, attrs, ctx, "test", True, with_attrs=False)
    assert np.array_equal(p_x, got[0]) and np.array_equal(o_x, got[3]) and np.array_equal(o_c, got[5])


def test_dataset_lengths_and_collation():
    import torch
    from torch.utils.data import DataLoader

    rng = np.random.default_rng(0)
    n_items, n_attrs, n_ctx = 40, 5, 2
    attrs = rng.random((n_items, n_attrs), dtype=np.float32)
    attrs[0] = 0
    profiles = {u: [int(v) for v in rng.integers(1, n_items, size=int(rng.integers(1, 9)))] for u in range(12)}
    ctx = {(u, it): rng.random(n_ctx, dtype=np.float32) for u, p in profiles.items() for it in p}
    for mode, floor in (("train", 1), ("val", 2), ("test", 3)):
        ds = D.CARCADataset(list(profiles), list(range(1, n_items)), profiles, attrs, ctx, 6, 10, mode, test=True)
        assert len(ds) == sum(1 for p in profiles.values() if len(p) > floor)
    random.seed(1)
    ds = D.CARCADataset(list(profiles), list(range(1, n_items)), profiles, attrs, ctx, 6, 10, "val", test=True)
    batch = next(iter(DataLoader(ds, batch_size=4)))
    p_x, p_a, p_c, o_x, o_a, o_c, y = batch
    assert p_x.dtype == torch.int32 and p_a.shape == (4, 6, n_attrs) and o_x.shape == (4, 11) and y.dtype == torch.int32
    assert bool((o_x[:, 0] != 0).all()) and bool((y[:, 0] == 1).all())
