"""BASELINE.json configs[2] and configs[4] at FULL size (C3 = Fashion shape, B = 512 users, n_attrs = 4096 dense
image features, n_items = 166 k; C5 = Games shape, B = 128, 1 + 1000 candidates, n_attrs = 512 categorical MULTI-HOT {0,1}
attributes, 6 timestamp context features).  The oracle cannot run these batches in test time; the domain's invariants can:
a user's scores do not depend on who else is in the batch (the same users in smaller batches take other kernels, pinned
against the oracle at that size elsewhere), ids-only batches over the registered attribute table equal the dense batch,
and EVERY user is checked against the oracle directly (scores to 2e-5, every unambiguous rank identical; the CPU port
scores ~1 k users/s, so this is seconds)."""
import numpy as np
import pytest
import torch

from oracle import carca_oracle as O
from tests.model_util import assert_all_users_match_oracle, model_from_params

pytestmark = pytest.mark.gpu
Y_ATOL = 1e-4


def _ids(B, L, N, n_items, seed):
    """lengths U{3..L} left-padded, ids U{1..n_items-1}, N - 1 distinct negatives outside the profile."""
    rng = np.random.default_rng(seed)
    p_x = np.zeros((B, L), dtype=np.int32)
    o_x = np.zeros((B, N), dtype=np.int32)
    for u in range(B):
        ell = int(rng.integers(3, L + 1))
        p_x[u, L - ell:] = rng.integers(1, n_items, size=ell)
        seen = set(p_x[u].tolist())
        cand = rng.permutation(np.setdiff1d(rng.integers(1, n_items, size=3 * N), np.fromiter(seen, dtype=np.int64)))[:N]
        assert len(cand) == N
        o_x[u] = cand
    return torch.from_numpy(p_x), torch.from_numpy(o_x)


def _run(cfg, g, n_items, n_attrs, n_ctx, L, N, B, table, p_c, o_c, p_x, o_x, splits):
    P = O.perturb_params(O.init_params(cfg, n_items, g, n_ctx, n_attrs, L, seed=0), seed=1)
    model = model_from_params(P, cfg).eval()
    px, ox = p_x.cuda(), o_x.cuda()
    p_a, o_a = table[px.long()], table[ox.long()]  # the dense batch as the reference's DataLoader would deliver it
    with torch.no_grad():
        full = model(profile=(px, p_a, p_c), targets=[(ox, o_a, o_c)])
        for step in splits:
            parts = [model(profile=(px[i:i + step], p_a[i:i + step], p_c[i:i + step]),
                           targets=[(ox[i:i + step], o_a[i:i + step], o_c[i:i + step])]) for i in range(0, B, step)]
            split = torch.cat(parts, dim=0)
            assert split.shape == full.shape == (B, N)
            assert float((full - split).abs().max()) < 2e-5, step
        model.embeds.register_attr_table(table)
        ids_only = model(profile=(px, None, p_c), targets=[(ox, None, o_c)])
        model.embeds.register_attr_table(None)
    assert float((full - ids_only).abs().max()) < 1e-6
    # EVERY user against the oracle, 128 at a time (the dense batch goes to the host in slices: C3's is 1.3 GB)
    for i in range(0, B, 128):
        c = lambda t: t[i:i + 128].cpu()  # noqa: E731
        want = O.carca_forward(P, cfg, (c(p_x), c(p_a), c(p_c)), [(c(o_x), c(o_a), c(o_c))], training=False)
        assert_all_users_match_oracle(full[i:i + 128], want, 2e-5)
    assert bool(torch.isfinite(full).all()) and float(full.std()) > 0
    return full


def test_c3_fashion_shape_full_size():
    """B = 512 users x (50 + 101) rows x 4102 features: the feature GEMM's multi-round path (77,312 rows = 4 x C2) over a
    2.7 GB attribute table."""
    cfg = O.CarcaConfig(d=90, H=3, n_blocks=2)
    g, n_items, n_attrs, n_ctx, L, N, B = 450, 166_001, 4096, 6, 50, 101, 512
    gen = torch.Generator(device="cuda").manual_seed(7)
    table = torch.rand(n_items, n_attrs, device="cuda", generator=gen)
    table[0] = 0
    p_x, o_x = _ids(B, L, N, n_items, seed=31)
    p_c = torch.rand(B, L, n_ctx, device="cuda", generator=gen) * (p_x.cuda() != 0)[..., None]
    o_c = torch.rand(B, 1, n_ctx, device="cuda", generator=gen).expand(B, N, n_ctx).contiguous()
    _run(cfg, g, n_items, n_attrs, n_ctx, L, N, B, table, p_c, o_c, p_x, o_x, splits=(128, 16))


def test_c5_games_shape_full_size():
    """1 + 1000 candidates per user (134,528 embedded rows; 63 target tiles per user in the scoring kernel), 512 categorical
    multi-hot attributes (about 4 % ones -- {0,1} exactly), context = 6 timestamp features (year, month, day, weekday, hour,
    day-of-year scaled to [0, 1], data.py:17-25 hands them over as a float vector) shared by a user's candidates."""
    cfg = O.CarcaConfig(d=90, H=3, n_blocks=2)
    g, n_items, n_attrs, n_ctx, L, N, B = 450, 30_001, 512, 6, 50, 1001, 128
    gen = torch.Generator(device="cuda").manual_seed(11)
    table = (torch.rand(n_items, n_attrs, device="cuda", generator=gen) < 0.04).float()
    table[0] = 0
    assert set(table.unique().tolist()) == {0.0, 1.0}
    p_x, o_x = _ids(B, L, N, n_items, seed=37)

    def stamps(shape):
        ts = torch.randint(1_300_000_000, 1_700_000_000, shape, device="cuda", generator=gen).double()  # unix seconds
        day = torch.floor(ts / 86400.0)
        feats = [(1970.0 + day / 365.25 - 2010.0) / 15.0, (day / 30.4375) % 12.0 / 12.0, (day % 30.4375) / 30.4375,
                 ((day + 4.0) % 7.0) / 7.0, (ts % 86400.0) / 86400.0, (day % 365.25) / 365.25]
        return torch.stack(feats, dim=-1).float()

    p_c = stamps((B, L)) * (p_x.cuda() != 0)[..., None]
    o_c = stamps((B, 1)).expand(B, N, n_ctx).contiguous()
    full = _run(cfg, g, n_items, n_attrs, n_ctx, L, N, B, table, p_c, o_c, p_x, o_x, splits=(16,))
    # full-ranking metrics over 1001 candidates: the device's sort-free count equals a sort on the host
    from carca_replication_amd import ops

    sums, rank = ops.rank_metrics(full, 10, want_rank=True)
    y = full.cpu()
    want_rank = (y[:, 1:] > y[:, :1]).sum(1)
    assert torch.equal(rank.cpu().long(), want_rank) and float(sums[0]) == float((want_rank < 10).sum())
