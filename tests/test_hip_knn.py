"""GPU parity of the KNN baseline model (reference src/knn.py:8-21) through the C ABI (carca_knn_score).

Raw fp32 dot products over n_attrs features: the summation order differs from ATen's (wave tree vs its vectorised
loop), so scores agree to ~1e-6 relative of sum |a_i b_i|; the bound is written per test.
"""
import numpy as np
import pytest
import torch

from oracle import carca_oracle as O
from tests.golden_util import load

pytestmark = pytest.mark.gpu


def _model():
    from src.knn import KNN  # the drop-in import path

    return KNN().cuda().eval()


@pytest.mark.parametrize("tag", ["f32", "f37"])
def test_g10_knn_matches_reference(tag):
    fx = load("g10_knn")
    L = fx.dim["L"]
    i = {k[len(tag) + 1:]: v.cuda() for k, v in fx.ins.items() if k.startswith(tag + "/")}
    model = _model()
    y = model(profile=(i["p_x"], i["p_a"], i["p_c"]), targets=[(i["o_x"], i["o_a"], i["o_c"])])
    ref = fx.outs[tag + "/y"]
    assert y.shape == ref.shape and y.dtype == torch.float32
    assert torch.allclose(y.cpu(), ref, atol=1e-5, rtol=1e-6)
    # the train-loop call shape (train.py:86-91): two groups that are torch.split views of the 2L-wide tensors
    pos = tuple(torch.split(i["train/" + k], L, dim=1)[0] for k in ("o_x", "o_a", "o_c"))
    neg = tuple(torch.split(i["train/" + k], L, dim=1)[1] for k in ("o_x", "o_a", "o_c"))
    yt = model(profile=(i["train/p_x"], i["train/p_a"], i["train/p_c"]), targets=[pos, neg])
    assert torch.allclose(yt.cpu(), fx.outs[tag + "/train/y"], atol=1e-5, rtol=1e-6)


@pytest.mark.parametrize("B,L,T,F", [(1, 1, 1, 1), (3, 5, 7, 4), (5, 9, 33, 1023), (128, 50, 101, 4096), (2, 3, 1000, 516)])
def test_knn_vs_oracle_and_table_mode(B, L, T, F):
    g = torch.Generator().manual_seed(B * 1000 + F)
    n_items = 400
    table = torch.rand(n_items, F, generator=g)
    table[0] = 0
    p_x = torch.randint(0, n_items, (B, L), generator=g, dtype=torch.int32)
    o_x = torch.randint(0, n_items, (B, T), generator=g, dtype=torch.int32)
    p_a, o_a = table[p_x.long()], table[o_x.long()]
    ref = O.knn_forward((p_x, p_a, None), [(o_x, o_a, None)])
    tol = 2e-6 * F * 1.0 + 1e-6  # entries in [0, 1): sum |a b| <= F
    model = _model()
    y = model(profile=(p_x.cuda(), p_a.cuda(), None), targets=[(o_x.cuda(), o_a.cuda(), None)])
    assert float((y.cpu() - ref).abs().max()) <= tol
    # ids-only batches with the attribute table resident on the device: identical arithmetic, identical bits
    with pytest.raises(ValueError):
        model(profile=(p_x.cuda(), None, None), targets=[(o_x.cuda(), None, None)])
    model.register_attr_table(table.cuda())
    y2 = model(profile=(p_x.cuda(), None, None), targets=[(o_x.cuda(), None, None)])
    assert torch.equal(y2, y)


def test_knn_table_ids_outside_score_zero_and_unaligned_views():
    from carca_replication_amd import ops

    table = torch.rand(10, 8).cuda()
    p_x = torch.tensor([[1, 2], [3, 99]], dtype=torch.int32).cuda()
    o_x = torch.tensor([[4, -1, 5], [6, 7, 8]], dtype=torch.int32).cuda()
    y = ops.knn_score(None, None, p_x, o_x, table=table).cpu()
    t = table.cpu()
    assert y[0, 1] == 0 and torch.all(y[1] == 0)
    assert abs(float(y[0, 0]) - float((t[2] * t[4]).sum())) < 1e-6
    # a dense operand that starts 4 bytes into its storage takes the scalar path
    buf = torch.rand(2 * 3 * 8 + 1).cuda()
    p_a = buf[1:].view(2, 3, 8)
    o_a = torch.rand(2, 5, 8).cuda()
    y = ops.knn_score(p_a, o_a).cpu()
    ref = (p_a[:, -1:, :] * o_a).sum(-1).cpu()
    assert torch.allclose(y, ref, atol=1e-6)


def test_knn_bad_arguments_raise():
    from carca_replication_amd import ops

    with pytest.raises(ValueError):
        ops.knn_score(torch.rand(2, 3, 8).cuda(), torch.rand(2, 5, 9).cuda())
    with pytest.raises(ops.CarcaHipError):
        ops.knn_score(torch.rand(2, 3, 8).cuda().double(), torch.rand(2, 5, 8).cuda().double())
