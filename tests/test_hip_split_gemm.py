"""The opt-in split-precision feature GEMM (csrc/gemm_split.hip, tuning key 16; SURVEY 7 hard part 1: "fp32 MFMA, or a
bf16x3 split-accumulate scheme validated against fixtures"): AllEmbedding.feats_embed (carca.py:86) on the 16-bit MFMA pipe
with both operands split into 16-bit parts -- 'bf16x3' (six products) and 'fp16x2' (three products, scaled residuals).

Parity is held at the DEFAULT path's tolerances (2e-5 on probabilities, 1e-4 on activations, every user's rank identical):
  * the reference's fixtures G1 / G8 / G11 (the kernel forced at fixture sizes, where the launcher would not pick it),
  * the C2-sized oracle tests of tests/test_hip_forward.py, re-run under each mode,
  * the product alone at C2 size against an fp64 matmul, beside the exact-fp32 MFMA kernel's own error.
Every test asserts that the split kernel really ran (carca_split_launch_count)."""
import pytest
import torch

import tests.test_hip_forward as F
from tests.golden_util import G1_NAMES

pytestmark = pytest.mark.gpu

MODES = ["bf16x3", "fp16x2"]


@pytest.fixture(params=MODES)
def split_forced(request):
    """The mode selected, forced wherever the kernel's own conditions hold; the count of its launches when the test began."""
    from carca_replication_amd import ops

    ops.set_feature_gemm_precision(request.param, force=True)
    start = ops.split_launch_count()
    try:
        yield lambda: ops.split_launch_count() - start
    finally:
        ops.set_feature_gemm_precision("fp32")


@pytest.mark.parametrize("name", G1_NAMES)
def test_g1_eval_forward_matches_reference_on_the_split_path(split_forced, name):
    from tests.golden_util import load

    F.test_g1_eval_forward_matches_reference(name)
    # (n_attrs = 32 / 19 / 32 / 40: the kernel wants 16-byte groups -- n_attrs % 4 == 0 --, d90h2 keeps the fp32 kernels)
    takes = int(load("g1_" + name).dim["n_attrs"]) % 4 == 0
    assert (split_forced() >= 1) == takes


def test_g8_ranking_is_identical_on_the_split_path(split_forced):
    F.test_g8_ranking_hr_ndcg_identical()
    assert split_forced() >= 1


def test_g11_ranking_at_c2_model_dims_is_identical_on_the_split_path(split_forced):
    F.test_g11_ranking_at_c2_model_dims_is_identical()
    assert split_forced() >= 2


def test_c2_sized_batch_vs_oracle_on_the_split_path(split_forced):
    F.test_c2_sized_batch_vs_oracle()
    assert split_forced() >= 1


def test_c2_full_batch_is_batch_split_invariant_on_the_split_path(split_forced):
    F.test_c2_full_batch_is_batch_split_invariant(90, 3, 450)
    assert split_forced() >= 9  # (the full batch and its eight parts)


@pytest.mark.parametrize("mode", MODES)
def test_unforced_mode_takes_the_kernel_at_c2_size_and_leaves_small_products_alone(mode):
    """Without the force bit the launcher admits the split kernel only where the one-workgroup-per-CU kernel would run:
    the C2 batch takes it (with the packed planes bound by the module's weight cache), a fixture-sized model does not."""
    from carca_replication_amd import ops
    from tests.golden_util import load
    from tests.model_util import model_from_fixture

    ops.set_feature_gemm_precision(mode)
    try:
        n0 = ops.split_launch_count()
        F.test_c2_full_batch_is_batch_split_invariant(90, 3, 450)
        n1 = ops.split_launch_count()
        assert n1 - n0 == 1  # the B = 128 batch; its eight 16-user parts keep the fp32 tiled kernel
        fx = load("g1_d90h3")
        model = model_from_fixture(fx).eval()
        with torch.no_grad():
            model(*F._eval_in(fx))
        assert ops.split_launch_count() == n1
    finally:
        ops.set_feature_gemm_precision("fp32")


def _product_inputs(rows, K0, K1, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.rand(rows, K0, device="cuda", generator=g)  # attributes U[0, 1) as BASELINE.md draws them
    c = torch.rand(rows, K1, device="cuda", generator=g)
    bound = (6.0 / (K0 + K1 + N)) ** 0.5  # xavier-uniform, the reference's init (carca.py:77-79)
    w = (torch.rand(N, K0 + K1, device="cuda", generator=g) * 2 - 1) * bound
    b = torch.randn(N, device="cuda", generator=g) * 0.01
    return a, c, w, b


@pytest.mark.parametrize("mode", MODES)
def test_product_alone_at_c2_size_against_fp64(mode):
    """q = [a ; c] W^T + b for C2's 19,328 rows x (4096 + 6) x 450 through carca_gemm_rows under each mode and under the
    default, against torch's fp64 matmul: the split kernels must stay within 2x of the exact-fp32 MFMA kernel's own error
    (which is the fp32 accumulation both share), bf16x3 and fp16x2 alike."""
    from carca_replication_amd import ops

    rows, K0, K1, N = 19328, 4096, 6, 450
    a, c, w, b = _product_inputs(rows, K0, K1, N, seed=7)
    want = (torch.cat([a, c], 1).double() @ w.double().t() + b.double())

    def run():
        (q,) = ops.gemm_rows([dict(a0=a, a1=c)], w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b)
        return q[:, :N]

    base = run()
    err32 = float((base.double() - want).abs().max())
    ops.set_feature_gemm_precision(mode)
    try:
        n0 = ops.split_launch_count()
        got = run()
        assert ops.split_launch_count() == n0 + 1
    finally:
        ops.set_feature_gemm_precision("fp32")
    err = float((got.double() - want).abs().max())
    scale = float(want.abs().max())
    assert err32 < 1e-5 * scale  # (sanity of the yardstick: a chain of K = 4102 fp32 fused multiply-adds, ~3.5e-7 sum |a b|)
    assert err <= 2.0 * err32 + 1e-7 * scale, (mode, err, err32)
    print(f"[{mode}] max |q - q_fp64|: split {err:.3e}, exact-fp32 MFMA {err32:.3e}, max |q| {scale:.3f}")
    # masked rows and the row mask's exact zeros
    ids = torch.ones(rows, dtype=torch.int32, device="cuda")
    ids[5::7] = 0
    ops.set_feature_gemm_precision(mode)
    try:
        (qm,) = ops.gemm_rows([dict(a0=a, a1=c, ids=ids)], w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b, mask_rows=True)
    finally:
        ops.set_feature_gemm_precision("fp32")
    assert float(qm[5::7].abs().max()) == 0.0 and torch.equal(qm[ids != 0][:, :N], got[ids != 0])


@pytest.mark.parametrize("mode", MODES)
def test_ragged_rows_segments_and_column_tail(mode):
    """Two row segments that end inside a 384-row tile, N = 450 (66 columns in the last block), K0 = 1000 (a ragged last K
    step): the forced kernel against the fp32 kernels."""
    from carca_replication_amd import ops

    K0, K1, N = 1000, 6, 450
    a, c, w, b = _product_inputs(1000 + 333, K0, K1, N, seed=11)
    segs = [dict(a0=a[:1000], a1=c[:1000]), dict(a0=a[1000:], a1=c[1000:])]

    def run():
        return ops.gemm_rows(segs, w[:, :K0], N, K0, 452, bt1=w[:, K0:], K1=K1, bias=b)

    base = run()
    ops.set_feature_gemm_precision(mode, force=True)
    try:
        n0 = ops.split_launch_count()
        got = run()
        assert ops.split_launch_count() == n0 + 1
    finally:
        ops.set_feature_gemm_precision("fp32")
    for x, y in zip(got, base):
        assert x.shape == y.shape
        assert float((x[:, :N] - y[:, :N]).abs().max()) < 2e-6 * float(y[:, :N].abs().max())
