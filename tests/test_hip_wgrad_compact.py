"""gemm_wgrad_cu_kernel leaves masked rows OUT of its row table (csrc/wgrad_cu.hip: WgPlan): the weight gradient of
feats_embed (carca.py:86 seen from the backward side) over the rows that take part only.  Rows with ids == 0 contribute
nothing by definition (mask_rows = the e * mask of carca.py:94); dropping them must give the bits' worth of the same sums as
carrying them as zeros (tuning variant 22), for every mix: scattered pads, a segment that is all pads, nothing masked, row
counts that are no multiple of the 32-row chunk."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rows_per_seg, pad_frac, seed, N=450, K=1536, K1=6):
    g = torch.Generator(device="cuda").manual_seed(seed)
    segs, ref_dy, ref_x = [], [], []
    for i, rows in enumerate(rows_per_seg):
        dy = torch.randn(rows, N, device="cuda", generator=g)
        x = torch.rand(rows, K, device="cuda", generator=g)
        x1 = torch.rand(rows, K1, device="cuda", generator=g)
        ids = torch.randint(1, 100, (rows,), device="cuda", generator=g, dtype=torch.int32)
        frac = pad_frac[i] if isinstance(pad_frac, (list, tuple)) else pad_frac
        ids[torch.rand(rows, device="cuda", generator=g) < frac] = 0
        segs.append(dict(dy=dy, x=x, x1=x1, ids=ids))
        keep = (ids != 0).float()[:, None]
        ref_dy.append(dy * keep)
        ref_x.append(torch.cat([x, x1], 1))
    want = torch.cat(ref_dy).double().t() @ torch.cat(ref_x).double()
    want_b = torch.cat(ref_dy).double().sum(0)
    return segs, want, want_b, N, K, K1


@pytest.mark.parametrize("rows_per_seg,pad_frac", [
    ((6400, 6400, 6400), 0.47),          # a C2 training batch's row counts and pad share
    ((6400, 6400, 6400), (1.0, 0.3, 0.0)),  # one segment all pads, one without any
    ((5000, 3333), 0.9),                 # few rows left, ragged chunk tails
    ((6400, 6400, 6400), 0.0),           # nothing masked: the table as it always was
], ids=["c2-mix", "all-pad-segment", "mostly-pads", "no-pads"])
def test_compacted_weight_gradient_equals_the_uncompacted_one_and_fp64(rows_per_seg, pad_frac):
    from carca_replication_amd import ops

    segs, want, want_b, N, K, K1 = _case(rows_per_seg, pad_frac, seed=3)

    def run(variant):
        dw = torch.zeros(N, K + K1, device="cuda")
        db = torch.zeros(N, device="cuda")
        ops.set_tuning(0, variant)
        try:
            ops.gemm_wgrad(segs, N, K, dw, db, mask_rows=True, K1=K1)
        finally:
            ops.set_tuning(0, 0)
        return dw, db

    dw_c, db_c = run(0)
    dw_u, db_u = run(22)
    scale = float(want.abs().max())
    assert float((dw_c.double() - want).abs().max()) < 2e-5 * scale
    assert float((dw_u.double() - want).abs().max()) < 2e-5 * scale
    assert float((db_c.double() - want_b).abs().max()) < 2e-5 * float(want_b.abs().max() + 1.0)
    # the same rows in the same order, minus rows of zeros: the chunk boundaries move, so the partial sums group
    # differently -- round-off apart, not bitwise
    assert float((dw_c - dw_u).abs().max()) < 2e-6 * scale


def test_everything_masked_is_a_zero_gradient():
    from carca_replication_amd import ops

    segs, _, _, N, K, K1 = _case((6400, 6400), 1.0, seed=4)
    dw = torch.zeros(N, K + K1, device="cuda")
    db = torch.zeros(N, device="cuda")
    ops.gemm_wgrad(segs, N, K, dw, db, mask_rows=True, K1=K1)
    torch.cuda.synchronize()
    assert float(dw.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
