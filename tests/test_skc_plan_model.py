"""The stretch arithmetic of gemm_rows_skc_kernel (csrc/gemm.hip, steps 3 and 5 of its prologue), restated in Python and checked
for its invariants over many shapes -- no GPU needed: every K step of every row block is multiplied exactly once per kind of
workgroup, every partial tile that is handed over has exactly one taker waiting for it (a taker never waits for nobody), no
piece is empty, and no stretch touches more row blocks than the kernel's LDS lists hold (the launcher's bound).
The GPU tests (tests/test_hip_stream_k.py) run the kernel itself; this pins the plan the kernel and the launcher must agree on."""
import itertools

import pytest

SKC_RB = 8


def split_grid(nblk, nfull, cheap, total):
    """x teams of nfull workgroups + y lone ones (gemm.hip: 'Two kinds of workgroups')."""
    cap = max(1, total // 16)
    x = max(1, min(cap, nblk * 100 // (nfull * 100 + cheap)))

    def span(xx):
        yy = max(1, min(cap, nblk - xx * nfull))
        return max(total * 100 // xx, total * cheap // yy)

    if (x + 1) * nfull < nblk and x + 1 <= cap and span(x + 1) < span(x):
        x += 1
    y = max(1, min(cap, nblk - x * nfull))
    return x, y


def cut(j, nteams, nrb, nfast, ov):
    """First step of stretch j: the row blocks laid out `ov` steps longer each (the cost of owning one, in front of its first
    step), cut into equal stretches, mapped back (gemm.hip, step 3: 'equal in COST, not in steps')."""
    vw = nfast + ov
    v = vw * nrb * j // nteams
    b, r = divmod(v, vw)
    return b * nfast + max(0, r - ov)


def pieces(tj, nteams, total, nfast, ov=0):
    nrb = total // nfast
    lo, hi = cut(tj, nteams, nrb, nfast, ov), cut(tj + 1, nteams, nrb, nfast, ov)
    rbA, sA = lo // nfast, lo % nfast
    rbB, sB = hi // nfast, hi % nfast
    if sB == 0:
        rbB, sB = rbB - 1, nfast
    out = []
    for rb in range(rbA, rbB + 1):
        s0, s1 = (sA if rb == rbA else 0), (sB if rb == rbB else nfast)
        mode, ntake = ("give" if s0 > 0 else ("take" if s1 < nfast else "plain")), 0
        if mode == "take":
            end = (rb + 1) * nfast
            j2 = tj + 1
            while j2 < nteams and cut(j2, nteams, nrb, nfast, ov) < end:
                ntake, j2 = ntake + 1, j2 + 1
            if ntake <= 0:
                mode = "plain"
        out.append((rb, s0, s1, mode, ntake))
    return out


@pytest.mark.parametrize("ov", [0, 4, 12])
@pytest.mark.parametrize("nblk,nfull,cheap", [(256, 4, 74), (255, 4, 74), (256, 6, 68), (256, 2, 68), (304, 4, 71)])
def test_every_step_once_every_partial_taken(nblk, nfull, cheap, ov):
    for nrb, nfast in itertools.product([1, 2, 3, 7, 27, 43, 51, 85, 170, 202, 203], [64, 65, 128, 200]):
        total = nrb * nfast
        x, y = split_grid(nblk, nfull, cheap, total)
        assert x * nfull + y <= nblk and x >= 1 and y >= 1
        for nteams in (x, y):
            cover = [[0] * nfast for _ in range(nrb)]
            gives = {}   # row block -> stretches that hand a partial tile to its owner
            takes = {}   # row block -> (owner stretch, partials it waits for)
            for tj in range(nteams):
                ps = pieces(tj, nteams, total, nfast, ov)
                assert len(ps) >= 1
                assert sum(1 for p in ps if p[3] == "give") <= 1 and (not ps or all(p[3] != "give" for p in ps[1:]))
                for rb, s0, s1, mode, ntake in ps:
                    assert 0 <= s0 < s1 <= nfast  # no empty piece
                    for k in range(s0, s1):
                        cover[rb][k] += 1
                    if mode == "give":
                        gives.setdefault(rb, []).append(tj)
                    elif mode == "take":
                        takes[rb] = (tj, ntake)
                    else:
                        assert (s0, s1) == (0, nfast)  # a plain piece is a whole row block
            assert all(c == 1 for row in cover for c in row)
            for rb, (owner, ntake) in takes.items():  # the takers' partials are exactly the stretches right behind the owner
                assert gives.get(rb) == list(range(owner + 1, owner + 1 + ntake))
            for rb, g in gives.items():
                assert rb in takes and takes[rb][1] == len(g)


@pytest.mark.parametrize("nfast", [64, 128])
def test_the_launchers_bound_on_row_blocks_per_stretch(nfast):
    """launch_gemm_rows_skc admits a product only if ceil(nrb_max (nfast + SKC_OV_MAX) / (teams nfast)) + 1 stays within
    SKC_RB for x teams and for y lone workgroups, with y taken for x + 1 teams (the kernel may pick either): then no stretch,
    whatever the kept rows and whatever ownership cost up to SKC_OV_MAX, touches more blocks."""
    nblk, nfull, cheap, ov_max = 255, 4, 74, 12
    wv = nfast + ov_max
    for nrb_max in range(1, 260):
        x0 = max(1, nblk * 100 // (nfull * 100 + cheap))
        y0 = max(1, nblk - (x0 + 1) * nfull)
        admitted = ((nrb_max * wv + x0 * nfast - 1) // (x0 * nfast) + 1 <= SKC_RB and
                    (nrb_max * wv + y0 * nfast - 1) // (y0 * nfast) + 1 <= SKC_RB)
        if not admitted:
            continue
        for nrb in {1, nrb_max // 2 + 1, nrb_max}:
            total = nrb * nfast
            x, y = split_grid(nblk + 1, nfull, cheap, total)
            for nteams in (x, y):
                for ov in (0, 4, ov_max):
                    assert max(len(pieces(tj, nteams, total, nfast, ov)) for tj in range(nteams)) <= SKC_RB
