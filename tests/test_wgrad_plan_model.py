"""The unit ranges of gemm_wgrad_cu_kernel (csrc/wgrad_cu.hip: WgRanges, and the plan wgrad_rowtab_kernel writes once it has
counted the rows that take part), restated in Python -- no GPU needed.  The launcher sizes the groups' partial-tile slots
BEFORE the device knows how many chunks survive the compaction: slots_pg = nkb // ngroups + 2 must hold for every chunk
count >= 1, every weight of the slow k block and every group count the chip can host (the kernel indexes `part`
with kb - kb_first; since round 5 it also checks that index on the device and raises the library's error word)."""
import itertools


def ranges(nkb, nchunks, ngroups_host, slow_w, has_slow):
    """(begin of every group's unit range, number of groups that get units): units = (k block, chunk) pairs laid end to end,
    the last k block's units weighing slow_w / 256 when it carries the second k-source's columns."""
    total = nkb * nchunks
    n_slow = nchunks if has_slow else 0
    n_fast = total - n_slow
    total_w = n_fast * 256 + n_slow * slow_w
    per_w = max(1, (total_w + ngroups_host - 1) // ngroups_host)
    ngroups = (total_w + per_w - 1) // per_w

    def item_at(wt):
        fast_w = n_fast * 256
        return wt // 256 if wt <= fast_w else n_fast + (wt - fast_w) // slow_w

    def begin_of(g):
        return total if g >= ngroups else min(total, item_at(g * per_w))

    return [begin_of(g) for g in range(ngroups + 1)], ngroups


def test_a_groups_range_never_touches_more_k_blocks_than_it_has_slots():
    worst = 0
    for nkb, ngroups_host in itertools.product([1, 2, 3, 5, 11, 12, 33, 43, 64, 129], [1, 3, 8, 25, 48, 51, 64, 128]):
        slots_pg = nkb // ngroups_host + 2
        for nchunks, slow_w, has_slow in itertools.product([1, 2, 3, 7, 8, 100, 317, 600, 4200], [256, 264], [False, True]):
            begin, ngroups = ranges(nkb, nchunks, ngroups_host, slow_w, has_slow)
            assert ngroups <= ngroups_host and begin[0] == 0 and begin[ngroups] == nkb * nchunks
            covered = 0
            for g in range(ngroups):
                lo, hi = begin[g], begin[g + 1]
                assert lo <= hi
                covered += hi - lo
                if hi > lo:
                    touched = (hi - 1) // nchunks - lo // nchunks + 1  # k blocks kb_first .. kb of the units lo .. hi - 1
                    worst = max(worst, touched - slots_pg)
                    assert touched <= slots_pg, (nkb, ngroups_host, nchunks, slow_w, has_slow, g, touched, slots_pg)
            assert covered == nkb * nchunks  # every unit in exactly one range
    assert worst <= 0
