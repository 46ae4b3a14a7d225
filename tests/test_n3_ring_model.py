"""CPU model of k_normals3 / k_normals3s' ring addressing (te_normals3.hip, march3): which physical ring slot every read of
a slide touches, for the full ring (2R + 2 rows) and the slim one (2R rows, k_normals3s), replayed over several chunk
rotations for every radius -- the slots must hold exactly the map rows the slide's formulas name, the staged row must
land on the oldest row after that step's reads, and in the slim ring the cell read back before the overwrite must be the
next step's trailing centre cell.  (The arithmetic below restates the kernel's constexpr expressions one for one.)"""
import math

import pytest


def chunk_rows(nr, slim):
    pref = [4, 3, 2, 5, 6, 7, 8, 9, 10, 11] if slim else [4, 5, 6, 3, 7, 8, 9, 10, 11, 2]
    for c in pref:
        if nr % c == 0:
            return c
    return 1


def half_heights(q, r):
    return [math.isqrt(q - d * d) for d in range(r + 1)]


@pytest.mark.parametrize("r", range(1, 11))
@pytest.mark.parametrize("slim", [False, True])
def test_ring_slots_hold_the_rows_the_slide_reads(r, slim):
    q = r * r  # the shapes the slim ring serves; the full ring's arithmetic does not depend on the shape
    hw = half_heights(q, r)
    if slim and (r < 2 or hw[1] >= r):
        pytest.skip("not a slim shape")
    nr = 2 * r if slim else 2 * r + 2
    lead = 1 if slim else 2
    old = r - 1 if slim else r
    c = chunk_rows(nr, slim)
    nc = nr // c
    js = 100
    vb = [k * c for k in range(nc)]           # chunk base "registers": physical slot of the chunk's first row
    ring = {}                                  # physical slot -> map row
    for p in range(nr):                        # the start: rows js - old .. in ring rows 0 .. nr - 1
        ring[vb[p // c] + p % c] = js - old + p
    j = js
    ctr_old_row = js - r                       # slim: the row whose own cell the march holds in a register
    for _ in range(6 * nc):                    # several full rotations
        for u in range(c):
            for d in range(r + 1):
                h = hw[d]
                if slim and d == 0:
                    assert ctr_old_row == j - r  # trailing centre cell: from the register
                    continue                      # (leading centre cell: the row staged below, from its prefetch register)
                pl, pt = u + old + 1 + h, u + old - h
                sl = vb[(pl // c) % nc] + pl % c
                st = vb[(pt // c) % nc] + pt % c
                assert ring[sl] == j + 1 + h, (r, slim, j, d)
                assert ring[st] == j - h, (r, slim, j, d)
            target = vb[0] + u                  # stage_row(j + lead + r, vb[0], u)
            assert ring[target] == j - old      # the oldest row, all of whose reads are behind us
            if slim:
                ctr_old_row = ring[target]      # read back before the overwrite: row j - r + 1 = (j + 1) - r
            ring[target] = j + lead + r
            j += 1
        vb = vb[1:] + vb[:1]                    # rotate()
    assert sorted(ring.values()) == list(range(j - old, j - old + nr))
