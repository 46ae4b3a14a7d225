"""CPU model of how k_normals3 / k_normals3s cut a map (or a region of it) into strips: launch3's planning and n3_block
(te_normals3.hip), restated in Python.  Checked for whole maps and regions, one round of blocks, oversubscribed grids and
the short strips of the unobserved-region march: every cell of the region is owned by exactly one block, the blocks that
take the closed-form tail hold no cell whose disc leaves the map, and strips never exceed their planned height."""
import numpy as np
import pytest

LANES = 64
SHORT = 32  # kN3ShortStripRows


def plan(rows, cols, R, region, capacity, short_strips=False):
    """N3Args as normals_fast3 + launch3 fill them (rows = cells along i, the lane axis; cols = rows of the march)."""
    i_lo, i_hi, j_lo, j_hi = region
    a = dict(rows=rows, cols=cols, R=R, i_lo=i_lo, i_hi=i_hi, j_lo=j_lo, j_hi=j_hi)
    nbx = (i_hi - i_lo + LANES - 1) // LANES

    def is_edge(bx):
        i0 = i_lo + bx * LANES
        i0 = i_hi - LANES if i0 + LANES > i_hi else i0
        return i0 < R or i0 + LANES - 1 > rows - 1 - R

    e0 = 0
    while e0 < nbx and is_edge(e0) and i_lo + e0 * LANES < R:
        e0 += 1
    e1 = 0
    while e1 < nbx - e0 and is_edge(nbx - 1 - e1):
        e1 += 1
    H = j_hi - j_lo
    ne = e0 + e1
    n_int = nbx - ne
    jf_lo = j_lo if j_lo > R else (R if R < j_hi else j_hi)
    jf_hi = j_hi if j_hi < cols - R else (cols - R if cols - R > jf_lo else jf_lo)
    n_top = n_int if jf_lo > j_lo else 0
    n_bottom = n_int if j_hi > jf_hi else 0
    Hf = jf_hi - jf_lo
    edge_rows = lambda h: (50 * h + 99) // 100
    rows_int, fits = 512, False
    for h in range(8, 513):
        he = edge_rows(h)
        if n_int * -(-Hf // h) + ne * -(-H // he) + n_top + n_bottom <= capacity:
            rows_int, fits = h, True
            break
    if not fits:
        c0, best = float(R + 6), 0.0
        for h in range(16, 513, 8):
            he = edge_rows(h)
            si, se = float(-(-Hf // h)), float(-(-H // he))
            work = n_int * (Hf + si * c0 if si > 0 else 0.0) + 1.5 * ne * (H + se * c0) + (n_top + n_bottom) * (R + c0)
            t = work / capacity + 0.5 * (h + c0)
            if best == 0.0 or t < best:
                best, rows_int = t, h
    if short_strips and fits and rows_int > SHORT:
        rows_int = SHORT
    a.update(nbx=nbx, edge0=e0, edge1=e1, n_int=n_int, jf_lo=jf_lo, jf_hi=jf_hi, n_top=n_top, rows_int=rows_int,
             rows_edge=edge_rows(rows_int))
    a["s_int"] = -(-Hf // rows_int) if n_int > 0 and Hf > 0 else 0
    a["s_edge"] = -(-H // a["rows_edge"]) if ne > 0 else 0
    a["nblocks"] = n_int * a["s_int"] + ne * a["s_edge"] + n_top + n_bottom
    return a


def n3_block(a, b):
    nb_fast, ne = a["n_int"] * a["s_int"], a["edge0"] + a["edge1"]
    general = True
    if b < nb_fast:
        general = False
        bx = a["edge0"] + b % a["n_int"]
        js = a["jf_lo"] + (b // a["n_int"]) * a["rows_int"]
        jend = min(js + a["rows_int"], a["jf_hi"])
    elif b - nb_fast < ne * a["s_edge"]:
        b -= nb_fast
        q = b % ne
        bx = q if q < a["edge0"] else a["nbx"] - ne + q
        js = a["j_lo"] + (b // ne) * a["rows_edge"]
        jend = min(js + a["rows_edge"], a["j_hi"])
    else:
        b -= nb_fast + ne * a["s_edge"]
        bottom = b >= a["n_top"]
        bx = a["edge0"] + (b - a["n_top"] if bottom else b)
        js, jend = (a["jf_hi"], a["j_hi"]) if bottom else (a["j_lo"], a["jf_lo"])
    own_lo = a["i_lo"] + bx * LANES
    i0 = a["i_hi"] - LANES if own_lo + LANES > a["i_hi"] else own_lo
    return (js < jend), i0, own_lo, js, jend, general


CASES = [
    # rows, cols, R, region (i_lo, i_hi, j_lo, j_hi; None: the whole map), slots
    (4096, 4096, 9, None, 11 * 256),
    (4096, 4096, 9, None, 12 * 256),
    (1024, 1024, 5, None, 12 * 256),
    (100, 133, 2, None, 12 * 256),
    (700, 333, 10, None, 300),          # more blocks than slots whatever the height
    (512, 512, 5, None, 6),             # a batch's share of the slots
    (521, 481, 4, None, 12 * 256),      # the last block of a row of blocks shifted left
    (64, 64, 3, None, 12 * 256),        # one block column, both borders in it
    (2048, 2048, 9, (300, 900, 100, 700), 11 * 256),     # a region in the interior
    (2048, 2048, 9, (0, 200, 0, 50), 11 * 256),          # a region in the corner, shorter than the frame is wide
    (2048, 2048, 9, (1900, 2048, 2000, 2048), 11 * 256),
    (4096, 4096, 1, None, 12 * 256),
]


@pytest.mark.parametrize("rows,cols,R,region,slots", CASES)
@pytest.mark.parametrize("short", [False, True])
def test_every_cell_has_one_owner(rows, cols, R, region, slots, short):
    region = region or (0, rows, 0, cols)
    a = plan(rows, cols, R, region, slots, short_strips=short)
    i_lo, i_hi, j_lo, j_hi = region
    assert i_hi - i_lo >= LANES
    owner = np.zeros((j_hi - j_lo, i_hi - i_lo), dtype=np.int32)
    closed = np.zeros_like(owner, dtype=bool)
    live = 0
    for b in range(a["nblocks"]):
        ok, i0, own_lo, js, jend, general = n3_block(a, b)
        if not ok:
            continue
        live += 1
        assert i_lo <= i0 and i0 + LANES <= i_hi and j_lo <= js < jend <= j_hi
        nb_fast, nb_edge = a["n_int"] * a["s_int"], (a["edge0"] + a["edge1"]) * a["s_edge"]
        limit = a["rows_int"] if b < nb_fast else a["rows_edge"] if b < nb_fast + nb_edge else R  # interior / edge column / frame rows
        assert jend - js <= limit
        lo = max(i0, own_lo)  # the lanes of a shifted block that its neighbour owns store nothing
        owner[js - j_lo:jend - j_lo, lo - i_lo:i0 + LANES - i_lo] += 1
        if not general:
            closed[js - j_lo:jend - j_lo, lo - i_lo:i0 + LANES - i_lo] = True
            # every lane of the block, owned or not, runs the closed form: none of their discs may leave the map
            assert i0 >= R and i0 + LANES - 1 <= rows - 1 - R and js >= R and jend - 1 <= cols - 1 - R
    assert (owner == 1).all(), (int((owner == 0).sum()), int((owner > 1).sum()))
    # the closed-form blocks are the bulk of a large map
    if rows >= 1024 and cols >= 1024 and region == (0, rows, 0, cols):
        assert closed.mean() > 0.8
    if short and live <= slots * 8:  # (a grid that fitted one round before is cut into strips of at most 32 rows)
        assert a["rows_int"] <= max(SHORT, 8) or a["nblocks"] > slots


def test_short_strips_only_shorten():
    for rows, cols, R, region, slots in CASES:
        region = region or (0, rows, 0, cols)
        a, b = plan(rows, cols, R, region, slots), plan(rows, cols, R, region, slots, short_strips=True)
        assert b["rows_int"] <= a["rows_int"] and b["nblocks"] >= a["nblocks"]
        if a["rows_int"] <= SHORT:
            assert a == b


@pytest.mark.parametrize("R", [1, 2, 5, 9, 10])
@pytest.mark.parametrize("slim", [False, True])
def test_clean_march_hands_over_at_a_finished_row(R, slim):
    """march3<HOLES = 0>: the rows it stages before it stops (the first window, then row j + LEAD + R in the step of row
    j) and the row it returns, for every position of the first dirty row: every row below the returned one was finished
    with a disc that holds no dirty row, and the march does not stop earlier than the staging makes it."""
    js, jend = 40, 40 + 37
    lead = 1 if slim else 2
    first_window = range(js - R, js + R + 1) if slim else range(js - R, js + R + 2)  # (SLIM: row js - R is the lane's own cell, read apart)
    for dirty in range(js - R - 3, jend + R + 4):
        if dirty in first_window:
            ret = js
        else:
            ret, j = jend, js
            while j < jend:
                staged = j + lead + R   # tail(j), slide, stage_row(staged), store_row, ++j
                j += 1
                if staged == dirty and j < jend:
                    ret = j
                    break
        for r in range(js, ret):  # finished by the clean march: closed form, no invalid cell may be in the disc
            assert not (r - R <= dirty <= r + R), (dirty, r, ret)
        if ret < jend:            # ... and the row it hands over is the first one the dirty row can matter to, or one before
            assert ret + R >= dirty - 1
