"""The C oracle of the footprint path against a second, independently written restatement (tests/ref_py/grid_map_ref.py):
SpiralIterator order and getCurrentRadius, CircleIterator membership, LineIterator, checkForSlope / checkForStep /
checkForRoughness (getSubmap included) and isTraversable(circle), on randomised small maps with borders, kerbs, holes,
tie radii and map offsets.  CPU only."""
import math

import numpy as np
import pytest

from tests.ref_py.grid_map_ref import GridMapRef, TraversabilityMapRef


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.mark.parametrize("rows,cols,res,pos", [(23, 31, 0.05, (0.0, 0.0)), (17, 12, 0.03, (1.25, -0.5)), (9, 40, 0.1, (-30.0, 12.5))])
def test_spiral_order_and_radius(O, rows, cols, res, pos):
    g = O.geom(rows, cols, res, pos)
    gm = GridMapRef(rows, cols, res, pos)
    rng = np.random.default_rng(rows * 100 + cols)
    for radius in (0.5 * res, 1.7 * res, 3 * res, 3 * res * (1 + 1e-6), 4.5 * res, 6.2 * res):
        for _ in range(12):
            ci, cj = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            di, dj, ring = O.spiral_offsets(g, ci, cj, radius)
            mine = list(gm.spiral(gm.position((ci, cj)), radius))
            assert [(q[0] - ci, q[1] - cj) for q, _ in mine] == list(zip(di.tolist(), dj.tolist())), (radius, ci, cj)
            assert [r for _, r in mine] == [float(k) * res for k in ring.tolist()]


def test_circle_membership(O):
    for rows, cols, res, pos in ((20, 27, 0.05, (0.0, 0.0)), (15, 15, 0.04, (3.0, -7.0))):
        g = O.geom(rows, cols, res, pos)
        gm = GridMapRef(rows, cols, res, pos)
        rng = np.random.default_rng(7)
        for radius in (0.4 * res, res, 1.67 * res, 2.5 * res, 3 * res, 5 * res, 5 * res * (1 + 1e-6)):
            for _ in range(25):
                ci, cj = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                assert len(list(gm.circle(gm.position((ci, cj)), radius))) == O.circle_count(g, ci, cj, radius), (radius, ci, cj)


def _random_case(seed):
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(6, 34)), int(rng.integers(6, 34))
    res = float(rng.choice([0.03, 0.05, 0.1]))
    pos = (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))) if seed % 3 else (0.0, 0.0)
    x = np.linspace(0, 3, rows)[:, None] + np.linspace(0, 2, cols)[None, :]
    elev = (0.05 * np.sin(3 * x) + rng.normal(0, 0.01, (rows, cols))).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):  # kerbs and pits
        i0, j0 = int(rng.integers(0, rows)), int(rng.integers(0, cols))
        elev[i0:i0 + int(rng.integers(1, 6)), j0:j0 + int(rng.integers(1, 6))] += float(rng.choice([-0.3, 0.2, 0.5]))
    if seed % 4 == 0:
        elev[rng.random((rows, cols)) < 0.03] = np.nan

    def score():
        s = rng.random((rows, cols)).astype(np.float32)
        s[rng.random((rows, cols)) < rng.choice([0.02, 0.15, 0.4])] = 0.0
        return s
    slope, step, rough = score(), score(), score()
    trav = ((slope + step + rough) * np.float32(1 / 3)).astype(np.float32)
    if seed % 5 == 0:
        trav[rng.random((rows, cols)) < 0.05] = np.nan
    return rows, cols, res, pos, elev, slope, step, rough, trav


@pytest.mark.parametrize("seed", range(24))
def test_footprint_matches_the_c_oracle(O, seed):
    rows, cols, res, pos, elev, slope, step, rough, trav = _random_case(seed)
    rng = np.random.default_rng(1000 + seed)
    rmin = float(rng.choice([0.0, 1.2, 2.0, 3.0])) * res * (1.0 if seed % 2 else 1 + 1e-6)
    off = float(rng.choice([0.6, 1.5, 2.0])) * res
    check_rough = bool(seed % 3 == 0)
    max_gap = float(rng.choice([0.1, 0.3]))
    crit = float(rng.choice([0.05, 0.12]))
    g = O.geom(rows, cols, res, pos)
    p = O.default_params(fp_radius=rmin, fp_offset=off, fp_default=0.3, fp_max_gap=max_gap, fp_critical_step=crit,
                         fp_check_roughness=int(check_rough))
    # the oracle's layers are [column][row] flat (column-major matrices); the restatement indexes [row, column]
    layers = {"traversability_slope": slope.T, "traversability_step": step.T, "traversability_roughness": rough.T,
              "traversability": trav.T}
    want = O.footprint(g, p, np.ascontiguousarray(elev.T), {k: np.ascontiguousarray(v) for k, v in layers.items()},
                       want_memo=True)
    fp_c, memo_c = want
    ref = TraversabilityMapRef(GridMapRef(rows, cols, res, pos), elev, slope, step, rough, trav, default=0.3, max_gap=max_gap,
                               critical_step=crit, check_roughness=check_rough)
    mine = ref.traversability_footprint(rmin, off)
    got = np.asarray(fp_c, np.float32).reshape(cols, rows).T
    assert np.array_equal(np.isnan(mine), np.isnan(got))
    assert np.array_equal(mine[~np.isnan(mine)].view(np.uint32), got[~np.isnan(got)].view(np.uint32)), \
        (seed, float(np.nanmax(np.abs(mine - got))))
    # the memo layers: the reference fills them lazily (only for the cells a spiral reached), the oracle for every cell
    for name, lazy in (("slope_footprint", ref.slope_fp), ("step_footprint", ref.step_fp), ("roughness_footprint", ref.rough_fp)):
        full = np.asarray(memo_c[name], np.float32).reshape(cols, rows).T
        seen = np.isfinite(lazy)
        if name == "roughness_footprint" and not check_rough:
            assert not seen.any()
            continue
        assert np.array_equal(lazy[seen], full[seen]), (seed, name)
