"""checkFootprintPath for circular footprints (TraversabilityMap.cpp:320-462): the C oracle against an independent
pure-Python restatement on small maps (CPU), and the HIP kernel against the oracle (GPU, through the C-ABI)."""
import math

import numpy as np
import pytest


def py_index(g, x, y):
    """getIndexFromPosition + checkIfPositionWithinMap (grid_map_core), plain Python doubles."""
    tx = -((x - g.pos_x) - 0.5 * g.len_x)
    ty = -((y - g.pos_y) - 0.5 * g.len_y)
    inside = tx >= 0.0 and ty >= 0.0 and tx < g.len_x and ty < g.len_y
    i = int(-(((x - 0.5 * g.len_x) - g.pos_x) / g.res))
    j = int(-(((y - 0.5 * g.len_y) - g.pos_y) / g.res))
    return inside and 0 <= i < g.rows and 0 <= j < g.cols, inside, i, j


def py_line(si, sj, ei, ej):
    """grid_map::LineIterator: list of cells from (si, sj) to (ei, ej)."""
    dx, dy = abs(ei - si), abs(ej - sj)
    inc1i = inc2i = 1 if ei >= si else -1
    inc1j = inc2j = 1 if ej >= sj else -1
    if dx >= dy:
        inc1i, inc2j, den, num, numadd, n = 0, 0, dx, dx // 2, dy, dx + 1
    else:
        inc2i, inc1j, den, num, numadd, n = 0, 0, dy, dy // 2, dx, dy + 1
    cells, i, j = [], si, sj
    for _ in range(n):
        cells.append((i, j))
        num += numadd
        if num >= den:
            num -= den
            i += inc1i
            j += inc1j
        i += inc2i
        j += inc2j
    return cells


def py_inclination(g, rs, sx, sy, ex, ey):
    """checkInclination(start, end) :748-762 -> (ok, outside)."""
    if ex == sx and ey == sy:
        ok, inside, i, j = py_index(g, sx, sy)
        if not ok:
            return False, True
        return not (float(rs[j * g.rows + i]) == 0.0), False
    ok_s, _, si, sj = py_index(g, sx, sy)
    ok_e, _, ei, ej = py_index(g, ex, ey)
    if not (ok_s and ok_e):
        return False, True
    for a, b in py_line(si, sj, ei, ej):  # from the start index to the end index
        v = float(rs[b * g.rows + a])
        if not math.isfinite(v):
            continue
        if v == 0.0:
            return False, False
    return True, False


def py_check_path(g, fp, default, poses, rs=None):
    """checkCircularFootprintPath on a complete footprint layer (memo branch of isTraversable); rs: the layer
    robot_slope when footprint/check_robot_inclination is set."""
    n = len(poses)
    if n == 0:
        return False, 0.0, 2
    res_trav, length_path = 0.0, 0.0
    ex = ey = 0.0
    for i in range(n):
        sx, sy = ex, ey
        ex, ey = float(poses[i][0]), float(poses[i][1])
        if rs is not None and (n == 1 or i > 0):
            good, outside = py_inclination(g, rs, ex, ey, ex, ey) if n == 1 else py_inclination(g, rs, sx, sy, ex, ey)
            if not good:
                return False, 0.0, int(outside)
        if n == 1:
            ok, inside, ci, cj = py_index(g, ex, ey)
            t = float(fp[cj * g.rows + ci]) if inside else default
            if not (t != 0.0):
                return False, 0.0, 0
            res_trav = t
        if n > 1 and i > 0:
            ok_s, _, si, sj = py_index(g, sx, sy)
            ok_e, _, ei, ej = py_index(g, ex, ey)
            if not (ok_s and ok_e):
                return False, 0.0, 1
            cells = py_line(ei, ej, si, sj)[::4]  # every fourth cell (nSkip = 3)
            vals = [float(fp[b * g.rows + a]) for a, b in cells]
            s, cnt = 0.0, 0
            for v in vals:
                if not (v != 0.0):
                    return False, 0.0, 0
                s += v
                cnt += 1
            t = s / cnt
            seg = math.sqrt((ex - sx) * (ex - sx) + (ey - sy) * (ey - sy))
            if i > 1:
                prev = length_path
                length_path += seg
                res_trav = (seg * t + prev * res_trav) / length_path
            else:
                length_path = seg
                res_trav = t
    return True, res_trav, 0


def random_paths(rng, g, count, outside_fraction=0.05):
    lo_x, hi_x = g.pos_x - 0.5 * g.len_x, g.pos_x + 0.5 * g.len_x
    lo_y, hi_y = g.pos_y - 0.5 * g.len_y, g.pos_y + 0.5 * g.len_y
    paths = []
    for _ in range(count):
        n = int(rng.integers(1, 7))
        p = np.stack([rng.uniform(lo_x, hi_x, n), rng.uniform(lo_y, hi_y, n)], axis=1)
        if rng.random() < outside_fraction:
            p[int(rng.integers(0, n))] += 3.0 * g.len_x
        if rng.random() < 0.1 and n > 1:
            p[1] = p[0]  # a zero-length segment
        paths.append(p)
    paths.append(np.zeros((0, 2)))  # "This path has no poses to check"
    return paths


def test_oracle_paths_against_python(oracle):
    rng = np.random.default_rng(5)
    g = oracle.geom(57, 43, 0.1, (1.25, -0.75))
    fp = rng.uniform(0.2, 1.0, size=g.rows * g.cols).astype(np.float32)
    fp[rng.random(fp.shape) < 0.02] = 0.0  # untraversable footprints
    paths = random_paths(rng, g, 300)
    safe, trav, st = oracle.check_circular_paths(g, fp, 0.3, paths)
    n_safe = 0
    for k, p in enumerate(paths):
        want = py_check_path(g, fp, 0.3, p)
        assert (bool(safe[k]), int(st[k])) == (want[0], want[2]), (k, p)
        assert trav[k] == want[1] or (math.isnan(trav[k]) and math.isnan(want[1])), (k, trav[k], want[1])
        n_safe += want[0]
    assert 20 < n_safe < len(paths) - 20  # both outcomes are exercised
    # the layer value is taken as is: a uniform layer gives exactly that value on every safe path
    fp[:] = 0.625
    safe, trav, st = oracle.check_circular_paths(g, fp, 0.0, paths)
    assert all(abs(t - 0.625) < 1e-12 or math.isnan(t) for t, s in zip(trav, safe) if s)  # NaN: all segments of length 0
    single_outside = [np.array([[1e3, 1e3]])]
    assert oracle.check_circular_paths(g, fp, 0.3, single_outside)[1][0] == 0.3  # traversabilityDefault_
    assert not oracle.check_circular_paths(g, fp, 0.0, single_outside)[0][0]


@pytest.mark.gpu
def test_gpu_paths_against_oracle(oracle):
    from traversability_estimation_amd import capi, synth
    from tests.helpers import to_te_params
    capi.load()
    rng = np.random.default_rng(11)
    rows, cols, res = 300, 260, 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=21, amplitude=0.15), 14, seed=22)
    r = synth.benchmark_radius(3, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=0.3,
                               fp_offset=0.15)
    g = oracle.geom(rows, cols, res, (4.0, -2.5))
    with capi.Context(0) as ctx:
        with pytest.raises(capi.TeError):
            ctx.check_footprint_paths([np.zeros((2, 2))])  # nothing computed yet
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, (4.0, -2.5))
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        fp = ctx.download("traversability_footprint")
        assert (fp == 0).sum() > 100 and np.isfinite(fp).all()
        paths = random_paths(rng, g, 5000)
        safe, trav, st = ctx.check_footprint_paths(paths)
        assert ctx.check_footprint_paths([])[0].size == 0
    want_safe, want_trav, want_st = oracle.check_circular_paths(g, fp, op.fp_default, paths)
    assert np.array_equal(st, want_st)
    assert np.array_equal(safe, want_safe)
    assert np.array_equal(trav, want_trav)  # same double arithmetic: bit-identical
    assert 100 < safe.sum() < len(paths) - 100
