"""footprint/check_robot_inclination (TraversabilityMap.cpp:114): checkInclination (:748-762) on the layer robot_slope, alone
and inside both path checks (:366-370, :390-394, :526-528, :553-557); and FootprintPath.compute_untraversable_polygon for
polygons (:592-645).  The C oracle against the pure-Python restatements (CPU), the HIP kernels against the oracle through
the C-ABI (GPU)."""
import numpy as np
import pytest

from tests.test_paths import py_check_path, py_inclination, random_paths
from tests.test_polygon import (POINTS_XYZ, chain_layers, gpu_setup, py_check_polygon_path, random_pose_paths, terrain,
                                untraversable_mask)


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    return capi


def robot_slope_layer(rng, g, zero_fraction=0.004, nan_fraction=0.05):
    """What an inclination estimator leaves behind: mostly 1, a few 0 (too steep for the robot), holes."""
    rs = np.ones(g.rows * g.cols, np.float32)
    rs[rng.random(rs.shape) < 0.3] = 0.5
    rs[rng.random(rs.shape) < zero_fraction] = 0.0
    rs[rng.random(rs.shape) < nan_fraction] = np.nan
    rs[rng.random(rs.shape) < 0.002] = -0.0  # == 0.0 as well
    return rs


def random_segments(rng, g, count):
    lo = np.array([g.pos_x - 0.5 * g.len_x, g.pos_y - 0.5 * g.len_y])
    hi = np.array([g.pos_x + 0.5 * g.len_x, g.pos_y + 0.5 * g.len_y])
    a = rng.uniform(lo, hi, (count, 2))
    b = rng.uniform(lo, hi, (count, 2))
    short = rng.random(count) < 0.5
    b[short] = a[short] + rng.uniform(-0.4, 0.4, (int(short.sum()), 2))
    same = rng.random(count) < 0.2
    b[same] = a[same]  # start == end: a single cell
    out = rng.random(count) < 0.05
    b[out] += 3.0 * g.len_x
    return np.hstack([a, b])


def test_oracle_inclination_against_python(oracle):
    rng = np.random.default_rng(3)
    g = oracle.geom(57, 43, 0.1, (1.25, -0.75))
    rs = robot_slope_layer(rng, g, zero_fraction=0.01)
    seg = random_segments(rng, g, 2000)
    ok, st = oracle.check_inclination(g, rs, seg)
    for k, (sx, sy, ex, ey) in enumerate(seg):
        assert (bool(ok[k]), bool(st[k])) == py_inclination(g, rs, float(sx), float(sy), float(ex), float(ey)), (k, seg[k])
    assert 200 < ok.sum() < len(seg) - 200 and 0 < st.sum() < 400
    # an all-NaN layer: nothing is valid and NaN == 0.0 is false, so every in-map segment passes
    ok, st = oracle.check_inclination(g, np.full_like(rs, np.nan), seg)
    assert np.array_equal(ok, st == 0)


def test_oracle_paths_with_inclination_against_python(oracle):
    rng = np.random.default_rng(6)
    g = oracle.geom(57, 43, 0.1, (1.25, -0.75))
    fp = rng.uniform(0.2, 1.0, size=g.rows * g.cols).astype(np.float32)
    fp[rng.random(fp.shape) < 0.01] = 0.0
    rs = robot_slope_layer(rng, g)
    paths = random_paths(rng, g, 400)
    safe, trav, st = oracle.check_circular_paths(g, fp, 0.3, paths, robot_slope=rs)
    plain = oracle.check_circular_paths(g, fp, 0.3, paths)
    for k, p in enumerate(paths):
        want = py_check_path(g, fp, 0.3, p, rs)
        assert (bool(safe[k]), int(st[k])) == (want[0], want[2]), (k, p)
        assert trav[k] == want[1] or (np.isnan(trav[k]) and np.isnan(want[1])), (k, trav[k], want[1])
    assert safe.sum() < plain[0].sum() and not (safe & ~plain[0]).any()  # the option only ever removes paths
    assert safe.sum() > 20
    # a layer without zeros changes nothing but the single poses off the map (atPosition throws there: status 1)
    ones = np.ones_like(rs)
    safe1, trav1, st1 = oracle.check_circular_paths(g, fp, 0.3, paths, robot_slope=ones)
    single_out = np.array([len(p) == 1 and st1[k] == 1 for k, p in enumerate(paths)])
    assert np.array_equal(safe1[~single_out], plain[0][~single_out]) and np.array_equal(trav1[~single_out], plain[1][~single_out])
    assert not safe1[single_out].any()


def test_oracle_polygon_paths_with_inclination_against_python(oracle):
    rows, cols, res = 60, 45, 0.05
    g = oracle.geom(rows, cols, res, (0.3, -1.1))
    p = oracle.default_params(fp_default=0.3)
    elev = terrain(rows, cols, seed=15, boxes=6)
    layers = chain_layers(oracle, g, p, elev)
    untrav = untraversable_mask(oracle, g, p, elev, layers)
    pts = np.array(POINTS_XYZ) * 0.5
    rng = np.random.default_rng(22)
    rs = robot_slope_layer(rng, g, zero_fraction=0.01)
    paths, cons = random_pose_paths(g, rng, 120, scale=0.5)
    paths.append(np.zeros((0, 7)))
    cons = np.append(cons, 0).astype(np.uint8)
    args = (g, p, elev, layers["traversability_slope"], layers["traversability_step"], layers["traversability_roughness"],
            layers["traversability"], paths, pts, cons)
    safe, val, area, st = oracle.check_polygon_paths(*args, robot_slope=rs)
    plain = oracle.check_polygon_paths(*args)
    for k, path in enumerate(paths):
        want = py_check_polygon_path(g, untrav, layers["traversability"], 0.3, path, [tuple(v) for v in pts], bool(cons[k]), rs)
        assert (bool(safe[k]), val[k], area[k], int(st[k])) == want, (k, want)
    assert 5 < safe.sum() < plain[0].sum() and (st == 1).any() and st[-1] == 2


@pytest.mark.gpu
def test_gpu_inclination_against_oracle(oracle):
    from traversability_estimation_amd import capi, synth
    from tests.helpers import to_te_params
    capi.load()
    rng = np.random.default_rng(12)
    rows, cols, res = 300, 260, 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=21, amplitude=0.15), 14, seed=22)
    r = synth.benchmark_radius(3, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=0.3, fp_offset=0.15)
    g = oracle.geom(rows, cols, res, (4.0, -2.5))
    rs = robot_slope_layer(rng, g, zero_fraction=0.0015)
    seg = random_segments(rng, g, 20000)
    paths = random_paths(rng, g, 5000)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, (4.0, -2.5))
        with pytest.raises(capi.TeError, match="robot_slope"):
            ctx.check_inclination(seg[:4])  # the layer does not exist yet
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        fp = ctx.download("traversability_footprint")
        plain = ctx.check_footprint_paths(paths)
        ctx.set_check_robot_inclination(True)
        with pytest.raises(capi.TeError, match="robot_slope"):
            ctx.check_footprint_paths(paths)
        ctx.upload_layer("robot_slope", rs)
        assert np.array_equal(ctx.download("robot_slope").view(np.uint32), rs.view(np.uint32))
        ok, st_seg = ctx.check_inclination(seg)
        assert ctx.check_inclination(np.zeros((0, 4)))[0].size == 0
        safe, trav, st = ctx.check_footprint_paths(paths)
        ctx.set_check_robot_inclination(False)
        again = ctx.check_footprint_paths(paths)
        # the layer survives in GridMap buffer order too (te_upload_layer_circular), e.g. from a moved map
        si, sj = 17, 203
        buf = np.roll(np.roll(rs.reshape(cols, rows), sj, axis=0), si, axis=1)
        ctx.upload_layer_circular("robot_slope", buf, (si, sj))
        assert np.array_equal(ctx.download("robot_slope").view(np.uint32), rs.view(np.uint32))
        # te_set_layer_present: a map that comes without the layer declares it absent again; asking for the device
        # pointer alone does not make it present
        ctx.set_layer_present("robot_slope", False)
        with pytest.raises(capi.TeError, match="robot_slope"):
            ctx.check_inclination(seg[:4])
        ctx.device_ptr("robot_slope")
        with pytest.raises(capi.TeError, match="robot_slope"):
            ctx.check_inclination(seg[:4])
        ctx.set_layer_present("robot_slope", True)
        ok2, _ = ctx.check_inclination(seg)
        assert np.array_equal(ok2, ok)
        with pytest.raises(capi.TeError):
            ctx.set_layer_present("elevation", False)
    want_ok, want_st = oracle.check_inclination(g, rs, seg)
    assert np.array_equal(ok, want_ok) and np.array_equal(st_seg, want_st)
    assert 2000 < ok.sum() < len(seg) - 2000
    want = oracle.check_circular_paths(g, fp, op.fp_default, paths, robot_slope=rs)
    assert np.array_equal(safe, want[0]) and np.array_equal(trav, want[1]) and np.array_equal(st, want[2])
    assert 100 < safe.sum() < plain[0].sum()
    for a, b in zip(plain, again):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_polygon_paths_with_inclination(capi, oracle):
    rows, cols, res = 160, 140, 0.05
    elev = terrain(rows, cols, seed=31, boxes=12)
    ctx, g, op, layers = gpu_setup(capi, oracle, rows, cols, res, (2.0, -3.0), elev, fp_default=0.3)
    rng = np.random.default_rng(5)
    rs = robot_slope_layer(rng, g, zero_fraction=0.003)
    paths, cons = random_pose_paths(g, rng, 5000)
    paths.append(np.zeros((0, 7)))
    cons = np.append(cons, 1).astype(np.uint8)
    with ctx:
        plain = ctx.check_polygon_footprint_paths(paths, POINTS_XYZ, cons)
        ctx.upload_layer("robot_slope", rs)
        ctx.set_check_robot_inclination(True)
        got = ctx.check_polygon_footprint_paths(paths, POINTS_XYZ, cons)
    args = (g, op, elev, layers["traversability_slope"], layers["traversability_step"], layers["traversability_roughness"],
            layers["traversability"], paths, POINTS_XYZ, cons)
    want = oracle.check_polygon_paths(*args, robot_slope=rs)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[3], want[3])
    assert np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64))
    assert np.array_equal(got[2].view(np.uint64), want[2].view(np.uint64))
    assert 20 < got[0].sum() < plain[0].sum() and (got[3] == 1).any()


# ---------------------------------------------------------------- compute_untraversable_polygon (:592-645)
def py_polygon_hull(g, untrav, trav, default, verts):
    """isTraversable(polygon, computeUntraversablePolygon=True): every cell visited, the untraversable positions collected."""
    from tests.test_polygon import py_bound, py_hull, py_inside
    cx = lambda i: (g.pos_x + (0.5 * g.len_x - 0.5 * g.res)) + g.res * float(-i)  # noqa: E731
    cy = lambda j: (g.pos_y + (0.5 * g.len_y - 0.5 * g.res)) + g.res * float(-j)  # noqa: E731
    xs, ys = [v[0] for v in verts], [v[1] for v in verts]
    tlx, tly = py_bound(max(xs), g.len_x, g.pos_x), py_bound(max(ys), g.len_y, g.pos_y)
    brx, bry = py_bound(min(xs), g.len_x, g.pos_x), py_bound(min(ys), g.len_y, g.pos_y)
    idx = lambda x, half, pos, n: min(max(int(-(((x - half) - pos) / g.res)), 0), n - 1)  # noqa: E731
    ti, bi = idx(tlx, 0.5 * g.len_x, g.pos_x, g.rows), idx(brx, 0.5 * g.len_x, g.pos_x, g.rows)
    tj, bj = idx(tly, 0.5 * g.len_y, g.pos_y, g.cols), idx(bry, 0.5 * g.len_y, g.pos_y, g.cols)
    n, t, bad = 0, 0.0, []
    for a in range(ti, bi + 1):
        for b in range(tj, bj + 1):
            if not py_inside(verts, cx(a), cy(b)):
                continue
            o = b * g.rows + a
            if untrav[o]:
                bad.append((cx(a), cy(b)))
            else:
                n += 1
                t += float(trav[o]) if np.isfinite(trav[o]) else default
    if bad:
        return False, 0.0, py_hull(bad)
    if n == 0:
        return default != 0.0, default, []
    return True, t / n, []


def hull_polygons(g, rng, count):
    """Footprint-sized polygons, slivers one cell wide (all untraversable cells in one or two rows) and tiny ones."""
    from tests.test_polygon import random_polygons
    polys = random_polygons(g, rng, count)
    for k in range(count // 2):
        cx = g.pos_x + (rng.random() - 0.5) * g.len_x * 0.9
        cy = g.pos_y + (rng.random() - 0.5) * g.len_y * 0.9
        w, h = (g.res * rng.uniform(0.6, 2.2), g.res * rng.uniform(3, 30)) if k % 2 else (g.res * rng.uniform(3, 30), g.res * rng.uniform(0.6, 2.2))
        polys.append(np.array([[cx - w, cy - h], [cx + w, cy - h], [cx + w, cy + h], [cx - w, cy + h]]) * 0.5 + np.array([cx, cy]) * 0.5)
    return polys


def test_oracle_untraversable_hull_against_python(oracle):
    rows, cols, res = 60, 45, 0.05
    g = oracle.geom(rows, cols, res, (0.3, -1.1))
    p = oracle.default_params(fp_default=0.3)
    elev = terrain(rows, cols, seed=15, boxes=14)
    layers = chain_layers(oracle, g, p, elev)
    untrav = untraversable_mask(oracle, g, p, elev, layers)
    args = (g, p, elev, layers["traversability_slope"], layers["traversability_step"], layers["traversability_roughness"],
            layers["traversability"])
    sizes = []
    for poly in hull_polygons(g, np.random.default_rng(2), 160):
        ok, val, hull = oracle.polygon_untraversable_hull(*args, poly)
        want = py_polygon_hull(g, untrav, layers["traversability"], 0.3, [tuple(float(c) for c in v) for v in poly])
        assert (ok, val) == want[:2]
        assert hull.shape == (len(want[2]), 2) and np.array_equal(hull, np.array(want[2], dtype=np.float64).reshape(-1, 2))
        sizes.append(len(hull))
        # the plain query agrees on everything but the polygon
        ok2, val2 = oracle.polygons_traversable(*args, [poly])
        assert (bool(ok2[0]), val2[0]) == (ok, val)
    assert sizes.count(0) > 20 and any(0 < s <= 3 for s in sizes) and max(sizes) >= 6


@pytest.mark.gpu
def test_gpu_untraversable_hull_against_oracle(capi, oracle):
    rows, cols, res = 160, 140, 0.05
    elev = terrain(rows, cols, seed=31, boxes=20)
    ctx, g, op, layers = gpu_setup(capi, oracle, rows, cols, res, (2.0, -3.0), elev, fp_default=0.3)
    polys = hull_polygons(g, np.random.default_rng(9), 400)
    polys.append(np.array([[g.pos_x - g.len_x, g.pos_y - g.len_y], [g.pos_x + g.len_x, g.pos_y - g.len_y],
                           [g.pos_x + g.len_x, g.pos_y + g.len_y], [g.pos_x - g.len_x, g.pos_y + g.len_y]]))  # the whole map
    with ctx:
        got = [ctx.polygon_untraversable_hull(q) for q in polys]
        with pytest.raises(capi.TeError, match="room for"):
            ctx.polygon_untraversable_hull(polys[-1], cap=2)
    args = (g, op, elev, layers["traversability_slope"], layers["traversability_step"], layers["traversability_roughness"],
            layers["traversability"])
    sizes = []
    for q, (ok, val, hull) in zip(polys, got):
        w_ok, w_val, w_hull = oracle.polygon_untraversable_hull(*args, q)
        assert ok == w_ok and np.float64(val).view(np.uint64) == np.float64(w_val).view(np.uint64)
        assert hull.shape == w_hull.shape and np.array_equal(hull.view(np.uint64), w_hull.view(np.uint64))
        sizes.append(len(hull))
    assert sizes.count(0) > 30 and any(0 < s <= 3 for s in sizes) and max(sizes) >= 8
