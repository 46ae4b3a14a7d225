"""Polygon footprints: TraversabilityMap::isTraversable(polygon) (TraversabilityMap.cpp:586-645) and
traversabilityFootprint(footprintYaw) (:239-305).  The C oracle against an independent pure-Python restatement on
small maps (CPU); the HIP kernels against the oracle, bit for bit (GPU, through the C-ABI)."""
import math

import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, to_te_params

FOOTPRINT = [[0.45, 0.30], [0.45, -0.30], [-0.45, -0.30], [-0.45, 0.30]]  # robot_footprint_parameter.yaml:3


def terrain(rows, cols, seed, holes=0.02, boxes=6):
    from traversability_estimation_amd import synth
    e = synth.perlin_elevation(rows, cols, seed=seed)
    if boxes and rows > 8 and cols > 8:
        e = synth.with_steps(e, boxes, seed=seed + 1)
    if holes:
        e = synth.with_holes(e, holes, seed=seed + 2)
    return np.ascontiguousarray(e, dtype=np.float32).reshape(-1)


def chain_layers(oracle, g, p, elev):
    return oracle.chain(g, p, elev)


# ---------------------------------------------------------------- pure-Python restatement
def py_bound(position, length, mappos):
    shifted = position - mappos + 0.5 * length
    eps = 10.0 * 2.220446049250313e-16
    if abs(position) > 1.0:
        eps *= abs(position)
    if shifted <= 0:
        shifted = eps
    elif shifted >= length:
        shifted = length - eps
    return shifted + mappos - 0.5 * length


def py_inside(verts, px, py):
    cross = 0
    n = len(verts)
    j = n - 1
    for i in range(n):
        xi, yi = verts[i]
        xj, yj = verts[j]
        if ((yi > py) != (yj > py)) and (px < (xj - xi) * (py - yi) / (yj - yi) + xi):
            cross += 1
        j = i
    return cross % 2 == 1


def py_polygon(g, untrav, trav, default, verts):
    cx = lambda i: (g.pos_x + (0.5 * g.len_x - 0.5 * g.res)) + g.res * float(-i)  # noqa: E731
    cy = lambda j: (g.pos_y + (0.5 * g.len_y - 0.5 * g.res)) + g.res * float(-j)  # noqa: E731
    xs, ys = [v[0] for v in verts], [v[1] for v in verts]
    tlx, tly = py_bound(max(xs), g.len_x, g.pos_x), py_bound(max(ys), g.len_y, g.pos_y)
    brx, bry = py_bound(min(xs), g.len_x, g.pos_x), py_bound(min(ys), g.len_y, g.pos_y)
    idx = lambda x, half, pos, n: min(max(int(-(((x - half) - pos) / g.res)), 0), n - 1)  # noqa: E731
    ti, bi = idx(tlx, 0.5 * g.len_x, g.pos_x, g.rows), idx(brx, 0.5 * g.len_x, g.pos_x, g.rows)
    tj, bj = idx(tly, 0.5 * g.len_y, g.pos_y, g.cols), idx(bry, 0.5 * g.len_y, g.pos_y, g.cols)
    n, t = 0, 0.0
    for a in range(ti, bi + 1):
        for b in range(tj, bj + 1):
            if not py_inside(verts, cx(a), cy(b)):
                continue
            o = b * g.rows + a
            if untrav[o]:
                return False, 0.0
            n += 1
            t += float(trav[o]) if np.isfinite(trav[o]) else default
    if n == 0:
        return default != 0.0, default
    return True, t / n


def untraversable_mask(oracle, g, p, elev, layers):
    _, memo = oracle.footprint(g, p, elev, layers, want_memo=True)
    return (memo["slope_footprint"] == 0) | (memo["step_footprint"] == 0) | (memo["roughness_footprint"] == 0)


def random_polygons(g, rng, count):
    """Triangles .. octagons (convex or not) around points in and around the map, some tiny, some far outside."""
    polys = []
    for k in range(count):
        cx = g.pos_x + (rng.random() - 0.5) * g.len_x * 1.3
        cy = g.pos_y + (rng.random() - 0.5) * g.len_y * 1.3
        nv = int(rng.integers(3, 9))
        scale = g.res * (0.2 if k % 7 == 0 else rng.uniform(1.5, 9.0))
        ang = np.sort(rng.random(nv)) * 2 * math.pi
        rad = scale * rng.uniform(0.4, 1.0, nv)
        polys.append(np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1))
    # cell-aligned rectangles: every edge passes through cell centres or cell borders
    for k in range(count // 4):
        i, j = int(rng.integers(0, g.rows)), int(rng.integers(0, g.cols))
        cx = (g.pos_x + (0.5 * g.len_x - 0.5 * g.res)) + g.res * float(-i)
        cy = (g.pos_y + (0.5 * g.len_y - 0.5 * g.res)) + g.res * float(-j)
        hx, hy = g.res * int(rng.integers(1, 5)) * (1.0 if k % 2 else 0.5), g.res * int(rng.integers(1, 5))
        polys.append(np.array([[cx + hx, cy + hy], [cx + hx, cy - hy], [cx - hx, cy - hy], [cx - hx, cy + hy]]))
    return polys


@pytest.mark.parametrize("shape", [(40, 31), (9, 50)])
def test_oracle_matches_python_restatement(oracle, shape):
    rows, cols = shape
    res = 0.04
    g = oracle.geom(rows, cols, res, (0.3, -1.1))
    p = oracle.default_params(fp_default=0.3)
    elev = terrain(rows, cols, seed=5)
    layers = chain_layers(oracle, g, p, elev)
    untrav = untraversable_mask(oracle, g, p, elev, layers)
    assert untrav.any() and not untrav.all()
    rng = np.random.default_rng(3)
    polys = random_polygons(g, rng, 60)
    ok, val = oracle.polygons_traversable(g, p, elev, layers["traversability_slope"], layers["traversability_step"],
                                          layers["traversability_roughness"], layers["traversability"], polys)
    n_true = 0
    for k, poly in enumerate(polys):
        want_ok, want_val = py_polygon(g, untrav, layers["traversability"], 0.3, [tuple(v) for v in poly])
        assert bool(ok[k]) == want_ok, k
        assert val[k] == want_val, (k, val[k], want_val)
        n_true += want_ok
    assert 5 < n_true < len(polys) - 5  # both outcomes are exercised
    # the footprint layers: yaw 0 turns nothing, and the layer is the per-cell polygon query
    pts = np.array(FOOTPRINT) * (res / 0.15)
    tx, trot = oracle.polygon_footprint(g, p, elev, layers["traversability_slope"], layers["traversability_step"],
                                        layers["traversability_roughness"], layers["traversability"], pts, 0.0)
    assert np.array_equal(tx.view(np.uint32), trot.view(np.uint32))
    for (i, j) in [(0, 0), (rows - 1, cols - 1), (rows // 2, cols // 3), (3, cols - 2)]:
        cx = (g.pos_x + (0.5 * g.len_x - 0.5 * g.res)) + g.res * float(-i)
        cy = (g.pos_y + (0.5 * g.len_y - 0.5 * g.res)) + g.res * float(-j)
        want_ok, want_val = py_polygon(g, untrav, layers["traversability"], 0.3, [(x + cx, y + cy) for x, y in pts])
        assert tx[j * rows + i] == np.float32(want_val if want_ok else 0.0)


def test_rotation_is_eigens_quaternion_matrix(oracle):
    """yaw -> (cos(yaw/2), 0, 0, sin(yaw/2)) -> Eigen toRotationMatrix; pi/2 is the node's default footprint_yaw."""
    pts = np.array(FOOTPRINT)
    out = oracle.rotate_footprint(pts, math.pi / 2)
    assert np.allclose(out, np.stack([-pts[:, 1], pts[:, 0]], axis=1), atol=1e-15)
    assert np.array_equal(oracle.rotate_footprint(pts, 0.0), pts)
    yaw = 0.7
    w, z = math.cos(yaw / 2), math.sin(yaw / 2)
    r00, r01 = 1.0 - (2 * z) * z, -((2 * z) * w)
    assert oracle.rotate_footprint([[1.0, 2.0]], yaw)[0, 0] == r00 * 1.0 + r01 * 2.0


# ---------------------------------------------------------------- on the device
@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    return capi


def gpu_setup(capi, oracle, rows, cols, res, pos, elev, **over):
    op = oracle.default_params(**over)
    g = oracle.geom(rows, cols, res, pos)
    ctx = capi.Context(0)
    ctx.set_params(to_te_params(capi, op))
    ctx.set_geometry(rows, cols, 1, res, pos)
    ctx.upload_elevation(elev)
    ctx.run_chain(capi.RUN_FOOTPRINT)
    ctx.sync()
    layers = {k: ctx.download(k) for k in OUT_LAYERS}
    return ctx, g, op, layers


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(rows=100, cols=133, res=0.03, pts=FOOTPRINT, yaw=math.pi / 2),
    dict(rows=150, cols=70, res=0.05, pts=FOOTPRINT, yaw=0.4),
    dict(rows=64, cols=64, res=0.05, pts=[[0.3, 0.0], [-0.2, 0.25], [-0.2, -0.25]], yaw=-2.2),
    dict(rows=61, cols=90, res=0.04, pts=[[0.3, 0.3], [0.3, -0.3], [0.0, -0.3], [0.0, 0.0], [-0.3, 0.0], [-0.3, 0.3]], yaw=1.0),
    dict(rows=33, cols=20, res=0.1, pts=[[0.01, 0.01], [0.01, -0.01], [-0.01, -0.01]], yaw=0.3),  # covers no cell centre
    dict(rows=5, cols=7, res=0.1, pts=FOOTPRINT, yaw=0.9),  # footprint larger than the map
])
def test_polygon_footprint_layers(capi, oracle, case):
    rows, cols, res = case["rows"], case["cols"], case["res"]
    elev = terrain(rows, cols, seed=rows + cols)
    ctx, g, op, layers = gpu_setup(capi, oracle, rows, cols, res, (0.7, -0.2), elev, fp_default=0.3)
    with ctx:
        ctx.run_polygon_footprint(case["pts"], case["yaw"])
        ctx.sync()
        got_x, got_rot = ctx.download("traversability_x"), ctx.download("traversability_rot")
    # the oracle works on the layers the device produced, so this isolates the polygon pass
    want_x, want_rot = oracle.polygon_footprint(g, op, elev, layers["traversability_slope"], layers["traversability_step"],
                                                layers["traversability_roughness"], layers["traversability"],
                                                case["pts"], case["yaw"])
    assert np.array_equal(got_x.view(np.uint32), want_x.view(np.uint32)), int((got_x != want_x).sum())
    assert np.array_equal(got_rot.view(np.uint32), want_rot.view(np.uint32)), int((got_rot != want_rot).sum())
    if rows > 30 and max(abs(v) for pt in case["pts"] for v in pt) > res:
        assert (want_x == 0).any() and (want_x > 0).any()


@pytest.mark.gpu
def test_polygon_batch(capi, oracle):
    rows, cols, res = 120, 90, 0.05
    elev = terrain(rows, cols, seed=77, boxes=40)
    ctx, g, op, layers = gpu_setup(capi, oracle, rows, cols, res, (-1.0, 2.0), elev, fp_default=0.3)
    polys = random_polygons(g, np.random.default_rng(9), 400)
    with ctx:
        ok, val = ctx.polygons_traversable(polys)
        empty_ok, empty_val = ctx.polygons_traversable([])
        with pytest.raises(capi.TeError, match="no vertices"):
            ctx.polygons_traversable([polys[0], np.zeros((0, 2))])
        with pytest.raises(capi.TeError, match="not finite"):
            ctx.polygons_traversable([[[0.0, 0.0], [float("nan"), 1.0], [1.0, 1.0]]])
        with pytest.raises(capi.TeError, match="footprint points"):
            ctx.run_polygon_footprint(np.zeros((40, 2)), 0.0)
    assert len(empty_ok) == 0 and len(empty_val) == 0
    want_ok, want_val = oracle.polygons_traversable(g, op, elev, layers["traversability_slope"], layers["traversability_step"],
                                                    layers["traversability_roughness"], layers["traversability"], polys)
    assert np.array_equal(ok, want_ok)
    assert np.array_equal(val.view(np.uint64), want_val.view(np.uint64))
    assert 20 < ok.sum() < len(polys) - 20


@pytest.mark.gpu
def test_polygon_calls_need_the_mask(capi, oracle):
    rows, cols = 40, 40
    with capi.Context(0) as ctx:
        ctx.set_params(capi.default_params())
        ctx.set_geometry(rows, cols, 1, 0.05, (0.0, 0.0))
        ctx.upload_elevation(terrain(rows, cols, seed=1))
        ctx.run_chain(0)
        with pytest.raises(capi.TeError, match="footprint pass first"):
            ctx.run_polygon_footprint(FOOTPRINT, 0.0)
        with pytest.raises(capi.TeError, match="footprint pass first"):
            ctx.polygons_traversable([FOOTPRINT])
        with pytest.raises(capi.TeError):
            ctx.download("traversability_x")  # the layers do not exist before the first polygon pass
        # changing only the footprint parameters keeps the filter layers: the footprint pass alone is enough
        ctx.set_params(capi.default_params(fp_radius=0.2))
        ctx.run_footprint()
        ctx.run_polygon_footprint(FOOTPRINT, 0.0)
        ctx.sync()
        assert np.isfinite(ctx.download("traversability_x")).all()
