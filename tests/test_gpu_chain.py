"""Parity of the HIP chain (through the C-ABI) against the CPU oracle and the bag golden vector."""
import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, compare_layer, to_te_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no MI355X visible"
    return capi


def run_gpu(capi, elev, rows, cols, res, p, pos=(0.0, 0.0), flags=0, layers=OUT_LAYERS, batch=1):
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, batch, res, pos)
        ctx.upload_elevation(elev)
        ctx.run_chain(flags)
        ctx.sync()
        return {k: ctx.download(k) for k in layers}


def both(capi, oracle, elev, rows, cols, res, pos=(0.0, 0.0), **over):
    op = oracle.default_params(**over)
    g = oracle.geom(rows, cols, res, pos)
    want = oracle.chain(g, op, elev, want_normals=True)
    got = run_gpu(capi, elev, rows, cols, res, to_te_params(capi, op), pos, flags=capi.RUN_KEEP_NORMALS,
                  layers=OUT_LAYERS + ("surface_normal_x", "surface_normal_y", "surface_normal_z"))
    return got, want


def test_bag_golden_vector(capi, bag):
    """The reference's own golden outputs (default YAML) straight against the GPU."""
    rows, cols = int(bag["rows"]), int(bag["cols"])
    got = run_gpu(capi, bag["elevation"], rows, cols, float(bag["resolution"]), capi.default_params(),
                  tuple(bag["position"]))
    known = np.zeros(rows * cols, bool)
    for (i, j) in ((99, 117), (99, 118)):  # exactly planar border patches, golden used UnitZ
        known[j * rows + i] = True
    for k in OUT_LAYERS:
        n_bad, mx, _ = compare_layer(k, got[k][~known], bag[k][~known])
        assert n_bad == 0, (k, n_bad, mx)
    # the step filter is pure compare/select arithmetic: bit-exact
    assert (got["traversability_step"].view(np.uint32) == bag["traversability_step"].view(np.uint32)).all()


def test_bag_golden_vector_every_cell_with_the_2018_plane_rule(capi, oracle, bag):
    """TE_OPT_NORMALS_RANK_RULE: NormalVectorsFilter as the filter that wrote the bag had it (a disc whose scatter matrix is
    rank-deficient gets UnitZ: oracle/te_oracle.c, teo_set_normals_rank_rule).  With it the GPU reproduces the reference's
    golden layers on ALL 13 300 cells -- the two exactly planar, tilted border discs included -- and, being the generic
    kernels, bit for bit; without it those two cells are the only difference."""
    rows, cols = int(bag["rows"]), int(bag["cols"])
    with capi.Context(0) as ctx:
        ctx.set_params(capi.default_params())
        ctx.set_geometry(rows, cols, 1, float(bag["resolution"]), tuple(bag["position"]))
        ctx.upload_elevation(bag["elevation"])
        ctx.set_option(capi.OPT_NORMALS_RANK_RULE, 1)
        ctx.run_chain(0)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
        ctx.run_filter("normals")  # the single plugin's entry point follows the option too
        ctx.sync()
        assert np.array_equal(ctx.download("traversability_slope").view(np.uint32), bag["traversability_slope"].reshape(-1).view(np.uint32))
        ctx.set_option(capi.OPT_NORMALS_RANK_RULE, 0)
        ctx.run_chain(capi.RUN_GENERIC_KERNELS)
        ctx.sync()
        plain = {k: ctx.download(k) for k in OUT_LAYERS}
    for k in OUT_LAYERS:
        bad = np.flatnonzero(got[k].view(np.uint32) != bag[k].reshape(-1).view(np.uint32))
        assert bad.size == 0, (k, [(int(c % rows), int(c // rows)) for c in bad[:5]])
    diff = np.flatnonzero(plain["traversability_slope"].view(np.uint32) != got["traversability_slope"].view(np.uint32))
    assert {(int(c % rows), int(c // rows)) for c in diff} == {(99, 117), (99, 118)}
    # the oracle's rule and the library's agree on a map of exact planes, steps and noise
    from traversability_estimation_amd import synth
    r2, c2, res = 120, 90, 0.0625  # (a dyadic resolution: the oracle's centred sums of an exact plane cancel exactly, as on the bag)
    e = synth.perlin_elevation(r2, c2, seed=31).reshape(c2, r2)
    ii, jj = np.meshgrid(np.arange(r2, dtype=np.float32), np.arange(c2, dtype=np.float32))
    e[:, :40] = (np.float32(0.25) * ii + np.float32(0.125) * jj)[:, :40] * np.float32(1.0 / 32)  # an exact plane (dyadic slopes)
    e[20:40, 50:80] = np.float32(0.75)                                                            # a flat plateau with a step around it
    op = oracle.default_params(normals_radius=synth.benchmark_radius(2, res), rough_radius=synth.benchmark_radius(2, res))
    g = oracle.geom(r2, c2, res)
    oracle.set_normals_rank_rule(True)
    try:
        want = oracle.chain(g, op, e)
    finally:
        oracle.set_normals_rank_rule(False)
    ref = oracle.chain(g, op, e)
    assert (want["traversability_slope"] != ref["traversability_slope"]).sum() > 1000  # the tilted plane's cells
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(r2, c2, 1, res)
        ctx.upload_elevation(e)
        ctx.set_option(capi.OPT_NORMALS_RANK_RULE, 1)
        ctx.run_chain(0)
        ctx.sync()
        got2 = {k: ctx.download(k) for k in OUT_LAYERS}
    assert_layers_match(got2, want, ctx="exact planes with the 2018 plane rule")


def test_bag_vs_oracle(capi, oracle, bag):
    got, want = both(capi, oracle, bag["elevation"], int(bag["rows"]), int(bag["cols"]), float(bag["resolution"]),
                     tuple(bag["position"]))
    assert_layers_match(got, want, ctx="bag, default YAML")


@pytest.mark.parametrize("cells", [1.4, 3, 5, 9])
def test_perlin_radius_sweep(capi, oracle, cells):
    from traversability_estimation_amd import synth
    rows, cols, res = 200, 160, 0.05
    elev = synth.perlin_elevation(rows, cols, seed=100 + int(cells * 10))
    r = synth.benchmark_radius(cells, res)
    got, want = both(capi, oracle, elev, rows, cols, res, normals_radius=r, rough_radius=r, step_radius1=r,
                     step_radius2=r)
    assert_layers_match(got, want, ctx=f"perlin r={cells} cells")


def test_different_radii_per_filter(capi, oracle):
    from traversability_estimation_amd import synth
    rows, cols, res = 150, 170, 0.04
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=7), 12, seed=8)
    got, want = both(capi, oracle, elev, rows, cols, res, pos=(12.5, -3.25), normals_radius=0.1, rough_radius=0.17,
                     step_radius1=0.09, step_radius2=0.13, slope_critical=0.7, step_critical=0.2, step_ncrit=3,
                     rough_critical=0.08)
    assert_layers_match(got, want, ctx="different radii")


def test_holes_and_steps(capi, oracle):
    from traversability_estimation_amd import synth
    rows, cols, res = 180, 140, 0.05
    elev = synth.with_holes(synth.with_steps(synth.perlin_elevation(rows, cols, seed=21), 10, seed=22), 0.05, seed=23)
    elev[10:40, 20:60] = np.nan  # a big unobserved region
    elev[100, 100] = np.inf      # non-finite == invalid
    r = synth.benchmark_radius(4, res)
    got, want = both(capi, oracle, elev, rows, cols, res, normals_radius=r, rough_radius=r, step_radius1=r,
                     step_radius2=r)
    assert_layers_match(got, want, ctx="holes+steps")
    assert np.isnan(got["traversability_slope"]).sum() == np.isnan(elev).sum() + 1


def test_degenerate_inputs(capi, oracle):
    """Flat map, tilted exact plane, single valid cell, all-NaN map, 1-row map, radius below res/2."""
    res = 0.05
    cases = {}
    cases["flat"] = np.full((40, 70), 0.25, np.float32)
    jj, ii = np.meshgrid(np.arange(40), np.arange(70), indexing="ij")
    cases["plane"] = (0.125 * ii + 0.0625 * jj).astype(np.float32) * np.float32(res)
    lone = np.full((40, 70), np.nan, np.float32)
    lone[20, 30] = 1.0
    lone[5, 5:8] = 0.5   # three collinear cells
    lone[30:32, 50:52] = [[0.1, 0.2], [0.3, 0.7]]
    cases["sparse"] = lone
    cases["allnan"] = np.full((40, 70), np.nan, np.float32)
    for name, elev in cases.items():
        for cells in (0.3, 1.0 + 1e-6, 2.5):
            r = cells * res
            got, want = both(capi, oracle, elev, 70, 40, res, normals_radius=r, rough_radius=r, step_radius1=r,
                             step_radius2=r)
            assert_layers_match(got, want, ctx=f"{name} r={cells}")
    strip = np.linspace(0, 1, 300, dtype=np.float32).reshape(1, 300) ** 2
    got, want = both(capi, oracle, strip, 300, 1, res)
    assert_layers_match(got, want, ctx="1-column map")
    got, want = both(capi, oracle, strip.reshape(300, 1), 1, 300, res)
    assert_layers_match(got, want, ctx="1-row map")


def test_tie_radii_follow_the_reference_rounding(capi, oracle):
    """Radii that are exact multiples of the resolution: membership of the cells ON the
    circle depends on the double rounding of the positions and must be decided per cell like the reference."""
    from traversability_estimation_amd import synth
    rows, cols, res = 199, 131, 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=31), 8, seed=32)
    for r in (0.05, 0.25, 0.1):
        got, want = both(capi, oracle, elev, rows, cols, res, pos=(0.37, -1.21), normals_radius=r, rough_radius=r,
                         step_radius1=r, step_radius2=r)
        assert_layers_match(got, want, ctx=f"tie radius {r}")


def test_batch_of_maps(capi, oracle):
    from traversability_estimation_amd import synth
    rows, cols, res, B = 96, 80, 0.05, 5
    elevs = np.stack([synth.perlin_elevation(rows, cols, seed=2000 + b) for b in range(B)])
    r = synth.benchmark_radius(5, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
    got = run_gpu(capi, elevs, rows, cols, res, to_te_params(capi, op), batch=B)
    g = oracle.geom(rows, cols, res)
    n = rows * cols
    for b in range(B):
        want = oracle.chain(g, op, elevs[b])
        assert_layers_match({k: got[k][b * n:(b + 1) * n] for k in OUT_LAYERS}, want, ctx=f"map {b}")


def test_dirty_region_refilter(capi, oracle):
    from traversability_estimation_amd import synth
    rows, cols, res = 256, 192, 0.05
    elev = synth.perlin_elevation(rows, cols, seed=77)
    r = synth.benchmark_radius(5, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
    new = elev.copy()
    row0, col0, h, w = 100, 60, 48, 40
    new[col0:col0 + w, row0:row0 + h] += synth.perlin_elevation(h, w, seed=78) * 0.5
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain()
        ctx.upload_tile(new[col0:col0 + w, row0:row0 + h], 0, row0, col0)
        ctx.run_chain_region(0, row0, col0, h, w)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
    want = oracle.chain(oracle.geom(rows, cols, res), op, new)
    assert_layers_match(got, want, ctx="dirty tile")


def test_error_paths(capi):
    with capi.Context(0) as ctx:
        with pytest.raises(capi.TeError) as e:
            ctx.run_chain()
        assert e.value.code == capi.TE_ERR_NOT_READY
        ctx.set_geometry(64, 64, 1, 0.05)
        with pytest.raises(capi.TeError) as e:
            ctx.run_chain()
        assert e.value.code == capi.TE_ERR_NOT_READY  # no elevation
        with pytest.raises(capi.TeError) as e:
            ctx.set_params(capi.default_params(normals_radius=5.0))  # 100 cells
        assert e.value.code == capi.TE_ERR_UNSUPPORTED
        with pytest.raises(capi.TeError) as e:
            ctx.set_params(capi.default_params(slope_critical=3.0))
        assert e.value.code == capi.TE_ERR_BAD_PARAM


# ------------------------------------------------------------------------------------------------
# circular footprint pass (TraversabilityMap.cpp:307-318, 654-746, 774-921)
# ------------------------------------------------------------------------------------------------
FP_LAYERS = ("traversability_footprint", "slope_footprint", "step_footprint", "roughness_footprint")


def both_fp(capi, oracle, elev, rows, cols, res, pos=(0.0, 0.0), **over):
    op = oracle.default_params(**over)
    g = oracle.geom(rows, cols, res, pos)
    want = oracle.chain(g, op, elev)
    fp, memo = oracle.footprint(g, op, elev, want, want_memo=True)
    want["traversability_footprint"] = fp
    want.update(memo)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, pos)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT | capi.RUN_FOOTPRINT_MEMO)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS + FP_LAYERS}
    return got, want, op


def check_fp(got, want, op, ctx):
    layers = list(OUT_LAYERS) + ["traversability_footprint", "slope_footprint", "step_footprint"]
    if op.fp_check_roughness:
        layers.append("roughness_footprint")
    assert_layers_match(got, want, layers=layers, ctx=ctx)


def test_footprint_bag_default_yaml(capi, oracle, bag):
    got, want, op = both_fp(capi, oracle, bag["elevation"], int(bag["rows"]), int(bag["cols"]),
                            float(bag["resolution"]), tuple(bag["position"]))
    check_fp(got, want, op, "bag footprint")
    fp = got["traversability_footprint"]
    assert not np.isnan(fp).any() and (fp == 0).sum() > 100 and (fp > 0.5).sum() > 1000


@pytest.mark.parametrize("seed,rough", [(41, 0), (42, 1)])
def test_footprint_perlin_with_obstacles(capi, oracle, seed, rough):
    from traversability_estimation_amd import synth
    rows, cols, res = 220, 180, 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=seed, amplitude=0.15), 14, seed=seed + 100)
    elev[60:70, 30:50] = np.nan
    r = synth.benchmark_radius(3, res)
    got, want, op = both_fp(capi, oracle, elev, rows, cols, res, pos=(3.0, -7.0), normals_radius=r, rough_radius=r,
                            step_radius1=r, step_radius2=r, fp_radius=synth.benchmark_radius(6, res),
                            fp_offset=synth.benchmark_radius(3, res), fp_check_roughness=rough)
    check_fp(got, want, op, f"perlin+boxes seed {seed}")
    fp = got["traversability_footprint"]
    assert (fp == 0).sum() > 50 and ((fp > 0) & (fp < 1)).sum() > 1000


def test_footprint_tie_radius_and_zero_rmin(capi, oracle):
    """Default footprint radii (0.30 + 0.15 m) on a 0.05 m map are exact multiples of the resolution."""
    from traversability_estimation_amd import synth
    rows, cols, res = 160, 150, 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=51, amplitude=0.1), 10, seed=52)
    r = synth.benchmark_radius(2, res)
    for fr, fo in ((0.30, 0.15), (0.0, 0.25), (0.2, 0.0)):
        got, want, op = both_fp(capi, oracle, elev, rows, cols, res, normals_radius=r, rough_radius=r,
                                step_radius1=r, step_radius2=r, fp_radius=fr, fp_offset=fo)
        check_fp(got, want, op, f"tie footprint {fr}+{fo}")


def test_single_plugin_entry_points(capi, oracle):
    """te_run_filter: each reference plugin on exactly the layers it reads from mapIn (normals are INPUT)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 170, 130, 0.04
    elev = synth.with_holes(synth.with_steps(synth.perlin_elevation(rows, cols, seed=61), 8, seed=62), 0.02, seed=63)
    op = oracle.default_params(normals_radius=0.09, rough_radius=0.15, step_radius1=0.1, step_radius2=0.07)
    want = oracle.chain(oracle.geom(rows, cols, res), op, elev, want_normals=True)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ctx.upload_elevation(elev)
        for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z"):
            ctx.upload_layer(k, want[k])
        for f in ("slope", "step", "roughness", "combine"):
            ctx.run_filter(f)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
    assert_layers_match(got, want, ctx="single plugins")
    # with the reference's float32 normals as input the slope is the same double acos of the same float
    assert (got["traversability_step"].view(np.uint32) == want["traversability_step"].view(np.uint32)).all()


@pytest.mark.parametrize("start", [(0, 0), (5, 0), (0, 7), (37, 101), (89, 129)])
def test_circular_buffer_layers(capi, start):
    """A GridMap that has been move()d stores logical cell (i, j) at ((i+si) % rows, (j+sj) % cols)
    (grid_map_core getBufferIndexFromIndex); the reference's iterators hide that.  Upload in buffer order,
    run, download in buffer order == roll of the plain run, bit for bit."""
    from traversability_estimation_amd import synth
    rows, cols, res = 90, 130, 0.04
    elev = np.array(synth.perlin_elevation(rows, cols, seed=11), dtype=np.float32).reshape(cols, rows)
    elev[20:23, 40] = np.nan
    si, sj = start
    buf = np.roll(elev, (sj, si), axis=(0, 1))  # buf[(j+sj)%cols, (i+si)%rows] = elev[j, i]
    p = capi.default_params()
    plain = run_gpu(capi, elev, rows, cols, res, p)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, 1, res, (0.0, 0.0))
        ctx.upload_layer_circular("elevation", buf, start)
        ctx.run_chain(0)
        ctx.sync()
        assert np.array_equal(ctx.download("elevation").view(np.uint32), elev.reshape(-1).view(np.uint32))
        for k in OUT_LAYERS:
            got = ctx.download_layer_circular(k, start)
            want = np.roll(plain[k].reshape(cols, rows), (sj, si), axis=(0, 1))
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, start)
        with pytest.raises(capi.TeError):
            ctx.upload_layer_circular("elevation", buf, (rows, 0))
        with pytest.raises(capi.TeError):
            ctx.download_layer_circular("traversability", (0, -1))


def test_pinned_host_buffers(capi):
    """te_pin_host: transfers from / into page-locked buffers give the same bytes as pageable ones."""
    from traversability_estimation_amd import synth
    rows, cols = 96, 80
    elev = np.ascontiguousarray(synth.perlin_elevation(rows, cols, seed=3), dtype=np.float32).reshape(-1)
    plain = run_gpu(capi, elev, rows, cols, 0.05, capi.default_params())
    src, dst = elev.copy(), np.empty(rows * cols, np.float32)
    capi.pin_host(src)
    capi.pin_host(dst)
    try:
        with capi.Context(0) as ctx:
            ctx.set_params(capi.default_params())
            ctx.set_geometry(rows, cols, 1, 0.05)
            ctx.upload_elevation(src)
            ctx.run_chain(0)
            for k in OUT_LAYERS:
                ctx.download_into(k, dst)
                assert np.array_equal(dst.view(np.uint32), plain[k].view(np.uint32)), k
    finally:
        capi.unpin_host(src)
        capi.unpin_host(dst)
    with pytest.raises(capi.TeError):
        capi.unpin_host(dst)  # not registered any more


def _roughness_given_numpy(elev, nx, ny, nz, rows, cols, res, radius, crit, cells):
    """RoughnessFilter::update (RoughnessFilter.cpp:84-126) for the listed cells, normals taken from the layers.
    Layout: o = j * rows + i; grid_map positions x = -res * i, y = -res * j (any common shift cancels)."""
    R = int(np.floor(radius / res))
    out = {}
    for (i, j) in cells:
        o = j * rows + i
        if not np.isfinite(nx[o]):
            out[(i, j)] = np.nan
            continue
        pts = []
        for dj in range(-R, R + 1):
            for di in range(-R, R + 1):
                a, b = i + di, j + dj
                if a < 0 or a >= rows or b < 0 or b >= cols:
                    continue
                if (di * di + dj * dj) * res * res > radius * radius:
                    continue
                z = elev[b * rows + a]
                if np.isfinite(z):
                    pts.append((-res * a, -res * b, float(z)))
        P = np.array(pts, dtype=np.float64).reshape(-1, 3)
        n = np.array([nx[o], ny[o], nz[o]], dtype=np.float64)
        if len(P) == 0:
            out[(i, j)] = 1.0 if crit > 0 else 0.0
            continue
        mean = P.sum(axis=0) / len(P)
        d = P @ n - mean @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            rough = np.sqrt(np.float64((d * d).sum()) / np.float64(len(P) - 1)) if len(P) > 1 else np.inf
        out[(i, j)] = 1.0 - rough / crit if rough < crit else 0.0
    return out


@pytest.mark.parametrize("radius_cells", [3.6, 5.4, 9.2])
def test_roughness_plugin_with_given_normals(capi, radius_cells):
    """te_run_filter(roughness) with normals that are NOT the chain's own (the unchanged-YAML case: they come from the
    upstream host filter): the sliding kernel's closed form n^T C n + fix-up pass against the generic kernel on every
    cell and against a numpy restatement of RoughnessFilter.cpp:84-126 on a sample -- the frame, cells beside holes,
    invalid centres under a valid normal, cells without a normal."""
    from traversability_estimation_amd import synth
    rows, cols, res = 330, 270, 0.05
    rng = np.random.default_rng(17)
    elev = synth.with_holes(synth.with_steps(synth.perlin_elevation(rows, cols, seed=71), 6, seed=72), 0.01, seed=73)
    elev = np.ascontiguousarray(elev, dtype=np.float32).reshape(-1)
    n = rows * cols
    v = rng.normal(size=(n, 3)) * np.array([0.3, 0.3, 0.0]) + np.array([0.0, 0.0, 1.0])
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    nx, ny, nz = (np.ascontiguousarray(v[:, k], dtype=np.float32) for k in range(3))
    gone = rng.random(n) < 0.03
    nx[gone] = np.nan
    radius = radius_cells * res
    p = capi.default_params(rough_radius=radius)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, 1, res)
        ctx.upload_elevation(elev)
        for k, a in (("surface_normal_x", nx), ("surface_normal_y", ny), ("surface_normal_z", nz)):
            ctx.upload_layer(k, a)
        ctx.run_filter("roughness")
        ctx.sync()
        fast = ctx.download("traversability_roughness")
        ctx.run_filter("roughness", capi.RUN_GENERIC_KERNELS)
        ctx.sync()
        generic = ctx.download("traversability_roughness")
        # the input layers are untouched
        assert np.array_equal(ctx.download("surface_normal_x").view(np.uint32), nx.view(np.uint32))
        assert np.array_equal(ctx.download("surface_normal_z").view(np.uint32), nz.view(np.uint32))
    assert np.array_equal(np.isnan(fast), np.isnan(generic))
    assert np.array_equal(np.isnan(fast), gone)
    ok = ~gone
    assert np.max(np.abs(fast[ok] - generic[ok])) <= 1e-5
    # a sample against the reference's own arithmetic
    invalid = np.flatnonzero(~np.isfinite(elev) & ok)
    cells = [(0, 0), (rows - 1, cols - 1), (0, cols // 2), (rows // 2, 0), (3, 5), (rows - 2, 7)]
    cells += [(int(o % rows), int(o // rows)) for o in invalid[:40]]
    cells += [(int(o % rows), int(o // rows)) for o in rng.integers(0, n, size=150)]
    crit = float(p.rough_critical)
    want = _roughness_given_numpy(elev, nx, ny, nz, rows, cols, res, radius, crit, cells)
    for (i, j), w in want.items():
        g = fast[j * rows + i]
        assert (np.isnan(w) and np.isnan(g)) or abs(float(g) - w) <= 1e-5, ((i, j), float(g), w)
