"""Round 6: the shapes the reference's DEFAULT parameters take at common map resolutions, through the C-ABI against the oracle.

robot_filter_parameter.yaml has normals / roughness radius 0.05 m and step windows 0.04 m.  On a 0.05 m map the 0.05 m radius is a
TIE radius of exactly ONE cell -- the runs hold the centre alone, the four edge neighbours lie on the circle and are kept or
dropped centre by centre by the rounding of the positions (CircleIterator::isInside) -- and the step windows hold one cell; on a
0.04 m map the step windows are the one-cell tie and the normals disc is the 5-point one.  Rounds 1-5 served a one-cell tie
radius with the generic kernels (k_normals: 0.99 ms on 4096^2 against 0.09 ms for the neighbouring tie-free shape)."""
import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, orient_horizontal_normals, to_te_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def _map(synth, rows, cols, seed, boxes=8, holes=False):
    e = synth.with_steps(synth.perlin_elevation(rows, cols, seed=seed, amplitude=0.12), boxes, seed=seed + 7)
    if holes:
        e = synth.with_holes(e, 0.004, seed=seed + 1)
        e[cols // 2:cols // 2 + 5, 30:70] = np.nan
    return e


@pytest.mark.parametrize("res,origin,holes,keep", [(0.05, (0.0, 0.0), False, False), (0.05, (1.5, -2.0), True, False), (0.05, (12.3, -7.7), False, True),
                                                   (0.04, (0.0, 0.0), False, False), (0.04, (-3.17, 8.4), True, False), (0.1, (0.35, 0.05), True, True),
                                                   (0.025, (0.0, 0.0), False, False)])
def test_default_yaml_radii_on_maps_where_they_are_one_cell_ties(capi, oracle, res, origin, holes, keep):
    """Default parameters, map resolutions that make 0.05 m / 0.04 m exactly one cell (res 0.05: normals + roughness; res
    0.04: the step windows; res 0.025: two cells; res 0.1: neither) -- k_normals_small, k_step_small."""
    from traversability_estimation_amd import synth
    rows, cols = 330, 210
    elev = _map(synth, rows, cols, 900 + int(res * 1000), holes=holes)
    op = oracle.default_params()
    g = oracle.geom(rows, cols, res, origin)
    want = oracle.chain(g, op, elev, want_normals=keep)
    want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
    layers = list(OUT_LAYERS) + ["traversability_footprint"] + (["surface_normal_x", "surface_normal_y", "surface_normal_z"] if keep else [])
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, origin)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT | (capi.RUN_KEEP_NORMALS if keep else 0))
        ctx.sync()
        got = {k: ctx.download(k) for k in layers}
    if keep:  # (the sign of a horizontal normal is rounding noise in the reference itself: tests/helpers.py)
        got, want = orient_horizontal_normals(got, want["surface_normal_z"]), orient_horizontal_normals(want, want["surface_normal_z"])
    assert_layers_match(got, want, layers=layers, ctx=f"default YAML at res {res}, origin {origin}")


def test_one_cell_tie_radius_on_a_batch_and_a_region(capi, oracle):
    """The same one-cell tie shapes on a batch of maps (blockIdx.z) and through te_run_chain_region after a tile changed."""
    from traversability_estimation_amd import synth
    rows, cols, res, B = 256, 192, 0.05, 3
    elevs = np.stack([_map(synth, rows, cols, 40 + b, holes=(b == 1)) for b in range(B)])
    op = oracle.default_params(step_radius1=0.05, step_radius2=0.05)  # (all four radii one cell)
    g = oracle.geom(rows, cols, res, (0.7, 0.2))
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, B, res, (0.7, 0.2))
        ctx.upload_elevation(elevs)
        ctx.run_chain(0)
        ctx.sync()
        per = rows * cols
        for b in range(B):
            want = oracle.chain(g, op, elevs[b])
            got = {k: ctx.download(k)[b * per:(b + 1) * per] for k in OUT_LAYERS}
            assert_layers_match(got, want, layers=list(OUT_LAYERS), ctx=f"batch map {b}")
        # a dirty tile in map 2
        tile = (elevs[2][60:124, 100:164] + 0.07).astype(np.float32)
        elevs[2][60:124, 100:164] = tile
        ctx.upload_tile(np.ascontiguousarray(tile), 2, 100, 60)
        ctx.run_chain_region(2, 100, 60, 64, 64)
        ctx.sync()
        want = oracle.chain(g, op, elevs[2])
        got = {k: ctx.download(k)[2 * per:3 * per] for k in OUT_LAYERS}
        assert_layers_match(got, want, layers=list(OUT_LAYERS), ctx="region run after a dirty tile")


@pytest.mark.parametrize("r1,r2", [(0.0, 0.0), (0.2, 0.0), (0.0, 0.15), (0.05, 0.05)])
def test_step_windows_of_radius_zero(capi, oracle, r1, r2):
    """A window radius of exactly 0: the disc is its centre, and the centre lies ON the circle (a tie cell that isInside always
    accepts).  The random sweep of round 6 (seeds 20477 ...) caught k_step_small counting it twice."""
    from traversability_estimation_amd import synth
    rows, cols, res = 200, 150, 0.05
    elev = _map(synth, rows, cols, 77, holes=True)
    op = oracle.default_params(step_radius1=r1, step_radius2=r2, step_ncrit=3, step_critical=0.08)
    g = oracle.geom(rows, cols, res, (0.3, -0.4))
    want = oracle.chain(g, op, elev)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, (0.3, -0.4))
        ctx.upload_elevation(elev)
        ctx.run_chain(0)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
    assert_layers_match(got, want, layers=list(OUT_LAYERS), ctx=f"step windows {r1} / {r2} m")
    if r1 > 0.0:  # (a first window of one cell makes every step height 0; otherwise the case has cells whose count matters)
        assert (np.asarray(want["traversability_step"]) < 1.0).sum() > 50


def test_whole_1024_map_default_yaml_res_005_against_the_oracle(capi, oracle):
    """Every cell of a 1024^2 map at res 0.05 with the default parameters (one-cell tie normals, footprint radius 0.45 m = 9 cells)."""
    from traversability_estimation_amd import synth
    n, res = 1024, 0.05
    elev = synth.with_steps(synth.perlin_elevation(n, n, seed=1234), 40, seed=5)
    op = oracle.default_params()
    g = oracle.geom(n, n, res)
    oracle.set_threads(16)
    try:
        want = oracle.chain(g, op, elev)
        want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
    finally:
        oracle.set_threads(1)
    layers = list(OUT_LAYERS) + ["traversability_footprint"]
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(n, n, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        got = {k: ctx.download(k) for k in layers}
    assert_layers_match(got, want, layers=layers, ctx="1024^2, default YAML, res 0.05")


def test_a_filter_beside_a_prefetch_of_its_own_input_joins_the_prefetch(capi, oracle):
    """te_run_filter / te_download_layer may run BESIDE a prefetch -- unless they touch one of its layers: TE_FILTER_STEP right
    behind an elevation prefetch must see the NEW elevation, whole (round 5's advisor: it read a half-written layer)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 1024, 768, 0.05
    e1 = synth.perlin_elevation(rows, cols, seed=11)
    e2 = synth.with_steps(synth.perlin_elevation(rows, cols, seed=12), 30, seed=13)
    op = oracle.default_params(step_radius1=synth.benchmark_radius(3, res), step_radius2=synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res)
    oracle.set_threads(8)
    try:
        want = oracle.step(g, e2, op.step_critical, op.step_radius1, op.step_radius2, op.step_ncrit)
    finally:
        oracle.set_threads(1)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ctx.upload_elevation(e1)
        ctx.run_filter("step")
        ctx.sync()
        for _ in range(3):  # (a race does not lose every time)
            ctx.prefetch_layers({"elevation": e2})
            ctx.run_filter("step")                       # reads elevation: joins the prefetch first
            got = ctx.download("traversability_step")    # (a layer the prefetch does not write: runs beside nothing by now)
            ctx.wait_prefetch()
            assert_layers_match({"traversability_step": got}, {"traversability_step": want}, layers=["traversability_step"], ctx="step beside an elevation prefetch")
            ctx.prefetch_layers({"elevation": e1})
            e_back = ctx.download("elevation")           # a download of the prefetched layer itself joins it too
            ctx.wait_prefetch()
            assert np.array_equal(e_back.view(np.uint32), np.ascontiguousarray(e1, np.float32).reshape(-1).view(np.uint32))


def test_graph_replay_option_gives_the_same_layers(capi):
    """TE_OPT_GRAPH_REPLAY: always / never -- a choice between launch forms, bit-identical layers (small map: the default is direct)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 320, 256, 0.05
    elev = _map(synth, rows, cols, 5, holes=True)
    r = synth.benchmark_radius(4, res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=synth.benchmark_radius(4, res),
                            fp_offset=synth.benchmark_radius(2, res))
    outs = []
    for mode in (2, 1, 0):
        with capi.Context(0) as ctx:
            ctx.set_params(p)
            ctx.set_geometry(rows, cols, 1, res)
            ctx.set_option(capi.OPT_GRAPH_REPLAY, mode)
            ctx.upload_elevation(elev)
            for _ in range(3):  # (capture, then two replays)
                ctx.run_chain(capi.RUN_FOOTPRINT)
            ctx.sync()
            outs.append({k: ctx.download(k) for k in list(OUT_LAYERS) + ["traversability_footprint"]})
    for k in outs[0]:
        for o in outs[1:]:
            assert np.array_equal(outs[0][k].view(np.uint32), o[k].view(np.uint32)), k
    with capi.Context(0) as ctx:
        with pytest.raises(capi.TeError):
            ctx.set_option(capi.OPT_GRAPH_REPLAY, 3)


@pytest.mark.parametrize("step_crit,weights,keep,fp", [(0.12, (1 / 3, 1, 1, 1), False, False), (0.0, (0.5, 2.0, 0.25, 1.0), True, True), (0.3, (1.0, 0.2, 0.5, 0.3), False, True)])
def test_single_cell_step_windows_fused_into_the_normals_kernel(capi, oracle, step_crit, weights, keep, fp):
    """Both step windows below one cell and a normals disc k_normals_small takes: that kernel writes the step layer (1 / NaN; 0
    for a critical value of 0) and the weighted sum itself -- the whole chain is one kernel + its fix-up pass.  A batch, holes,
    +inf cells, weights that are not the default, the footprint pass behind it (then the mask kernel combines)."""
    from traversability_estimation_amd import synth
    rows, cols, res, B = 300, 190, 0.05, 2
    elevs = np.stack([_map(synth, rows, cols, 60 + b, holes=True) for b in range(B)])
    elevs[1][40:44, 100:140] = np.inf
    op = oracle.default_params(step_critical=step_crit, step_radius1=0.04, step_radius2=0.03, w_scale=np.float32(weights[0]), w_slope=weights[1],
                               w_step=weights[2], w_rough=weights[3], fp_critical_step=max(step_crit, 0.05))
    g = oracle.geom(rows, cols, res, (4.1, -0.6))
    layers = list(OUT_LAYERS) + (["traversability_footprint"] if fp else []) + (["surface_normal_x", "surface_normal_y", "surface_normal_z"] if keep else [])
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, B, res, (4.1, -0.6))
        ctx.upload_elevation(elevs)
        ctx.run_chain((capi.RUN_FOOTPRINT if fp else 0) | (capi.RUN_KEEP_NORMALS if keep else 0))
        ctx.sync()
        per = rows * cols
        for b in range(B):
            want = oracle.chain(g, op, elevs[b], want_normals=keep)
            if fp:
                want["traversability_footprint"] = oracle.footprint(g, op, elevs[b], want)
            got = {k: ctx.download(k)[b * per:(b + 1) * per] for k in layers}
            if keep:
                got, want = orient_horizontal_normals(got, want["surface_normal_z"]), orient_horizontal_normals(want, want["surface_normal_z"])
            assert_layers_match(got, want, layers=layers, ctx=f"fused single-cell steps, map {b}")
            st = np.asarray(want["traversability_step"])
            assert set(np.unique(st[~np.isnan(st)])) <= {np.float32(1.0 if step_crit > 0 else 0.0)}


@pytest.mark.parametrize("nc,s1,s2,keep,fp", [(1.67, 1.33, 1.33, False, True),     # the bag map's shapes: 9-point disc, 5-point windows
                                               (1.0, 1.42, 1.2, True, False),      # one-cell tie radius; 3 x 3 first window, 5-point second
                                               (2.0, 0.5, 1.42, False, False),     # two-cell tie radius; single-cell first window
                                               (2.24, 1.2, 0.4, False, True),      # 21-point disc; single-cell second window
                                               (2.9, 1.45, 1.45, True, False),     # the whole 5 x 5 window; 3 x 3 windows
                                               (1.5, 1.0, 1.3, False, False)])     # a TIE first window: not this kernel's (the stage-by-stage path)
def test_the_whole_chain_of_a_small_map_in_one_kernel(capi, oracle, nc, s1, s2, keep, fp):
    """k_chain_window: small launches whose normals disc reaches at most two cells and whose step windows fit 3 x 3 -- normals,
    slope, roughness, both step passes and the weighted sum from the 5 x 5 window of elevations one thread holds.  Borders on
    all sides, holes, +inf, steps, a batch; against the oracle (and the last row: a shape the kernel must leave alone)."""
    from traversability_estimation_amd import synth
    rows, cols, res, B = 150, 131, 0.04, 2
    elevs = np.stack([_map(synth, rows, cols, 300 + b, boxes=10, holes=True) for b in range(B)])
    elevs[0][10:12, 30:60] = np.inf
    op = oracle.default_params(normals_radius=nc * res, rough_radius=nc * res, step_radius1=s1 * res, step_radius2=s2 * res, step_ncrit=3, step_critical=0.05,
                               fp_radius=synth.benchmark_radius(3, res), fp_offset=synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res, (-1.3, 2.2))
    layers = list(OUT_LAYERS) + (["traversability_footprint"] if fp else []) + (["surface_normal_x", "surface_normal_y", "surface_normal_z"] if keep else [])
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, B, res, (-1.3, 2.2))
        ctx.upload_elevation(elevs)
        ctx.run_chain((capi.RUN_FOOTPRINT if fp else 0) | (capi.RUN_KEEP_NORMALS if keep else 0))
        ctx.sync()
        per = rows * cols
        for b in range(B):
            want = oracle.chain(g, op, elevs[b], want_normals=keep)
            if fp:
                want["traversability_footprint"] = oracle.footprint(g, op, elevs[b], want)
            got = {k: ctx.download(k)[b * per:(b + 1) * per] for k in layers}
            if keep:
                got, want = orient_horizontal_normals(got, want["surface_normal_z"]), orient_horizontal_normals(want, want["surface_normal_z"])
            assert_layers_match(got, want, layers=layers, ctx=f"one-kernel chain, normals {nc} / windows {s1}, {s2} cells, map {b}")
