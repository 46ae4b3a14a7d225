"""Round-3 kernels and entry points against the oracle, through the C-ABI:
the fixed-point footprint kernel (k_fp_slide4) over footprint radii and obstacle densities, the double kernel it falls
back to for a traversability layer that was not written by the chain, the step filter's nCells shortcut, and
te_run_chain_region refusing a footprint flag it does not implement."""
import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, compare_layer, to_te_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def obstacle_map(synth, rows, cols, seed, boxes, amplitude=0.12):
    return synth.with_steps(synth.perlin_elevation(rows, cols, seed=seed, amplitude=amplitude), boxes, seed=seed + 7)


@pytest.mark.parametrize("fp_cells,off_cells,boxes", [(2, 1, 6), (4, 2, 10), (6, 3, 14), (6, 3, 0), (7.3, 2.4, 12), (9, 4, 10), (11, 4, 8)])
def test_fixed_point_footprint_over_radii(capi, oracle, fp_cells, off_cells, boxes):
    """k_fp_slide4: discs of 29 .. 709 cells (the fixed-point scale drops from 2^21 to 2^18 with the radius), clean maps
    and maps with boxes (spiral walks, inner-radius hits), borders on all sides (the map is 1.2 x the widest disc)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 230, 190, 0.05
    elev = obstacle_map(synth, rows, cols, 300 + int(10 * fp_cells), boxes)
    elev[100:104, 60:75] = np.nan
    r = synth.benchmark_radius(3, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(fp_cells, res), fp_offset=synth.benchmark_radius(off_cells, res))
    g = oracle.geom(rows, cols, res, (1.5, -2.0))
    want = oracle.chain(g, op, elev)
    want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, (1.5, -2.0))
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS + ("traversability_footprint",)}
    assert_layers_match(got, want, layers=list(OUT_LAYERS) + ["traversability_footprint"], ctx=f"footprint {fp_cells}+{off_cells} cells, {boxes} boxes")
    fp = got["traversability_footprint"]
    assert not np.isnan(fp).any()
    if boxes:
        assert (fp == 0).sum() > 20 and ((fp > 0) & (fp < 1)).sum() > 1000


def test_footprint_of_an_uploaded_traversability_layer(capi, oracle):
    """A traversability layer that does not come from the chain is not bounded by the weights: values above 1 (and
    NaN) must go through the double kernel, same result as the oracle's isTraversable on that layer."""
    from traversability_estimation_amd import synth
    rows, cols, res = 200, 170, 0.05
    elev = obstacle_map(synth, rows, cols, 77, 8)
    r = synth.benchmark_radius(3, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(5, res), fp_offset=synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res)
    layers = oracle.chain(g, op, elev)
    rng = np.random.default_rng(5)
    # (values up to ~4: the double kernel keeps the untraversable count apart from sums below 2048, i.e. 8 per cell here)
    t = (layers["traversability"].astype(np.float64) * 3.0 + rng.uniform(0.0, 1.0, size=rows * cols)).astype(np.float32)
    t[rng.random(rows * cols) < 0.01] = np.nan
    layers["traversability"] = t
    want = oracle.footprint(g, op, elev, layers)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(0)
        ctx.upload_layer("traversability", t)
        ctx.run_footprint()
        ctx.sync()
        got = ctx.download("traversability_footprint")
        n_bad, mx, _ = compare_layer("traversability_footprint", got, want, tol=4e-5)  # values reach 4
        assert n_bad == 0, (n_bad, mx)
        assert np.nanmax(got) > 1.5
        # the chain writes the layer again: back to the bounded (fixed-point) path, same answer as a fresh context
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        again = ctx.download("traversability_footprint")
    layers2 = oracle.chain(g, op, elev)
    want2 = oracle.footprint(g, op, elev, layers2)
    n_bad, mx, _ = compare_layer("traversability_footprint", again, want2)
    assert n_bad == 0, (n_bad, mx)


@pytest.mark.parametrize("ncrit,crit", [(1, 0.12), (4, 0.12), (9, 0.05), (40, 0.02), (4, 0.0)])
def test_step_filter_cell_count_cases(capi, oracle, ncrit, crit):
    """StepFilter.cpp:165-176: nCells == 0, 0 < nCells < nCellCritical, nCells >= nCellCritical -- the kernel computes
    the score arithmetic only in the middle case; every case must stay bit-identical."""
    from traversability_estimation_amd import synth
    rows, cols, res = 192, 160, 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=9, amplitude=0.04), 5, seed=10)
    elev[30:33, 100:140] = np.nan
    for cells in (2, 5):
        r = synth.benchmark_radius(cells, res)
        op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, step_critical=crit, step_ncrit=ncrit)
        want = oracle.chain(oracle.geom(rows, cols, res), op, elev)
        with capi.Context(0) as ctx:
            ctx.set_params(to_te_params(capi, op))
            ctx.set_geometry(rows, cols, 1, res)
            ctx.upload_elevation(elev)
            ctx.run_chain(0)
            ctx.sync()
            got = ctx.download("traversability_step")
        w = want["traversability_step"]
        assert np.array_equal(np.isnan(got), np.isnan(w))
        assert np.array_equal(got[~np.isnan(got)].view(np.uint32), w[~np.isnan(w)].view(np.uint32)), (ncrit, crit, cells)
        if crit > 0 and ncrit > 1:
            assert ((w > 0) & (w < 1)).sum() > 0  # the middle case occurs


@pytest.mark.parametrize("ties", [False, True])
def test_region_run_with_the_footprint_flag(capi, oracle, ties):
    """te_run_chain_region(TE_RUN_FOOTPRINT): after a sequence of dirty rectangles (interior, touching borders, overlapping)
    every layer incl. traversability_footprint and the memo layers equals the oracle's whole-map result.  ties: every
    radius a whole number of cells (the TIES march, the step folds and the fixed-point footprint's tie variant on regions)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 300, 260, 0.05
    elev = obstacle_map(synth, rows, cols, 21, 10).reshape(cols, rows)
    r = 4 * res if ties else synth.benchmark_radius(4, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=5 * res if ties else synth.benchmark_radius(5, res),
                               fp_offset=2 * res if ties else synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res)
    layers = list(OUT_LAYERS) + ["traversability_footprint", "slope_footprint", "step_footprint"]
    rng = np.random.default_rng(3)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(0)
        with pytest.raises(capi.TeError):  # no complete footprint layer yet
            ctx.run_chain_region(0, 10, 10, 32, 32, flags=capi.RUN_FOOTPRINT)
        ctx.run_chain(capi.RUN_FOOTPRINT | capi.RUN_FOOTPRINT_MEMO)
        for (r0, c0, h, w) in ((100, 90, 40, 30), (0, 0, 25, 60), (rows - 20, cols - 35, 20, 35), (120, 100, 64, 64), (250, 5, 50, 17)):
            tile = elev[c0:c0 + w, r0:r0 + h] + rng.normal(0.0, 0.03, size=(w, h)).astype(np.float32)
            if h > 30:
                tile[3:6, 4:9] += np.float32(0.5)  # a box: untraversable cells, spiral walks
            elev[c0:c0 + w, r0:r0 + h] = tile
            ctx.upload_tile(np.ascontiguousarray(tile), 0, r0, c0)
            ctx.run_chain_region(0, r0, c0, h, w, flags=capi.RUN_FOOTPRINT | capi.RUN_FOOTPRINT_MEMO)
        ctx.sync()
        got = {k: ctx.download(k) for k in layers}
        # a tile of a layer comes back as uploaded tiles go in
        t = ctx.download_tile("traversability_footprint", 0, 117, 95, 33, 21)
        assert np.array_equal(t.view(np.uint32), got["traversability_footprint"].reshape(cols, rows)[95:116, 117:150].view(np.uint32))
    want = oracle.chain(g, op, elev)
    fp, memo = oracle.footprint(g, op, elev, want, want_memo=True)
    want["traversability_footprint"] = fp
    want.update(memo)
    assert_layers_match(got, want, layers=layers, ctx="region runs with the footprint flag")


def test_streaming_tiles_async_against_the_oracle(capi, oracle):
    """BASELINE configs[4] shape at a size the oracle finishes: 20 ticks of te_upload_tile_async + te_run_chain_region
    (+ footprint) + te_download_tile_async on page-locked buffers, two ticks in flight; every downloaded tile equals the
    oracle's result for the elevation of ITS tick, and the final layers equal the oracle's for the final map."""
    from traversability_estimation_amd import synth
    rows, cols, res, T = 512, 448, 0.05, 64
    elev = synth.perlin_elevation(rows, cols, seed=77).reshape(cols, rows).copy()
    r = synth.benchmark_radius(5, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(4, res), fp_offset=synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res)
    rng = np.random.default_rng(77)
    ticks = 20
    tiles_in = [np.empty((T, T), np.float32) for _ in range(ticks)]
    tiles_out = [np.empty((T, T), np.float32) for _ in range(ticks)]
    for b in tiles_in + tiles_out:
        capi.pin_host(b)
    origins, snapshots = [], []
    try:
        with capi.Context(0) as ctx:
            ctx.set_params(to_te_params(capi, op))
            ctx.set_geometry(rows, cols, 1, res)
            ctx.upload_elevation(elev)
            ctx.run_chain(capi.RUN_FOOTPRINT)
            for k in range(ticks):
                r0, c0 = int(rng.integers(0, rows - T + 1)), int(rng.integers(0, cols - T + 1))
                tiles_in[k][:] = synth.perlin_elevation(T, T, seed=1000 + k).reshape(T, T) + np.float32(0.2 * (k % 3))
                elev[c0:c0 + T, r0:r0 + T] = tiles_in[k]
                origins.append((r0, c0))
                snapshots.append(elev.copy())
                ctx.upload_tile_async(tiles_in[k], 0, r0, c0)
                ctx.run_chain_region(0, r0, c0, T, T, flags=capi.RUN_FOOTPRINT)
                ctx.download_tile_async("traversability_footprint", 0, r0, c0, tiles_out[k])
                if k % 2 == 1:
                    ctx.sync()  # two ticks in flight
            ctx.sync()
            final = {k: ctx.download(k) for k in OUT_LAYERS + ("traversability_footprint",)}
        oracle.set_threads(8)
        for k in (0, 7, 13, ticks - 1):
            want = oracle.chain(g, op, snapshots[k])
            fp = oracle.footprint(g, op, snapshots[k], want).reshape(cols, rows)
            r0, c0 = origins[k]
            n_bad, mx, _ = compare_layer("traversability_footprint", tiles_out[k], fp[c0:c0 + T, r0:r0 + T])
            assert n_bad == 0, (k, n_bad, mx)
        want = oracle.chain(g, op, elev)
        want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
        assert_layers_match(final, want, layers=list(OUT_LAYERS) + ["traversability_footprint"], ctx="after 20 streamed ticks")
    finally:
        oracle.set_threads(1)
        for b in tiles_in + tiles_out:
            capi.unpin_host(b)


@pytest.mark.parametrize("off_cells", [0.0, 2.0, -2.0])
def test_blocked_discs_of_a_batch(capi, oracle, off_cells):
    """The footprint pass's list of blocked cells (k_fp_slide4 appends, k_fp_blocked walks) indexes the cells of ALL
    maps of a batch; with radiusMin = 0 nothing is listed and a blocked disc is 0 outright (:694-704)."""
    from traversability_estimation_amd import synth
    rows, cols, res, B = 150, 130, 0.05, 3
    elevs = [obstacle_map(synth, rows, cols, 900 + 13 * b, 4 + 9 * b) for b in range(B)]
    r = synth.benchmark_radius(3, res)
    if off_cells < 0:  # a tie radius: 5 + 2 cells exactly
        fpr, fpo = 5 * res, -off_cells * res
    else:
        fpr, fpo = synth.benchmark_radius(5, res), synth.benchmark_radius(off_cells, res) if off_cells else 0.0
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=fpr, fp_offset=fpo)
    g = oracle.geom(rows, cols, res, (0.0, 0.0))
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, B, res)
        ctx.upload_elevation(np.stack(elevs))
        for _ in range(2):  # the second pass starts from an emptied list
            ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        got = ctx.download("traversability_footprint").reshape(B, cols, rows)
    for b in range(B):
        want = oracle.chain(g, op, elevs[b])
        fp = oracle.footprint(g, op, elevs[b], want)
        bad, worst, _ = compare_layer("traversability_footprint", got[b], fp)
        assert bad == 0, f"map {b}: {bad} cells differ (worst {worst})"
        assert (got[b] == 0).sum() > 20


@pytest.mark.parametrize("res,fp_cells,off_cells,origin", [(0.05, 4, 2, (1.5, -2.0)), (0.03, 10, 5, (0.0, 0.0)), (0.03, 12, 1, (-3.17, 8.4)),
                                                          (0.1, 5, 0, (0.35, 0.05)), (0.04, 7, 3, (12.3, -7.7))])
def test_fixed_point_footprint_at_tie_radii(capi, oracle, res, fp_cells, off_cells, origin):
    """radius + offset a whole number of cells (the reference's own 0.45 m at 0.03 m): the cells exactly on the circle
    belong to a disc or not as SpiralIterator::isInside decides from rounded positions, centre by centre -- the
    fixed-point kernel slides the shape with its circle and takes the rejected cells out per row (k_fp_slide4<Q, true>),
    k_fp_blocked applies the test to the flagged table entries.  Map origins that move the rounding around."""
    from traversability_estimation_amd import synth
    rows, cols = 200, 170
    elev = obstacle_map(synth, rows, cols, 500 + fp_cells, 10)
    elev[90:93, 40:60] = np.nan
    r = synth.benchmark_radius(3, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=fp_cells * res, fp_offset=off_cells * res)
    g = oracle.geom(rows, cols, res, origin)
    want = oracle.chain(g, op, elev)
    want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, origin)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS + ("traversability_footprint",)}
    assert_layers_match(got, want, layers=list(OUT_LAYERS) + ["traversability_footprint"], ctx=f"tie radius {fp_cells}+{off_cells} cells at {res} m")
    fp = got["traversability_footprint"]
    assert (fp == 0).sum() > 20 and ((fp > 0) & (fp < 1)).sum() > 1000


@pytest.mark.parametrize("res,cells,holes,origin", [(0.05, 3, False, (1.5, -2.0)), (0.03, 5, False, (0.0, 0.0)), (0.1, 5, True, (-3.17, 8.4)),
                                                    (0.04, 9, False, (0.35, 0.05)), (0.05, 10, True, (12.3, -7.7)), (0.02, 2, False, (0.0, 0.0))])
def test_chain_at_tie_radii(capi, oracle, res, cells, holes, origin):
    """normals / roughness / step radii a whole number of cells: CircleIterator::isInside decides the cells exactly on
    the circle from rounded positions, centre by centre.  k_normals3's TIES march slides the shape with its circle and
    takes the rejected cells out of the moments row by row (5 and 10 cells: the 3-4-5 offsets as well); a strip with
    invalid cells goes to the fix-up pass whole.  Borders on all sides, map origins that move the rounding around."""
    from traversability_estimation_amd import synth
    rows, cols = 210, 180
    elev = obstacle_map(synth, rows, cols, 700 + cells, 8)
    if holes:
        elev[90:93, 40:60] = np.nan
        elev[150, 100] = np.nan
    r = cells * res
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(4, res), fp_offset=synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res, origin)
    want = oracle.chain(g, op, elev)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, origin)
        ctx.upload_elevation(elev)
        ctx.run_chain(0)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
    assert_layers_match(got, want, layers=list(OUT_LAYERS), ctx=f"chain at {cells} cells exactly, res {res}")


@pytest.mark.parametrize("cells", [3, 5, 8])
def test_normals_kept_at_a_tie_radius(capi, oracle, cells):
    """TE_RUN_KEEP_NORMALS (the plugin path's normals filter) at a whole-cell radius: k_normals3<Q, true, 2, true>."""
    from traversability_estimation_amd import synth
    rows, cols, res = 190, 160, 0.05
    elev = obstacle_map(synth, rows, cols, 40 + cells, 6)
    r = cells * res
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=synth.benchmark_radius(3, res), step_radius2=synth.benchmark_radius(3, res))
    g = oracle.geom(rows, cols, res, (2.2, -1.1))
    want = oracle.chain(g, op, elev, want_normals=True)
    layers = list(OUT_LAYERS) + ["surface_normal_x", "surface_normal_y", "surface_normal_z"]
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, (2.2, -1.1))
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_KEEP_NORMALS)
        ctx.sync()
        got = {k: ctx.download(k) for k in layers}
    assert_layers_match(got, want, layers=layers, ctx=f"normals kept at {cells} cells exactly")


_DENSE_LIST_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
from traversability_estimation_amd import capi, synth
from tests.helpers import compare_layer, to_te_params
O.build()
capi.load()
rows, cols, res = 260, 230, 0.05
bad_total = 0
for fp_cells, off_cells, boxes, origin in ((6, 3, 30, (0.0, 0.0)), (9, 4, 18, (1.5, -2.0)), (4, 1, 40, (0.0, 0.0)), (-5, -2, 30, (0.7, 0.3))):
    tie = fp_cells < 0  # whole-cell radii: the cells on the circle are decided per centre
    fp_cells, off_cells = abs(fp_cells), abs(off_cells)
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=40 + boxes, amplitude=0.12), boxes, seed=47 + boxes)
    elev[100:104, 60:75] = np.nan
    r = synth.benchmark_radius(3, res)
    op = O.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                          fp_radius=fp_cells * res if tie else synth.benchmark_radius(fp_cells, res),
                          fp_offset=off_cells * res if tie else synth.benchmark_radius(off_cells, res))
    g = O.geom(rows, cols, res, origin)
    want = O.chain(g, op, elev)
    fp = O.footprint(g, op, elev, want)
    with capi.Context(0) as ctx:
        ctx.set_option(capi.OPT_FP_BLOCKED_WALK, 2)           # one disc per lane
        ctx.set_option(capi.OPT_FP_BLOCKED_BLOCKS_PER_CU, 1)  # 256 wavefronts: the lanes of a group are filled
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, origin)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        got = ctx.download("traversability_footprint")
    bad, worst, nn = compare_layer("traversability_footprint", got, fp)
    partial = int(((got > 0) & (got < 1)).sum())
    print(fp_cells, off_cells, boxes, "mismatches", bad, "worst", worst, "cells strictly between 0 and 1:", partial)
    bad_total += bad + (0 if partial > 1000 else 1)
sys.exit(1 if bad_total else 0)
"""


def test_blocked_discs_one_per_lane():
    """k_fp_blocked walks one disc per lane when the list is long for the launch (>= 8 cells per wavefront: 49 152
    on an MI355X, more than a map the oracle finishes in seconds can hold).  te_set_option(TE_OPT_FP_BLOCKED_WALK, 2)
    forces that walk and TE_OPT_FP_BLOCKED_BLOCKS_PER_CU = 1 shrinks the launch to 256 wavefronts, so that the lanes of
    a group are filled.  Maps with borders on all sides (the bounds-checked loops) and an interior large enough for the
    scalar-offset ones; the last case is a tie radius."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", _DENSE_LIST_SCRIPT, root], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
