"""Round 5: scores AT THEIR CLIP.

SlopeFilter / RoughnessFilter clip their scores at 0 (SlopeFilter.cpp:77-81, RoughnessFilter.cpp:119-124) and the
footprint checks memoise only cells whose score IS 0 (TraversabilityMap.cpp:869-871, :897): a score that is 0 on one
side of the comparison and 1e-7 on the other agrees within 1e-5 and still changes the NaN pattern of slope_footprint /
roughness_footprint.  The fast tails therefore leave every cell whose raw score lies within their own error of the clip
to the fix-up pass, which settles it with the generic (oracle-identical) arithmetic (round 6's k_normals_small and
k_chain_window run that arithmetic in place: no pass behind them).  These tests put cells exactly there:
the critical values are taken from the oracle's own normal / roughness of chosen cells, so that those cells sit ON the
clip (score exactly 0 in the oracle, neighbours in slope within 1e-7 of it), for every march of the normals kernels."""
import os

import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, to_te_params
from tests.test_gpu_chain import both_fp, check_fp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no MI355X visible"
    return capi


def _case(kind):
    """(rows, cols, res, elevation, radius, what the case exercises)"""
    from traversability_estimation_amd import synth
    res = 0.05
    if kind == "slim":  # hole-free, shape 81: k_normals3s
        rows, cols, cells = 200, 150, 9
        elev = synth.perlin_elevation(rows, cols, seed=501, amplitude=0.9)
        return rows, cols, res, elev, synth.benchmark_radius(cells, res)
    if kind == "clean":  # hole-free, shape 9: k_normals3's clean march
        rows, cols, cells = 200, 150, 3
        elev = synth.perlin_elevation(rows, cols, seed=502, amplitude=0.9)
        return rows, cols, res, elev, synth.benchmark_radius(cells, res)
    if kind == "sparse":  # 0.1 % speckle: the sparse-hole march and its queue
        rows, cols, cells = 260, 200, 5
        elev = synth.with_holes(synth.perlin_elevation(rows, cols, seed=503, amplitude=0.9), 0.001, seed=9)
        return rows, cols, res, elev, synth.benchmark_radius(cells, res)
    if kind == "dense":  # 3 % speckle: the dense-hole march
        rows, cols, cells = 200, 150, 4
        elev = synth.with_holes(synth.perlin_elevation(rows, cols, seed=504, amplitude=0.9), 0.03, seed=10)
        return rows, cols, res, elev, synth.benchmark_radius(cells, res)
    if kind == "ties":  # radius exactly 3 cells: the TIES march
        rows, cols = 200, 150
        elev = synth.perlin_elevation(rows, cols, seed=505, amplitude=0.9)
        return rows, cols, res, elev, 3 * res
    if kind == "narrow":  # fewer than 64 rows: k_normals_slide
        rows, cols, cells = 40, 180, 3
        elev = synth.perlin_elevation(rows, cols, seed=506, amplitude=0.9)
        return rows, cols, res, elev, synth.benchmark_radius(cells, res)
    if kind == "wide":  # radius 12 cells: k_normals_slide
        rows, cols, cells = 200, 150, 12
        elev = synth.perlin_elevation(rows, cols, seed=507, amplitude=1.5)
        return rows, cols, res, elev, synth.benchmark_radius(cells, res)
    if kind == "one_cell_tie":  # the default 0.05 m radius on a 0.05 m map: k_normals_small's tie folds (round 6)
        rows, cols = 600, 500
        elev = synth.perlin_elevation(rows, cols, seed=508, amplitude=2.5)
        return rows, cols, res, elev, res
    if kind == "one_cell_holes":  # the same with 2 % speckle: discs of three and four cells beside the collinear ones
        rows, cols = 600, 500
        elev = synth.with_holes(synth.perlin_elevation(rows, cols, seed=509, amplitude=2.5), 0.02, seed=11)
        return rows, cols, res, elev, res
    if kind == "window":  # a small launch, discs of reach 1, single-cell step windows: k_chain_window (round 6)
        rows, cols = 300, 200
        elev = synth.perlin_elevation(rows, cols, seed=510, amplitude=2.5)
        return rows, cols, res, elev, 1.2 * res
    raise ValueError(kind)


#: step windows of the cases that are not the file's usual two-cell ones
_STEP_RADIUS = {"window": 0.04}


@pytest.mark.parametrize("kind", ["slim", "clean", "sparse", "dense", "ties", "narrow", "wide", "one_cell_tie", "one_cell_holes", "window"])
def test_scores_at_their_clip(capi, oracle, kind):
    rows, cols, res, elev, radius = _case(kind)
    g = oracle.geom(rows, cols, res, (2.0, -3.0))
    step = _STEP_RADIUS.get(kind, 2 * res * 1.000001)
    base = dict(normals_radius=radius, rough_radius=radius, step_radius1=step, step_radius2=step,
                fp_radius=0.1, fp_offset=0.05, fp_check_roughness=1)
    op0 = oracle.default_params(slope_critical=1.0, rough_critical=0.05, **base)
    ref = oracle.chain(g, op0, elev, want_normals=True)
    nz = ref["surface_normal_z"].astype(np.float64)
    rough = (1.0 - ref["traversability_roughness"].astype(np.float64)) * 0.05  # roughness itself, good to 1e-9
    valid = np.isfinite(nz) & (nz < 0.9999) & (nz > 0.05)
    idx = np.flatnonzero(valid)
    assert idx.size > 1000
    # cells of the interior, of the frame and -- with holes -- next to holes all take part: quantiles of the slope
    order = idx[np.argsort(nz[idx])]
    picks = [order[int(q * (order.size - 1))] for q in (0.15, 0.5, 0.85)]
    n_at_clip = 0
    for c in picks:
        crit_s = float(np.arccos(nz[c]))  # SlopeFilter.cpp:74: the oracle's slope of cell c, so its score is exactly 0
        rv = rough[c] if np.isfinite(rough[c]) and rough[c] > 1e-4 else 0.02
        crit_r = float(rv)                # within 1e-9 of the oracle's roughness of cell c
        over = dict(base, slope_critical=crit_s, rough_critical=crit_r)
        got, want, op = both_fp(capi, oracle, elev, rows, cols, res, pos=(2.0, -3.0), **over)
        check_fp(got, want, op, f"{kind}: critical values of cell {c}: slope {crit_s!r} roughness {crit_r!r}")
        for k in ("traversability_slope", "traversability_roughness"):
            a, b = got[k].reshape(-1), want[k].reshape(-1)
            assert np.array_equal(a == 0, b == 0), (kind, k, "zero / non-zero pattern")
            # where the oracle's score is within the fast tails' band of the clip the values are the generic kernel's: the
            # same double arithmetic up to the order of its sums (1e-15 relative), not a float32 tail's (1e-7)
            near = np.isfinite(b) & (np.abs(b) < 1e-6) & (b != 0)
            d = np.abs(a[near].astype(np.float64) - b[near].astype(np.float64))
            assert d.size == 0 or d.max() <= 1e-10, (kind, k, "cells at the clip are settled by the generic arithmetic", float(d.max()))
            n_at_clip += int(near.sum()) + int((b[np.isfinite(b)] == 0).sum() > 0)
        assert want["traversability_slope"].reshape(-1)[c] == 0.0
    assert n_at_clip >= len(picks)


def test_scores_at_their_clip_given_normals(capi, oracle):
    """RoughnessFilter as a stand-alone plugin (normals are input layers): the sliding kernel's closed form n^T C n has the
    same clip, and the same escape."""
    from traversability_estimation_amd import synth
    rows, cols, res = 200, 150, 0.05
    elev = synth.perlin_elevation(rows, cols, seed=511, amplitude=0.9)
    radius = synth.benchmark_radius(4, res)
    g = oracle.geom(rows, cols, res)
    op0 = oracle.default_params(normals_radius=radius, rough_radius=radius, rough_critical=0.05)
    ref = oracle.chain(g, op0, elev, want_normals=True)
    rough = (1.0 - ref["traversability_roughness"].astype(np.float64)) * 0.05
    idx = np.flatnonzero(np.isfinite(rough) & (rough > 1e-3))
    order = idx[np.argsort(rough[idx])]
    for q in (0.3, 0.7):
        c = order[int(q * (order.size - 1))]
        crit_r = float(rough[c])
        op = oracle.default_params(normals_radius=radius, rough_radius=radius, rough_critical=crit_r)
        want = oracle.chain(g, op, elev)
        with capi.Context(0) as ctx:
            ctx.set_params(to_te_params(capi, op))
            ctx.set_geometry(rows, cols, 1, res)
            ctx.upload_elevation(elev)
            for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z"):
                ctx.upload_layer(k, ref[k])
            ctx.run_filter("roughness")
            ctx.sync()
            got = ctx.download("traversability_roughness").reshape(-1)
        b = want["traversability_roughness"].reshape(-1)
        assert np.array_equal(np.isnan(got), np.isnan(b))
        assert np.array_equal(got == 0, b == 0), "zero / non-zero pattern of the roughness plugin"
        ok = np.isfinite(b)
        assert np.abs(got[ok].astype(np.float64) - b[ok]).max() <= 1e-5


def test_seed_6193_memo_layers(capi, oracle):
    """The case round 4's sweep found: one cell of 250 601 whose slope score was 0 here and 1.8e-6 in the oracle, so
    slope_footprint had NaN on one side and 0 / 1 on the other."""
    from tests.test_gpu_random import draw_case
    seed = 6193
    rows, cols, res, pos, elev, over = draw_case(seed)
    if (over["fp_radius"] + over["fp_offset"]) / res > 19.5:
        over["fp_offset"] = 0.0
    got, want, op = both_fp(capi, oracle, elev, rows, cols, res, pos=pos, **over)
    check_fp(got, want, op, f"random case seed {seed}")
    for k in ("slope_footprint", "step_footprint", "roughness_footprint"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k


def test_whole_map_oracle_at_res_005_with_boxes_and_speckle(capi, oracle):
    """A whole map at the BASELINE resolution against the oracle, tie-free radius 5 cells, 40 boxes and 0.1 % speckle: all
    five layers and the three memo layers (NaN patterns included).  (The 4096^2 obstacle / hole tests run at res = 2^-4:
    their oracle sees bands, and a band does not share the rounded absolute positions that decide checkForStep's ties at
    0.05 m; here the oracle sees the whole map.)"""
    from traversability_estimation_amd import synth
    n, res = 1024, 0.05
    elev = synth.perlin_elevation(n, n, seed=1240)
    rng = np.random.default_rng(7)
    for _ in range(40):
        h, w = (int(v) for v in rng.integers(4, 40, size=2))
        r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
        elev[c0:c0 + w, r0:r0 + h] += np.float32(rng.uniform(0.15, 0.5) * rng.choice([-1.0, 1.0]))
    elev = synth.with_holes(elev, 0.001, seed=99)
    r = synth.benchmark_radius(5, res)
    over = dict(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=synth.benchmark_radius(6, res),
                fp_offset=synth.benchmark_radius(3, res), fp_check_roughness=1)
    oracle.set_threads(min(os.cpu_count() or 1, 128))
    try:
        got, want, op = both_fp(capi, oracle, elev, n, n, res, **over)
    finally:
        oracle.set_threads(1)
    check_fp(got, want, op, "1024^2 at res 0.05, 40 boxes, 0.1 % speckle: whole map against the oracle")
    for k in ("slope_footprint", "step_footprint", "roughness_footprint"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k
    assert (want["traversability_footprint"] == 0).sum() > 1000 and np.isnan(want["traversability_slope"]).sum() > 500


def test_region_footprint_with_a_shape_only_the_general_kernel_takes(capi, oracle):
    """A dirty region of one map of a batch when the footprint's reach (17 cells) is beyond the shape-specialised sum
    kernels: the general kernel recomputes that map's footprint layer -- and only that map's."""
    from traversability_estimation_amd import synth
    rows, cols, res, B = 160, 130, 0.05, 3
    maps = [synth.perlin_elevation(rows, cols, seed=900 + b, amplitude=0.3) for b in range(B)]
    r = synth.benchmark_radius(3, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(14, res), fp_offset=synth.benchmark_radius(3, res))
    g = oracle.geom(rows, cols, res)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, B, res)
        ctx.upload_elevation(np.stack(maps))
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        before = [ctx.download("traversability_footprint", b, 1) for b in range(B)]
        patch = synth.with_steps(synth.perlin_elevation(30, 24, seed=77, amplitude=0.4), 2, seed=78)
        maps[1] = maps[1].copy()
        maps[1][40:40 + 24, 50:50 + 30] = patch
        ctx.upload_tile(patch, 1, 50, 40)
        ctx.run_chain_region(1, 50, 40, 30, 24, flags=capi.RUN_FOOTPRINT)
        ctx.sync()
        after = [ctx.download("traversability_footprint", b, 1) for b in range(B)]
        got = {k: ctx.download(k, 1, 1) for k in OUT_LAYERS + ("traversability_footprint",)}
    for b in (0, 2):
        assert np.array_equal(before[b], after[b], equal_nan=True), f"map {b} is not the region's map"
    want = oracle.chain(g, op, maps[1])
    want["traversability_footprint"] = oracle.footprint(g, op, maps[1], want)
    assert_layers_match(got, want, layers=OUT_LAYERS + ("traversability_footprint",), ctx="region run, general footprint kernel")


def test_prefetched_layers_equal_uploaded_layers(capi, oracle):
    """te_prefetch_layers: the reference's three plugins in the unchanged-YAML order with the next plugin's inputs sent
    beside each plugin's filter and download -- same layers as with one upload at a time, and as the oracle's."""
    from traversability_estimation_amd import synth
    rows, cols, res = 1536, 1200, 0.05  # (layers above the 4 MiB below which transfers are not staged)
    elev = synth.perlin_elevation(rows, cols, seed=77, amplitude=0.6)
    r = synth.benchmark_radius(4, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
    g = oracle.geom(rows, cols, res)
    oracle.set_threads(min(os.cpu_count() or 1, 64))
    try:
        want = oracle.chain(g, op, elev, want_normals=True)
    finally:
        oracle.set_threads(1)
    nrm = {k: want[k] for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z")}
    outs = {}
    for prefetch in (False, True):
        with capi.Context(0) as ctx:
            ctx.set_params(to_te_params(capi, op))
            ctx.set_geometry(rows, cols, 1, res)
            ctx.upload_layer("surface_normal_z", nrm["surface_normal_z"])
            if prefetch:
                ctx.prefetch_layers({"elevation": elev})
            ctx.run_filter("slope")
            a = ctx.download("traversability_slope")
            if prefetch:
                ctx.wait_prefetch()
                ctx.prefetch_layers({"surface_normal_x": nrm["surface_normal_x"], "surface_normal_y": nrm["surface_normal_y"]})
            else:
                ctx.upload_elevation(elev)
            ctx.run_filter("step")
            b = ctx.download("traversability_step")
            if prefetch:
                ctx.wait_prefetch()
            else:
                ctx.upload_layer("surface_normal_x", nrm["surface_normal_x"])
                ctx.upload_layer("surface_normal_y", nrm["surface_normal_y"])
            ctx.run_filter("roughness")
            c = ctx.download("traversability_roughness")
            ctx.sync()
            outs[prefetch] = {"traversability_slope": a, "traversability_step": b, "traversability_roughness": c}
    for k in outs[True]:
        assert np.array_equal(outs[True][k], outs[False][k], equal_nan=True), k
    assert_layers_match(outs[True], want, layers=("traversability_slope", "traversability_step", "traversability_roughness"),
                        ctx="three plugins with prefetched inputs")
    # a prefetch that nobody waits for is finished by the next call that needs the context to itself
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ctx.prefetch_layers({"elevation": elev})
        ctx.run_chain(0)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
        ctx.wait_prefetch()
    assert_layers_match(got, want, ctx="chain after an unawaited prefetch")


@pytest.mark.parametrize("kind,amount", [("boxes", 300), ("speckle", 0.001)])
def test_whole_4096_map_against_the_oracle_at_res_005(capi, oracle, kind, amount):
    """The BASELINE size AND resolution, every cell: 4096^2 at 0.05 m, radius 9 cells, footprint 6 + 3 cells, with 300 boxes
    (k_fp_mask's memoised checkForStep, the inner disc, k_fp_blocked) or 0.1 % speckle (the sparse-hole march) -- all five
    layers and the memo layers against the OpenMP oracle on the WHOLE map (no crop: checkForStep's position-rounding ties are
    the map's own).  The oracle takes some ten seconds on the GPU box's host cores."""
    from traversability_estimation_amd import synth
    from tests.test_gpu_fullsize import _rough_map
    n, res = 4096, 0.05
    elev = _rough_map(synth, n, 1235, kind, amount)
    r = synth.benchmark_radius(9, res)
    over = dict(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=synth.benchmark_radius(6, res),
                fp_offset=synth.benchmark_radius(3, res))
    oracle.set_threads(min(os.cpu_count() or 1, 64))
    try:
        got, want, op = both_fp(capi, oracle, elev, n, n, res, **over)
    finally:
        oracle.set_threads(1)
    check_fp(got, want, op, f"4096^2 at res 0.05, {kind} {amount}: whole map against the oracle")
    for k in ("slope_footprint", "step_footprint"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k


@pytest.mark.parametrize("cells,keep", [(9, False), (5, True), (3, False)])
def test_unobserved_regions(capi, oracle, cells, keep):
    """Unobserved REGIONS (counted at upload, too many cells for the sparse march): k_normals3's dense march on strips of 32
    rows -- more blocks than resident slots --, the clean first attempt handing over at the row it reached, and the steps
    inside a region that find not one cell in the ring's window (nothing to compute).  Regions in the interior, in a
    corner of the map, a band narrower than a strip's window, a column of the map, and a few scattered cells; all layers
    (and the surface normals with TE_RUN_KEEP_NORMALS) against the oracle on the whole map."""
    from traversability_estimation_amd import synth
    rows, cols, res = 600, 520, 0.05
    elev = synth.perlin_elevation(rows, cols, seed=700 + cells, amplitude=0.6).copy()
    a0, a1 = elev.shape
    elev[int(0.30 * a0):int(0.72 * a0), int(0.25 * a1):int(0.70 * a1)] = np.nan   # interior, wider and taller than a window
    elev[:int(0.22 * a0), :int(0.3 * a1)] = np.nan                                # a corner of the map
    elev[int(0.80 * a0):int(0.84 * a0), int(0.1 * a1):int(0.9 * a1)] = np.nan     # a band one way ...
    elev[int(0.1 * a0):int(0.95 * a0), int(0.86 * a1):int(0.89 * a1)] = np.nan    # ... and the other
    elev[:, -1] = np.nan
    elev[-1, :] = np.inf
    rng = np.random.default_rng(5)
    for _ in range(12):
        elev[rng.integers(0, a0), rng.integers(0, a1)] = np.nan
    assert 0.2 < np.mean(~np.isfinite(elev)) < 0.5
    r = synth.benchmark_radius(cells, res)
    over = dict(normals_radius=r, rough_radius=r, step_radius1=synth.benchmark_radius(3, res), step_radius2=synth.benchmark_radius(3, res),
                fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))
    if not keep:
        got, want, op = both_fp(capi, oracle, elev, rows, cols, res, pos=(1.0, 2.0), **over)
        check_fp(got, want, op, f"unobserved regions, radius {cells} cells")
        return
    op = oracle.default_params(**over)
    g = oracle.geom(rows, cols, res, (1.0, 2.0))
    want = oracle.chain(g, op, elev, want_normals=True)
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res, (1.0, 2.0))
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_KEEP_NORMALS)
        ctx.sync()
        layers = list(OUT_LAYERS) + ["surface_normal_x", "surface_normal_y", "surface_normal_z"]
        got = {k: ctx.download(k) for k in layers}
    assert_layers_match(got, want, layers=layers, ctx=f"unobserved regions, normals kept, radius {cells} cells")
