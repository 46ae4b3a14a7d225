"""Full-size checks (BASELINE.json configs 2-5 shapes) through size-independent properties: the CPU oracle
needs minutes at these sizes, so the full maps are checked with (a) the shape-generic kernels, an
independent device implementation that IS checked against the oracle cell by cell in test_gpu_chain.py,
(b) crops against the oracle at the corners, an edge and the centre, (c) determinism / idempotence of the
dirty-region path, (d) the combine identity."""
import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, compare_layer, to_te_params

pytestmark = pytest.mark.gpu
ALL = OUT_LAYERS + ("traversability_footprint",)


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def bench_params(capi, synth, cells, res):
    r = synth.benchmark_radius(cells, res)
    return capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))


def oracle_params(oracle, p):
    op = oracle.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    return op


@pytest.mark.parametrize("n,cells,seed", [(1024, 5, 1234), (4096, 9, 1235)])
def test_full_map_fast_vs_generic_and_oracle_crops(capi, oracle, n, cells, seed):
    """configs[1] (1024^2, radius 5) and configs[2] (4096^2, radius 9 + footprint)."""
    from traversability_estimation_amd import synth
    res = 0.05
    elev = synth.perlin_elevation(n, n, seed=seed)
    p = bench_params(capi, synth, cells, res)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(n, n, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        fast = {k: ctx.download(k) for k in ALL}
        ctx.run_chain(capi.RUN_FOOTPRINT)  # determinism: a second run gives the same bits
        ctx.sync()
        for k in ALL:
            assert (ctx.download(k).view(np.uint32) == fast[k].view(np.uint32)).all(), k
        ctx.run_chain(capi.RUN_FOOTPRINT | capi.RUN_GENERIC_KERNELS | capi.RUN_SEQUENTIAL)
        ctx.sync()
        gen = {k: ctx.download(k) for k in ALL}
    assert_layers_match(fast, gen, layers=ALL, ctx=f"{n}^2 fast vs generic kernels")
    # combine identity on the full map (float32, left to right)
    t = np.float32(p.w_scale) * ((fast["traversability_slope"] + fast["traversability_step"]) + fast["traversability_roughness"])
    assert (t.view(np.uint32) == fast["traversability"].view(np.uint32)).all()
    # the chain's reach is 2*cells (step), the footprint adds 9: cells further than that from a crop's cut
    # edges see the same neighbourhood in the crop as in the full map
    m, margin = 160, 2 * cells + 10
    op = oracle_params(oracle, p)
    img = lambda a: a.reshape(n, n)  # [col j][row i]
    for (j0, i0) in ((0, 0), (n - m, n - m), (0, n // 2), (n // 2 - m // 2, n // 2 - m // 2)):
        crop = np.ascontiguousarray(elev[j0:j0 + m, i0:i0 + m])
        g = oracle.geom(m, m, res)
        want = oracle.chain(g, op, crop)
        want["traversability_footprint"] = oracle.footprint(g, op, crop, want)
        # keep only cells whose neighbourhood is not cut by the crop (map borders coincide where the crop touches them)
        lo_j = 0 if j0 == 0 else margin
        hi_j = m if j0 + m == n else m - margin
        lo_i = 0 if i0 == 0 else margin
        hi_i = m if i0 + m == n else m - margin
        for k in ALL:
            a = img(fast[k])[j0:j0 + m, i0:i0 + m][lo_j:hi_j, lo_i:hi_i]
            b = want[k].reshape(m, m)[lo_j:hi_j, lo_i:hi_i]
            n_bad, mx, _ = compare_layer(k, a, b)
            assert n_bad == 0, (k, (j0, i0), n_bad, mx)
    if n < 4096:
        return
    # the HBM-roofline map against the oracle on whole bands (full width / full height: every block column, every
    # strip boundary of the marching kernels is crossed), oracle on the host cores with OpenMP: 10 bands of 96 rows
    # or columns, 3.9 M of the 16.8 M cells
    import os
    band = 96
    oracle.set_threads(min(os.cpu_count() or 1, 64))
    try:
        for axis, starts in ((0, (0, 500, 1300, 2100, 3000, n - band)), (1, (0, 960, 2040, n - band))):
            for s0 in starts:
                sl = (slice(s0, s0 + band), slice(0, n)) if axis == 0 else (slice(0, n), slice(s0, s0 + band))
                crop = np.ascontiguousarray(elev[sl])
                g = oracle.geom(crop.shape[1], crop.shape[0], res)  # rows = extent along i (the fast axis)
                want = oracle.chain(g, op, crop)
                want["traversability_footprint"] = oracle.footprint(g, op, crop, want)
                lo = 0 if s0 == 0 else margin
                hi = band if s0 + band == n else band - margin
                keep = (slice(lo, hi), slice(0, n)) if axis == 0 else (slice(0, n), slice(lo, hi))
                for k in ALL:
                    a = img(fast[k])[sl][keep]
                    b = want[k].reshape(crop.shape)[keep]
                    n_bad, mx, _ = compare_layer(k, a, b)
                    assert n_bad == 0, (k, "band", axis, s0, n_bad, mx)
    finally:
        oracle.set_threads(1)


def test_batch_of_512_maps_shape(capi, oracle):
    """configs[3] shape (batch of independent 512x512 maps, radius 5) at a size that fits the test budget:
    32 maps resident at once, every map checked against the generic kernels, three against the oracle."""
    from traversability_estimation_amd import synth
    rows = cols = 512
    res, B = 0.05, 32
    elevs = np.stack([synth.perlin_elevation(rows, cols, seed=2000 + b) for b in range(B)])
    p = bench_params(capi, synth, 5, res)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, B, res)
        ctx.upload_elevation(elevs)
        ctx.run_chain(0)
        ctx.sync()
        fast = {k: ctx.download(k) for k in OUT_LAYERS}
        ctx.run_chain(capi.RUN_GENERIC_KERNELS)
        ctx.sync()
        gen = {k: ctx.download(k) for k in OUT_LAYERS}
    assert_layers_match(fast, gen, ctx="batch fast vs generic")
    op = oracle_params(oracle, p)
    oracle.set_threads(8)
    try:
        g = oracle.geom(rows, cols, res)
        per = rows * cols
        for b in (0, 17, 31):
            want = oracle.chain(g, op, elevs[b])
            assert_layers_match({k: fast[k][b * per:(b + 1) * per] for k in OUT_LAYERS}, want, ctx=f"map {b}")
    finally:
        oracle.set_threads(1)


def test_streaming_dirty_tiles(capi):
    """configs[4] shape: a large resident map, 256x256 dirty tiles re-filtered in place; after every update
    the incrementally maintained layers equal a from-scratch run on the same elevation (idempotence)."""
    from traversability_estimation_amd import synth
    n, res, tile = 2048, 0.05, 256
    elev = synth.perlin_elevation(n, n, seed=77)
    p = bench_params(capi, synth, 5, res)
    rng = np.random.default_rng(77)
    with capi.Context(0) as inc, capi.Context(0) as ref:
        for c in (inc, ref):
            c.set_params(p)
            c.set_geometry(n, n, 1, res)
        inc.upload_elevation(elev)
        inc.run_chain(0)
        for tick in range(4):
            r0, c0 = (int(v) for v in rng.integers(0, n - tile, size=2))
            if tick == 3:
                r0, c0 = 0, n - tile  # a tile in the corner
            patch = synth.perlin_elevation(tile, tile, seed=1000 + tick) * 0.5
            elev[c0:c0 + tile, r0:r0 + tile] = patch
            inc.upload_tile(patch, 0, r0, c0)
            inc.run_chain_region(0, r0, c0, tile, tile)
            inc.sync()
            ref.upload_elevation(elev)
            ref.run_chain(0)
            ref.sync()
            for k in OUT_LAYERS:
                a, b = inc.download(k), ref.download(k)
                n_bad, mx, _ = compare_layer(k, a, b)
                assert n_bad == 0, (tick, k, n_bad, mx)
