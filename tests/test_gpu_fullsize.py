"""Full-size checks (BASELINE.json configs 2-5 shapes) through size-independent properties: the CPU oracle
needs minutes at these sizes, so the full maps are checked with (a) the shape-generic kernels, an
independent device implementation that IS checked against the oracle cell by cell in test_gpu_chain.py,
(b) crops against the oracle at the corners, an edge and the centre, (c) determinism / idempotence of the
dirty-region path, (d) the combine identity."""
import os

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


def bench_params(capi, synth, cells, res, ties=False):
    if ties:  # every radius a whole number of cells: the cells on the circles are decided centre by centre
        return capi.default_params(normals_radius=cells * res, rough_radius=cells * res, step_radius1=cells * res, step_radius2=cells * res,
                                   fp_radius=6 * res, fp_offset=3 * res)
    r = synth.benchmark_radius(cells, res)
    return capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))


def oracle_params(oracle, p):
    op = oracle.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    return op


@pytest.mark.parametrize("n,cells,seed,ties", [(1024, 5, 1234, False), (4096, 9, 1235, False), (1024, 5, 1236, True), (4096, 9, 1237, True)])
def test_full_map_fast_vs_generic_and_oracle_crops(capi, oracle, n, cells, seed, ties):
    """configs[1] (1024^2, radius 5) and configs[2] (4096^2, radius 9 + footprint).  ties: the same maps with every radius
    a whole number of cells (the TIES march, the step folds, the fixed-point footprint's tie variant, hipGraph replay and
    the two streams at full size) against the generic kernels, which the oracle checks cell by cell on small maps; a
    crop cannot serve there: the decisions on the circles depend on the rounded cell positions, which a crop does not share."""
    from traversability_estimation_amd import synth
    res = 0.05
    elev = synth.perlin_elevation(n, n, seed=seed)
    if ties:  # something for the footprint's blocked discs and the step filter to find
        rng = np.random.default_rng(seed)
        for _ in range(40):
            h, w = (int(v) for v in rng.integers(4, 40, size=2))
            r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
            elev[c0:c0 + w, r0:r0 + h] += np.float32(0.3)
    p = bench_params(capi, synth, cells, res, ties)
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
    if ties:
        return
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


def test_cfg4_true_size_batch_of_512_maps(capi, oracle):
    """BASELINE configs[3] at its own size: 512 maps of 512 x 512, radius 5, one launch (the batch axis that the ranks
    shard).  The MPC-rollout shape: one base map + N(0, 1 cm) perturbations, seed = map index.  Eight maps spread over
    the batch against the oracle (chain + footprint); a map uploaded twice gives the same bits in both slots."""
    from traversability_estimation_amd import synth
    rows = cols = 512
    res, B = 0.05, 512
    base = synth.perlin_elevation(rows, cols, seed=2000)
    per = rows * cols
    p = bench_params(capi, synth, 5, res)
    picks = (0, 1, 63, 200, 255, 256, 400, 511)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, B, res)
        kept = {}
        for b0 in range(0, B, 64):  # uploaded in blocks of 64 maps (64 MiB of host memory at a time)
            blk = np.stack([(base + np.random.default_rng(2000 + b).normal(0.0, 0.01, size=base.shape).astype(np.float32)).astype(np.float32)
                            for b in range(b0, b0 + 64)])
            for b in picks:
                if b0 <= b < b0 + 64:
                    kept[b] = blk[b - b0].copy()
            if b0 == 256:
                blk[44] = kept[200]  # slot 300 holds map 200 again
            ctx.upload_elevation(blk, map0=b0)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        got = {k: ctx.download(k) for k in ALL}
    for k in ALL:
        a = got[k]
        assert np.array_equal(a[200 * per:201 * per].view(np.uint32), a[300 * per:301 * per].view(np.uint32)), k
    op = oracle_params(oracle, p)
    oracle.set_threads(8)
    try:
        g = oracle.geom(rows, cols, res)
        for b in picks:
            want = oracle.chain(g, op, kept[b])
            want["traversability_footprint"] = oracle.footprint(g, op, kept[b], want)
            assert_layers_match({k: got[k][b * per:(b + 1) * per] for k in ALL}, want, layers=ALL, ctx=f"cfg4 map {b} of 512")
    finally:
        oracle.set_threads(1)


def test_cfg5_true_size_streaming_8192(capi, oracle):
    """BASELINE configs[4] at its own size: an 8192 x 8192 resident map, 256 x 256 dirty tiles, re-filter of the dirty
    region incl. the footprint pass.  After every tick the oracle runs on a crop around the tile that holds the tile's
    whole zone of influence (chain reach 10, mask 3, footprint 9 cells, plus as much again for the crop's cut edges)."""
    from traversability_estimation_amd import synth
    # res = 2^-4 m: every cell position is exact in double, so checkForStep's geometric comparisons (perpendicular
    # directions, ray lengths) come out the same in a crop as in the whole map -- at 0.05 m they are rounding ties that
    # depend on the absolute position of the map, in the reference as here
    n, res, tile = 8192, 0.0625, 256
    # (a 2048^2 noise map repeated 4 x 4 under a slow ramp: the generator needs a minute for 8192^2, the content is beside the point)
    a = synth.perlin_elevation(2048, 2048, seed=77).reshape(2048, 2048)
    elev = np.tile(np.block([[a, a[:, ::-1]], [a[::-1, :], a[::-1, ::-1]]]), (2, 2))  # mirrored: no cliffs at the seams
    elev = (elev + np.linspace(0.0, 1.5, n, dtype=np.float32)[None, :]).astype(np.float32)
    p = bench_params(capi, synth, 5, res)
    op = oracle_params(oracle, p)
    rng = np.random.default_rng(78)
    reach = 2 * 5 + 3 + 9 + 6  # cells beyond the tile whose outputs can change
    margin = reach + 2 * 5 + 12  # cells of a crop that see its cut edge
    oracle.set_threads(8)
    try:
        with capi.Context(0) as ctx:
            ctx.set_params(p)
            ctx.set_geometry(n, n, 1, res)
            ctx.upload_elevation(elev)
            ctx.run_chain(capi.RUN_FOOTPRINT)
            for tick in range(5):
                r0, c0 = (int(v) for v in rng.integers(0, n - tile, size=2))
                if tick == 1:
                    r0, c0 = n - tile, 0          # a corner
                if tick == 2:
                    r0, c0 = 4000, n - tile - 3   # three cells from the right border
                patch = (synth.perlin_elevation(tile, tile, seed=3000 + tick).reshape(tile, tile) * np.float32(0.6)).astype(np.float32)
                elev[c0:c0 + tile, r0:r0 + tile] = patch
                ctx.upload_tile(np.ascontiguousarray(patch), 0, r0, c0)
                ctx.run_chain_region(0, r0, c0, tile, tile, flags=capi.RUN_FOOTPRINT)
                ctx.sync()
                # crop = zone of influence + margin, clipped to the map
                i_lo, i_hi = max(0, r0 - reach - margin), min(n, r0 + tile + reach + margin)
                j_lo, j_hi = max(0, c0 - reach - margin), min(n, c0 + tile + reach + margin)
                crop = np.ascontiguousarray(elev[j_lo:j_hi, i_lo:i_hi])
                g = oracle.geom(i_hi - i_lo, j_hi - j_lo, res)
                want = oracle.chain(g, op, crop)
                want["traversability_footprint"] = oracle.footprint(g, op, crop, want)
                ki = slice(0 if i_lo == 0 else margin, (i_hi - i_lo) if i_hi == n else (i_hi - i_lo) - margin)
                kj = slice(0 if j_lo == 0 else margin, (j_hi - j_lo) if j_hi == n else (j_hi - j_lo) - margin)
                for k in ALL:
                    a = ctx.download_tile(k, 0, i_lo, j_lo, i_hi - i_lo, j_hi - j_lo)[kj, ki]
                    b = want[k].reshape(j_hi - j_lo, i_hi - i_lo)[kj, ki]
                    n_bad, mx, _ = compare_layer(k, a, b)
                    assert n_bad == 0, (tick, k, n_bad, mx)
    finally:
        oracle.set_threads(1)


def _rough_map(synth, n, seed, kind, amount):
    """The 4096^2 maps round 3 timed without a parity check (profiles/r03_holes.json, r03_obstacles.json), as ab_chain.py builds them."""
    elev = synth.perlin_elevation(n, n, seed=seed)
    if kind == "speckle":
        return synth.with_holes(elev, amount, seed=99)
    rng = np.random.default_rng(99 if kind == "unobserved" else 7)
    if kind == "unobserved":  # rectangles of 100..400 cells a side until `amount` of the area is covered
        area = 0
        while area < amount * n * n:
            h, w = (int(v) for v in rng.integers(100, 400, size=2))
            r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
            elev[c0:c0 + w, r0:r0 + h] = np.nan
            area += h * w
        return elev
    for _ in range(int(amount)):  # boxes: raised / lowered rectangles of 4..40 cells a side (kerbs, crates)
        h, w = (int(v) for v in rng.integers(4, 40, size=2))
        r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
        elev[c0:c0 + w, r0:r0 + h] += np.float32(rng.uniform(0.15, 0.5) * rng.choice([-1.0, 1.0]))
    return elev


@pytest.mark.parametrize("kind,amount", [("speckle", 0.001), ("speckle", 0.01), ("unobserved", 0.15), ("boxes", 300), ("boxes", 3000)])
def test_full_size_holes_and_obstacles_against_oracle_bands(capi, oracle, kind, amount):
    """The paths round 3 optimised and timed at 4096^2 without an oracle check at that size: the sparse march and its hole
    queue (0.1 % speckle), the dense march (1 % speckle, unobserved rectangles), k_fp_blocked's long list and the mask
    kernel's tile-wide list of slow cells (300 / 3000 boxes) -- on the bench-sized map, against the oracle on full-width
    and full-height bands (every block column and every strip boundary is crossed).  res = 2^-4 m: checkForStep's
    geometric ties depend on the rounded ABSOLUTE cell positions at 0.05 m, which a band does not share with the map
    (DESIGN.md section 2); at a dyadic resolution every position is exact and band and map agree on every cell."""
    import os
    from traversability_estimation_amd import synth
    n, cells, res = 4096, 9, 2.0 ** -4
    elev = _rough_map(synth, n, 1235, kind, amount)
    p = bench_params(capi, synth, cells, res)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(n, n, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        fast = {k: ctx.download(k) for k in ALL}
    if kind == "boxes":  # the obstacle paths really ran: discs with an untraversable cell, partial footprint values
        fp = fast["traversability_footprint"]
        assert (fp == 0).sum() > 1000 * amount / 300 and ((fp > 0) & (fp < 0.3)).sum() > 0
    else:
        assert np.isnan(fast["traversability_slope"]).sum() >= np.isnan(elev).sum() > 0
    op = oracle_params(oracle, p)
    margin, band = 2 * cells + 10 + 8, 96  # (+ max_gap_width / res + 3: the rays of checkForStep)
    img = lambda a: a.reshape(n, n)
    oracle.set_threads(min(os.cpu_count() or 1, 64))
    try:
        for axis, starts in ((0, (0, 1777, n - band)), (1, (1300, n - band))):
            for s0 in starts:
                sl = (slice(s0, s0 + band), slice(0, n)) if axis == 0 else (slice(0, n), slice(s0, s0 + band))
                crop = np.ascontiguousarray(elev[sl])
                g = oracle.geom(crop.shape[1], crop.shape[0], res)
                want = oracle.chain(g, op, crop)
                want["traversability_footprint"] = oracle.footprint(g, op, crop, want)
                lo = 0 if s0 == 0 else margin
                hi = band if s0 + band == n else band - margin
                keep = (slice(lo, hi), slice(0, n)) if axis == 0 else (slice(0, n), slice(lo, hi))
                for k in ALL:
                    a = img(fast[k])[sl][keep]
                    b = want[k].reshape(crop.shape)[keep]
                    n_bad, mx, nn = compare_layer(k, a, b)
                    assert n_bad == 0, (kind, amount, k, "band", axis, s0, n_bad, mx, nn)
    finally:
        oracle.set_threads(1)


def test_whole_map_oracle_at_tie_radii_1024(capi, oracle):
    """Tie radii at a BASELINE size against the ORACLE on the whole map (the 4096^2 tie test above compares fast with generic
    GPU kernels: a crop does not share the rounded positions that decide the cells on the circles): 1024^2, every radius
    exactly 5 cells (a 3-4-5 radius: twelve circle cells), footprint 6 + 3 cells exactly, 40 boxes; the OpenMP oracle
    takes about a minute on the GPU box's host cores."""
    import os
    from traversability_estimation_amd import synth
    n, res = 1024, 0.05
    elev = _rough_map(synth, n, 1236, "boxes", 40)
    p = bench_params(capi, synth, 5, res, ties=True)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(n, n, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_FOOTPRINT)
        ctx.sync()
        fast = {k: ctx.download(k) for k in ALL}
    op = oracle_params(oracle, p)
    oracle.set_threads(min(os.cpu_count() or 1, 128))
    try:
        g = oracle.geom(n, n, res)
        want = oracle.chain(g, op, elev)
        want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
    finally:
        oracle.set_threads(1)
    assert_layers_match(fast, want, layers=ALL, ctx="1024^2 at tie radii, whole map against the oracle")


def test_batch_with_sparse_holes(capi, oracle):
    """A batch whose maps carry sensor speckle (0.1 %: the sparse-hole march of k_normals3 and its per-block queues).  With
    96 maps of 512^2 the grid no longer fits one round of resident blocks, which the queue scratch is sized for: the
    launcher must fall back to the dense march instead of indexing queues that do not exist."""
    from traversability_estimation_amd import synth
    rows = cols = 512
    res, B = 0.05, 96
    elevs = np.stack([synth.with_holes(synth.perlin_elevation(rows, cols, seed=2000 + b), 0.001, seed=b) for b in range(B)])
    p = bench_params(capi, synth, 5, res)
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, B, res)
        ctx.upload_elevation(elevs)
        ctx.run_chain(0)
        ctx.sync()
        fast = {k: ctx.download(k) for k in OUT_LAYERS}
    op = oracle_params(oracle, p)
    oracle.set_threads(8)
    try:
        g = oracle.geom(rows, cols, res)
        per = rows * cols
        for b in (0, 41, 95):
            want = oracle.chain(g, op, elevs[b])
            assert_layers_match({k: fast[k][b * per:(b + 1) * per] for k in OUT_LAYERS}, want, ctx=f"sparse holes, map {b}")
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("n", [16384, 28672, 32768])
def test_maps_of_2_28_and_2_30_cells(n):
    """Maximum sizes: ONE map of 16384^2 (layers of 1 GiB: the marching kernels, byte offsets up to 2^30), of 28672^2 (3.3 GB:
    the marching kernels beyond 2^31 bytes -- they re-base their 32-bit offsets strip by strip) and of 32768^2
    (layers of 4 GiB, a 62 GiB slab: byte offsets cross 2^31 and reach 2^32 -- the marching kernels address a map with
    32-bit byte offsets and hand such a map to the double kernels, te_normals3.hip / te_footprint5.hip: `>= 4294967296.0`).
    Chain + footprint pass at the bench radii; crops of 192 x 192 cells against the oracle at the corners, in the middle
    and around the columns whose byte offsets cross 2^30 / 2^31 / 2^32 (tools/dbg/large_map.py, its own process: the host
    arrays are 1 and 4 GiB)."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", "large_map.py"), str(n)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "mismatching cells: 0" in r.stdout
    assert r.stdout.count(": checked") >= 4


def test_batch_whose_layers_exceed_4_gib():
    """The batch axis sized for the device's memory: 4200 maps of 512 x 512 in one launch (4.4 GB per layer, a 63 GB slab).
    Map offsets are 64-bit, offsets within a map 32-bit: the maps on both sides of the 2^31- and 2^32-byte marks, the first
    and the last against the oracle (chain + footprint), one map in two slots bit-equal (tools/dbg/large_batch.py)."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", "large_batch.py"), "4200"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("checked") == 7 and r.stdout.strip().endswith("ok")
