"""Round-4 kernels through the C-ABI against the oracle:
k_normals3s -- the clean march of the normals kernel on its slim ring (2R rows instead of 2R + 2; shapes whose centre
column alone reaches rows j +- R; launches whose elevation layer the upload counted free of invalid cells) -- over every
such shape, over maps with several strips and block columns, on a batch, and with an invalid cell that turns up although
the count said none (the strip then goes to the fix-up pass);
launch3's strip height for grids beyond one round of resident blocks (a batch whose maps outnumber the wave slots)."""
import ctypes

import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, to_te_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def run_chain(capi, p, rows, cols, batch, res, elevs, pos=(0.0, 0.0), flags=0):
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, batch, res, pos)
        ctx.upload_elevation(elevs)
        ctx.run_chain(flags)
        ctx.sync()
        return {k: ctx.download(k) for k in OUT_LAYERS}


@pytest.mark.parametrize("cells,rows,cols", [(2, 130, 97), (3, 200, 150), (4, 257, 301), (5, 300, 260), (6, 190, 410), (7, 333, 222),
                                              (8, 150, 700), (9, 700, 130), (9, 320, 280), (10, 270, 330)])
def test_slim_ring_shapes_against_the_oracle(capi, oracle, cells, rows, cols):
    """Radii a hair above a whole number of cells (the shapes Q = R^2: only the centre column is 2R + 1 cells high) on
    hole-free maps: every block kind of the kernel -- interior strips, the left / right frame's block columns, the top
    and bottom frame rows, a shifted last block column."""
    from traversability_estimation_amd import synth
    res = 0.05
    elev = synth.with_steps(synth.perlin_elevation(rows, cols, seed=cells * 100 + rows, amplitude=0.25), 5, seed=cells)
    r = synth.benchmark_radius(cells, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
    g = oracle.geom(rows, cols, res, (-3.2, 7.9))
    want = oracle.chain(g, op, elev)
    got = run_chain(capi, to_te_params(capi, op), rows, cols, 1, res, elev, (-3.2, 7.9))
    assert_layers_match(got, want, ctx=f"slim ring, R = {cells}, {rows} x {cols}")
    for k in OUT_LAYERS:
        assert not np.isnan(got[k]).any(), k


def test_slim_ring_on_a_batch_and_beyond_one_round_of_blocks(capi, oracle):
    """Hole-free batches: a few maps (strips sized for one round), and more maps than the device has wave slots for
    (launch3's cost model picks the strip height: 4000 maps of 64 x 80 against 3072 slots)."""
    from traversability_estimation_amd import synth
    res = 0.05
    for rows, cols, batch, cells, check in ((200, 170, 4, 9, (0, 3)), (64, 80, 4000, 2, (0, 777, 3999))):
        rng = np.random.default_rng(batch)
        base = [synth.perlin_elevation(rows, cols, seed=50 + b, amplitude=0.2) for b in range(min(batch, 8))]
        elevs = np.stack([base[b % len(base)] + np.float32(0.01 * (b // len(base))) for b in range(batch)])
        r = synth.benchmark_radius(cells, res)
        op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
        got = run_chain(capi, to_te_params(capi, op), rows, cols, batch, res, elevs)
        g = oracle.geom(rows, cols, res, (0.0, 0.0))
        for b in check:
            want = oracle.chain(g, op, elevs[b])
            one = {k: got[k].reshape(batch, cols, rows)[b] for k in OUT_LAYERS}
            assert_layers_match(one, want, ctx=f"batch of {batch}, map {b}")


def test_an_invalid_cell_the_upload_did_not_count(capi, oracle):
    """The slim march is chosen from the count taken at upload.  A caller that keeps an elevation pointer from an earlier
    te_device_ptr and writes through it after a later upload defeats the count: the march meets the invalid cells, leaves
    its strip to the fix-up pass, and the layers are still the oracle's."""
    from traversability_estimation_amd import synth
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        pytest.skip("libamdhip64 not loadable from python")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rows, cols, res = 300, 260, 0.05
    elev = synth.perlin_elevation(rows, cols, seed=77, amplitude=0.2)
    r = synth.benchmark_radius(5, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
    poked = elev.copy()
    spots = [(40, 70), (41, 70), (150, 200), (255, 3), (3, 296), (120, 121)]  # (col index j, row index i) of the (cols, rows) array
    for j, i in spots:
        poked[j, i] = np.nan
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        ptr, nbytes = ctx.device_ptr("elevation")
        assert nbytes == rows * cols * 4
        ctx.upload_elevation(elev)  # counts: no invalid cell
        ctx.sync()
        nan = np.array([np.nan], np.float32)
        for j, i in spots:
            assert hip.hipMemcpy(ctypes.c_void_p(ptr + 4 * (j * rows + i)), nan.ctypes.data_as(ctypes.c_void_p), 4, 1) == 0
        ctx.run_chain(0)
        ctx.sync()
        got = {k: ctx.download(k) for k in OUT_LAYERS}
    want = oracle.chain(oracle.geom(rows, cols, res, (0.0, 0.0)), op, poked)
    assert_layers_match(got, want, ctx="invalid cells behind the count's back")


def test_untraversable_flags_follow_the_map(capi, oracle):
    """k_fp_mask's per-block flags ("holds an untraversable cell") decide whether k_fp_slide5 fetches a strip's mask bytes:
    they are rewritten by every pass over a tile, so a map with boxes, then a clean one, then boxes again in the same
    context give the oracle's footprint layer each time -- also when only a rectangle was re-filtered in between."""
    from traversability_estimation_amd import synth
    rows, cols, res = 330, 290, 0.05
    clean = synth.perlin_elevation(rows, cols, seed=9, amplitude=0.04)
    boxes = synth.with_steps(synth.perlin_elevation(rows, cols, seed=9, amplitude=0.04), 18, seed=3)
    r = synth.benchmark_radius(4, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))
    g = oracle.geom(rows, cols, res, (0.0, 0.0))
    want = {}
    for name, e in (("clean", clean), ("boxes", boxes)):
        w = oracle.chain(g, op, e)
        w["traversability_footprint"] = oracle.footprint(g, op, e, w)
        want[name] = w
    layers = list(OUT_LAYERS) + ["traversability_footprint"]
    with capi.Context(0) as ctx:
        ctx.set_params(to_te_params(capi, op))
        ctx.set_geometry(rows, cols, 1, res)
        for name, e in (("boxes", boxes), ("clean", clean), ("boxes", boxes), ("clean", clean)):
            ctx.upload_elevation(e)
            ctx.run_chain(capi.RUN_FOOTPRINT)
            ctx.sync()
            got = {k: ctx.download(k) for k in layers}
            assert_layers_match(got, want[name], layers=layers, ctx=f"{name} map after the other")
        assert not (got["traversability_footprint"] == 0).any()
        # a rectangle of the boxes map pasted into the clean one, re-filtered as a dirty region
        r0, c0, h, w = 100, 60, 90, 120
        mixed = clean.copy()
        mixed[c0:c0 + w, r0:r0 + h] = boxes[c0:c0 + w, r0:r0 + h]
        ctx.upload_tile(np.ascontiguousarray(mixed[c0:c0 + w, r0:r0 + h]), 0, r0, c0)
        ctx.run_chain_region(0, r0, c0, h, w, capi.RUN_FOOTPRINT)
        ctx.sync()
        got = {k: ctx.download(k) for k in layers}
    wm = oracle.chain(g, op, mixed)
    wm["traversability_footprint"] = oracle.footprint(g, op, mixed, wm)
    assert_layers_match(got, wm, layers=layers, ctx="boxes pasted into the clean map as a dirty region")
    assert (got["traversability_footprint"] == 0).sum() > 5
