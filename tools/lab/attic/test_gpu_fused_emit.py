"""The fused emit of the second step pass (te_step5.hip, TE_OPT_NO_FUSED_EMIT = 0, the default): te_run_chain with
TE_RUN_FOOTPRINT runs that pass LAST and lets it write the combined layer and the untraversable mask of every cell
none of whose scores is 0 (TraversabilityMap.cpp:796, 869, 897: such a cell passes isTraversableForFilters without a
look at its neighbours); the mask kernel then rewrites only the tiles the emit flagged.  Checked against the oracle and,
bit for bit, against the launch sequence that leaves both to the mask kernel (TE_OPT_NO_FUSED_EMIT = 1)."""
import numpy as np
import pytest

from tests.helpers import OUT_LAYERS, assert_layers_match, to_te_params

pytestmark = pytest.mark.gpu
ALL = OUT_LAYERS + ("traversability_footprint",)


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def obstacle_map(synth, rows, cols, seed, boxes, amplitude=0.12):
    return synth.with_steps(synth.perlin_elevation(rows, cols, seed=seed, amplitude=amplitude), boxes, seed=seed + 7)


def run(capi, p, rows, cols, batch, res, elevs, fused, pos=(0.0, 0.0), flags=None, twice=None):
    with capi.Context(0) as ctx:
        ctx.set_option(capi.OPT_NO_FUSED_EMIT, 0 if fused else 1)
        ctx.set_params(p)
        ctx.set_geometry(rows, cols, batch, res, pos)
        if twice is not None:  # an earlier launch on another map leaves its flags / its mask behind
            ctx.upload_elevation(twice)
            ctx.run_chain(capi.RUN_FOOTPRINT if flags is None else flags)
        ctx.upload_elevation(elevs)
        ctx.run_chain(capi.RUN_FOOTPRINT if flags is None else flags)
        ctx.sync()
        return {k: ctx.download(k) for k in ALL}


@pytest.mark.parametrize("rows,cols,radius_cells,boxes,holes", [(230, 190, 3, 10, True), (300, 260, 5, 0, False), (200, 333, 9, 14, True),
                                                                (640, 520, 4, 40, False), (100, 133, 2, 6, True)])
def test_fused_emit_against_the_oracle(capi, oracle, rows, cols, radius_cells, boxes, holes):
    from traversability_estimation_amd import synth
    res = 0.05
    elev = obstacle_map(synth, rows, cols, 40 + rows, boxes)
    if holes:
        elev[cols // 3:cols // 3 + 5, rows // 4:rows // 4 + 17] = np.nan
        elev = synth.with_holes(elev, 0.002, seed=5)
    r = synth.benchmark_radius(radius_cells, res)
    op = oracle.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(5, res), fp_offset=synth.benchmark_radius(2, res))
    g = oracle.geom(rows, cols, res, (1.5, -2.0))
    want = oracle.chain(g, op, elev)
    want["traversability_footprint"] = oracle.footprint(g, op, elev, want)
    p = to_te_params(capi, op)
    got = run(capi, p, rows, cols, 1, res, elev, True, (1.5, -2.0))
    assert_layers_match(got, want, layers=list(ALL), ctx=f"fused emit, R = {radius_cells}, {boxes} boxes")
    old = run(capi, p, rows, cols, 1, res, elev, False, (1.5, -2.0))
    for k in ALL:
        assert np.array_equal(got[k], old[k], equal_nan=True), f"{k}: fused emit and mask-kernel combine differ"
    if boxes:
        assert (got["traversability_footprint"] == 0).sum() > 20


def test_fused_emit_after_a_launch_on_another_map(capi, oracle):
    """The flags are consumed by the launch that set them and every launch rewrites every mask byte: the result on a clean
    map does not depend on the cluttered map filtered before it (and the other way round)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 260, 210, 0.05
    clean = synth.perlin_elevation(rows, cols, seed=3, amplitude=0.05)
    boxes = obstacle_map(synth, rows, cols, 11, 25)
    r = synth.benchmark_radius(4, res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                            fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))
    for first, second in ((boxes, clean), (clean, boxes)):
        a = run(capi, p, rows, cols, 1, res, second, True, twice=first)
        b = run(capi, p, rows, cols, 1, res, second, False)
        for k in ALL:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert not (run(capi, p, rows, cols, 1, res, clean, True, twice=boxes)["traversability_footprint"] == 0).any()


@pytest.mark.parametrize("rows,cols,batch", [(192, 160, 5), (2048, 1024, 1), (1100, 900, 2)])
def test_fused_emit_batches_and_tile_sizes(capi, rows, cols, batch):
    """All three tile heights of the mask kernel (4 / 8 / 32 rows: by the number of tiles) and the batch index of the flag
    grid, bit for bit against the unfused sequence; maps with boxes, so that flagged and unflagged tiles both occur."""
    from traversability_estimation_amd import synth
    res = 0.05
    elevs = np.stack([obstacle_map(synth, rows, cols, 70 + b, 12 + 5 * b, amplitude=0.08) for b in range(batch)])
    elevs[0][cols // 2:cols // 2 + 3, 10:40] = np.nan
    r = synth.benchmark_radius(5, res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                            fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))
    a = run(capi, p, rows, cols, batch, res, elevs, True)
    b = run(capi, p, rows, cols, batch, res, elevs, False)
    for k in ALL:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    fp = a["traversability_footprint"]
    assert (fp == 0).sum() > 20 and (fp > 0).sum() > fp.size // 2


def test_fused_emit_is_skipped_where_it_does_not_apply(capi, oracle):
    """Memo layers, the sequential flag, a tie radius of the second step pass and a roughness check that is switched
    off: same layers either way (the first three keep the old launch sequence, the last is a kernel argument)."""
    from traversability_estimation_amd import synth
    rows, cols, res = 220, 180, 0.05
    elev = obstacle_map(synth, rows, cols, 91, 9)
    r = synth.benchmark_radius(3, res)
    base = dict(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                fp_radius=synth.benchmark_radius(5, res), fp_offset=synth.benchmark_radius(2, res))
    for over, flags in ((dict(), capi.RUN_FOOTPRINT | capi.RUN_FOOTPRINT_MEMO), (dict(), capi.RUN_FOOTPRINT | capi.RUN_SEQUENTIAL),
                        (dict(step_radius2=3 * res), capi.RUN_FOOTPRINT), (dict(fp_check_roughness=0), capi.RUN_FOOTPRINT)):
        kw = dict(base)
        kw.update(over)
        try:
            p = capi.default_params(**kw)
        except (TypeError, AttributeError):
            continue  # (a parameter this binding does not know by that name)
        a = run(capi, p, rows, cols, 1, res, elev, True, flags=flags)
        b = run(capi, p, rows, cols, 1, res, elev, False, flags=flags)
        for k in ALL:
            assert np.array_equal(a[k], b[k], equal_nan=True), (k, over, flags)
