"""Thread safety of the C-ABI (threading row): the reference's callbacks run on AsyncSpinner(0) threads and its
chain call is under no mutex (traversability_estimation_node.cpp:18, TraversabilityMap.cpp:203-214, :764-772), so the
replacement's contexts must be re-entrant across instances and safe to call from several threads at once (ctypes releases
the GIL around every call: these threads really do run inside libtravgpu.so together)."""
import threading

import numpy as np
import pytest

from tests.helpers import OUT_LAYERS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def _case(capi, synth, k):
    rows, cols, res = 200 + 37 * k, 150 + 11 * k, (0.05, 0.03, 0.1, 0.05)[k % 4]
    cells = (1.0, 1.67, 3.0, 5.0)[k % 4]
    r = cells * res if k % 4 == 0 else synth.benchmark_radius(cells, res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=synth.benchmark_radius(1 + k % 3, res), step_radius2=synth.benchmark_radius(1 + k % 2, res),
                            fp_radius=synth.benchmark_radius(3, res), fp_offset=synth.benchmark_radius(1, res))
    elev = synth.with_holes(synth.with_steps(synth.perlin_elevation(rows, cols, seed=50 + k, amplitude=0.15), 6, seed=k), 0.003, seed=k)
    return rows, cols, res, p, elev


def _run(capi, ctx, case, flags):
    rows, cols, res, p, elev = case
    ctx.set_params(p)
    ctx.set_geometry(rows, cols, 1, res)
    ctx.upload_elevation(elev)
    ctx.run_chain(flags)
    return {k: ctx.download(k) for k in list(OUT_LAYERS) + ["traversability_footprint"]}


def test_contexts_are_reentrant_across_threads(capi):
    """Four threads, a context each, different maps / radii / kernel families, 15 rounds: every result bit-equal to the same
    case run alone."""
    from traversability_estimation_amd import synth
    cases = [_case(capi, synth, k) for k in range(4)]
    with capi.Context(0) as c0:
        want = [_run(capi, c0, case, capi.RUN_FOOTPRINT) for case in cases]
    errors = []

    def worker(k):
        try:
            with capi.Context(0) as ctx:
                for _ in range(15):
                    got = _run(capi, ctx, cases[k], capi.RUN_FOOTPRINT)
                    for name in got:
                        if not np.array_equal(got[name].view(np.uint32), want[k][name].view(np.uint32)):
                            errors.append((k, name))
                            return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errors and not any(t.is_alive() for t in ts), errors


def test_one_context_called_from_several_threads(capi):
    """Four threads share ONE context and each runs upload -> chain -> download of the same map over and over while a fifth
    re-sets the same parameters and reads them back: the context's mutex serialises the calls, any interleaving of them leaves
    the same layers, nothing crashes or deadlocks."""
    from traversability_estimation_amd import synth
    case = _case(capi, synth, 1)
    rows, cols, res, p, elev = case
    with capi.Context(0) as ctx:
        want = _run(capi, ctx, case, 0)["traversability"]
        errors, stop = [], threading.Event()

        def worker():
            try:
                for _ in range(25):
                    ctx.upload_elevation(elev)
                    ctx.run_chain(0)
                    got = ctx.download("traversability")
                    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                        errors.append("layer differs")
                        return
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        def fiddler():
            try:
                while not stop.is_set():
                    ctx.set_params(p)  # (the same parameters: the tables are rebuilt, the results are not)
                    q = ctx.get_params()
                    assert capi.params_to_bytes(q) == capi.params_to_bytes(p)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=worker) for _ in range(4)]
        f = threading.Thread(target=fiddler)
        f.start()
        [t.start() for t in ts]
        [t.join(600) for t in ts]
        stop.set()
        f.join(60)
        assert not errors and not any(t.is_alive() for t in ts) and not f.is_alive(), errors[:3]
