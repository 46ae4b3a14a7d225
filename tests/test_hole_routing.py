"""The hole routing on the maps the hole benches use, decided by the shim's OWN code: tests/cpu/hole_routing_check.cpp is
compiled from traversability_estimation_amd/csrc/te_hole_routing.h, the header te_shim.hip routes with.  The upload counts
the invalid cells and their RUNS in memory order (k_count_invalid; the definition is restated in count_invalid below and
checked on the device by tests/test_gpu_round6.py); scattered cells (runs of one) take the sparse march up to 2 per mille and
the dense march on long strips above, unobserved regions (runs of eight and more on average) the dense march on strips of
32 rows however few they are."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT
from traversability_estimation_amd import synth


@pytest.fixture(scope="module")
def router(tmp_path_factory):
    exe = tmp_path_factory.mktemp("route") / "hole_routing_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "traversability_estimation_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "hole_routing_check.cpp"), "-o", str(exe)], check=True, timeout=300)

    def route(elev):
        n, runs = count_invalid(elev)
        r = subprocess.run([str(exe), str(elev.size), str(n), str(runs)], capture_output=True, text=True, check=True, timeout=60)
        return r.stdout.strip().replace(", clean attempt skipped", "")
    return route


def count_invalid(elev):
    """k_count_invalid: invalid cells, and invalid cells whose predecessor in memory is valid (or that come first)."""
    flat = ~np.isfinite(np.ascontiguousarray(elev).ravel())
    prev = np.concatenate([[False], flat[:-1]])
    return int(flat.sum()), int((flat & ~prev).sum())


def regions(n, fraction, seed=99):
    e = synth.perlin_elevation(n, n, seed=3)
    rng = np.random.default_rng(seed)
    area, target = 0, fraction * n * n
    while area < target:  # rectangles of 100..400 cells a side, as tools/ab_chain.py --holes 0.5+fraction draws them
        h, w = int(rng.integers(100, 400)), int(rng.integers(100, 400))
        a, b = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
        e[a:a + h, b:b + w] = np.nan
        area += h * w
    return e


def test_routing_of_the_hole_bench_maps(router):
    route = router
    n = 1024
    base = synth.perlin_elevation(n, n, seed=3)
    assert route(base) == "clean"
    assert route(synth.with_holes(base, 0.0003, seed=1)) == "sparse"
    assert route(synth.with_holes(base, 0.001, seed=1)) == "sparse"
    assert route(synth.with_holes(base, 0.003, seed=1)) == "dense"
    assert route(synth.with_holes(base, 0.01, seed=1)) == "dense"
    assert route(synth.with_holes(base, 0.2, seed=1)) == "dense"  # runs of 1.25 cells on average
    assert route(regions(n, 0.05)) == "dense, short strips"
    assert route(regions(n, 0.2)) == "dense, short strips"
    one = base.copy()
    one[300:340, 500:530] = np.nan  # 1.1 per mille of the map, one region: not the sparse march (5x dearer inside a region)
    assert route(one) == "dense, short strips"
    mixed = synth.with_holes(regions(n, 0.05), 0.0005, seed=2)  # regions and a little speckle: still runs of >= 8 on average
    assert route(mixed) == "dense, short strips"


def test_runs_follow_the_memory_order():
    e = np.zeros((6, 8), dtype=np.float32)
    e[2, 3:7] = np.nan   # one run of four
    e[4, 7] = np.nan     # the last cell of a row ...
    e[5, 0] = np.inf     # ... and the first of the next: one run of two in memory order
    e[0, 0] = np.nan     # the layer's first cell starts a run
    assert count_invalid(e) == (7, 3)
