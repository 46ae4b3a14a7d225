#!/usr/bin/env python3
"""Quick A/B timing of one chain launch on the bench workload, without torch (a fresh GPU box pays 1-2 minutes for the
first `import torch`; this starts in seconds).  The kernels' measurement switches are environment variables read once
per process (TE_NO_F4, TE_NO_N3, TE_F4_BLOCKS_PER_CU, ...), so a variant is one process:

    TE_NO_F4=1 python tools/ab_chain.py --tag no_f4

Prints one JSON line: median / p10 / p90 of the event-timed launch (te_time_chain_samples) and, with --loops, the
host-timed loop of K launches + te_sync for several K (what bench.py's timed region does): its slope is the per-step
time, its intercept the fixed cost of the first launch and the final synchronisation.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--radius-cells", type=float, default=9.0)
    ap.add_argument("--res", type=float, default=0.05)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--sequential", action="store_true")
    ap.add_argument("--no-footprint", action="store_true")
    ap.add_argument("--normals-only", action="store_true")
    ap.add_argument("--footprint-only", action="store_true", help="time te_run_footprint alone (mask + slide) after one chain")
    ap.add_argument("--holes", type=float, default=0.0)
    ap.add_argument("--boxes", type=int, default=0, help="raised / lowered rectangles of 4..40 cells a side (kerbs, crates): untraversable cells")
    ap.add_argument("--exact-cells", action="store_true", help="footprint radius and offset exactly 6 and 3 cells (a tie radius: cells on the circle decided per centre)")
    ap.add_argument("--exact-chain", action="store_true", help="normals / roughness / step radii exactly --radius-cells cells (tie radii: the generic kernels)")
    ap.add_argument("--loops", type=str, default="", help="comma-separated K: host-timed loops of K launches + sync")
    ap.add_argument("--fb-walk", type=int, default=0, help="te_set_option(TE_OPT_FP_BLOCKED_WALK): 1 one disc per wavefront, 2 one disc per lane (0: by the length of the list)")
    ap.add_argument("--tag", type=str, default="")
    ap.add_argument("--check", action="store_true", help="after the timing, compare what the launches left on the device with the "
                    "oracle (bench.py's parity_check: a corner crop and a full-width band; exit code 1 on a mismatch) -- no number "
                    "goes into profiles/ from an unchecked run")
    ap.add_argument("--check-whole", action="store_true", help="--check on EVERY cell of the map (the OpenMP oracle on the whole map, at the map's own "
                    "resolution: no crop, so checkForStep's position-rounding ties are the map's own)")
    a = ap.parse_args()
    a.check = a.check or a.check_whole
    from traversability_estimation_amd import capi, synth
    capi.load()
    n, B = a.size, a.batch
    r = synth.benchmark_radius(a.radius_cells, a.res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                            fp_radius=synth.benchmark_radius(6.0, a.res), fp_offset=synth.benchmark_radius(3.0, a.res))
    if a.exact_cells:
        p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=6.0 * a.res, fp_offset=3.0 * a.res)
    if a.exact_chain:
        rr = a.radius_cells * a.res
        p = capi.default_params(normals_radius=rr, rough_radius=rr, step_radius1=rr, step_radius2=rr, fp_radius=p.fp_radius, fp_offset=p.fp_offset)
    elevs = [synth.perlin_elevation(n, n, seed=1235 + b) for b in range(B)]
    if 0 < a.holes < 0.5:
        elevs = [synth.with_holes(e, a.holes, seed=99 + b) for b, e in enumerate(elevs)]
    elif a.holes >= 0.5:  # rectangles of 100..400 cells a side until (holes - 0.5) of the area is covered (as bench.py --holes)
        rng = np.random.default_rng(99)
        for b in range(B):
            area, target = 0, (a.holes - 0.5) * n * n
            while area < target:
                h, w = (int(v) for v in rng.integers(100, 400, size=2))
                r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
                elevs[b][c0:c0 + w, r0:r0 + h] = np.nan
                area += h * w
    if a.boxes > 0:
        rng = np.random.default_rng(7)
        for b in range(B):
            for _ in range(a.boxes):
                h, w = (int(v) for v in rng.integers(4, 40, size=2))
                r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
                elevs[b][c0:c0 + w, r0:r0 + h] += np.float32(rng.uniform(0.15, 0.5) * rng.choice([-1.0, 1.0]))
    flags = 0 if a.no_footprint else capi.RUN_FOOTPRINT
    if a.sequential:
        flags |= capi.RUN_SEQUENTIAL
    if a.normals_only:
        flags = capi.RUN_NORMALS_ONLY
    out = {"tag": a.tag, "size": n, "batch": B, "radius_cells": a.radius_cells, "flags": flags, "holes": a.holes, "boxes": a.boxes,
           "env": {k: v for k, v in os.environ.items() if k.startswith("TE_")}}
    with capi.Context(0) as ctx:
        ctx.set_params(p)
        if a.fb_walk:
            ctx.set_option(capi.OPT_FP_BLOCKED_WALK, a.fb_walk)
        ctx.set_geometry(n, n, B, a.res)
        ctx.upload_elevation(np.stack(elevs))
        if a.footprint_only:
            ctx.run_chain(flags)
            ctx.sync()
            ts = []
            for _ in range(a.iters + 10):
                t0 = time.perf_counter()
                ctx.run_footprint()
                ctx.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            s = np.array(ts[10:])
            out["what"] = "te_run_footprint + te_sync, host-timed"
        else:
            s = ctx.time_chain_samples(flags, warmup=20, iters=a.iters)
        out.update(ms_median=float(np.median(s)), ms_p10=float(np.percentile(s, 10)), ms_p90=float(np.percentile(s, 90)),
                   cells_per_s=B * n * n / (float(np.median(s)) * 1e-3))
        if a.loops:
            loops = {}
            for K in [int(k) for k in a.loops.split(",")]:
                best = None
                for _ in range(5):
                    ctx.sync()
                    t0 = time.perf_counter()
                    for _ in range(K):
                        ctx.run_chain(flags)
                    ctx.sync()
                    d = (time.perf_counter() - t0) * 1e3
                    best = d if best is None or d < best else best
                loops[str(K)] = {"ms_total_best_of_5": best, "ms_per_step": best / K}
            out["host_loops"] = loops
        if a.check and not a.normals_only:
            import types
            import bench
            ctx.run_chain(flags)
            ctx.sync()
            args = types.SimpleNamespace(radius_cells=a.radius_cells, res=a.res)
            rep = bench.parity_check(args, ctx, elevs[0], p, not a.no_footprint, n, whole=a.check_whole)
            out["parity_check"] = {"ok": rep["ok"], "windows": rep["windows"],
                                   "mismatches": {k: v["mismatches"] for k, v in rep["layers"].items()},
                                   "max_abs_err": max(v["max_abs_err"] for v in rep["layers"].values()),
                                   "cells_per_layer": next(iter(rep["layers"].values()))["cells"]}
    print(json.dumps(out), flush=True)
    if a.check and not out.get("parity_check", {"ok": True})["ok"]:
        sys.exit(1)


if __name__ == "__main__":
    main()
