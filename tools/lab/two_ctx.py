#!/usr/bin/env python3
"""Upper bound for cross-launch pipelining: N independent contexts (own layers, own streams) on one device, each
holding the bench map, launched round-robin; host-timed loops of K launches + sync of all contexts.  With N = 1 this
is bench.py's timed region; with N = 2 the footprint phase of one launch can run under the chain phase of the next
without any double-buffering inside a context.

    python tools/lab/two_ctx.py --nctx 2 --loops 20,100
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--nctx", type=int, default=2)
    ap.add_argument("--loops", type=str, default="20,100")
    ap.add_argument("--no-footprint", action="store_true")
    ap.add_argument("--tag", type=str, default="")
    a = ap.parse_args()
    from traversability_estimation_amd import capi, synth
    capi.load()
    n, res = a.size, 0.05
    r = synth.benchmark_radius(9.0, res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                            fp_radius=synth.benchmark_radius(6.0, res), fp_offset=synth.benchmark_radius(3.0, res))
    elev = synth.perlin_elevation(n, n, seed=1235)
    flags = 0 if a.no_footprint else capi.RUN_FOOTPRINT
    ctxs = [capi.Context(0) for _ in range(a.nctx)]
    out = {"tag": a.tag, "nctx": a.nctx, "size": n, "flags": flags, "env": {k: v for k, v in os.environ.items() if k.startswith("TE_")}}
    for c in ctxs:
        c.set_params(p)
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(elev)
    for _ in range(10):
        for c in ctxs:
            c.run_chain(flags)
    for c in ctxs:
        c.sync()
    single = ctxs[0].time_chain_samples(flags, warmup=10, iters=60)
    out["single_launch_event_ms_median"] = float(np.median(single))
    loops = {}
    for K in [int(k) for k in a.loops.split(",")]:
        best = None
        for _ in range(5):
            for c in ctxs:
                c.sync()
            t0 = time.perf_counter()
            for k in range(K):
                ctxs[k % a.nctx].run_chain(flags)
            for c in ctxs:
                c.sync()
            d = (time.perf_counter() - t0) * 1e3
            best = d if best is None or d < best else best
        loops[str(K)] = {"ms_total_best_of_5": best, "ms_per_step": best / K}
    out["host_loops"] = loops
    for c in ctxs:
        c.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
