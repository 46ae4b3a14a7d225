#!/usr/bin/env python3
"""Latency of the footprint pass on small / untraversable-heavy maps for one build (TRAVGPU_LIB) and environment:
the reference's bag map (100 x 133 at 0.03 m, default YAML), a 1024^2 map strewn with steps, cfg2, and the bench map."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi, synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import params  # noqa: E402


def med(c, flags, iters=100):
    return float(np.median(c.time_chain_samples(flags, warmup=10, iters=iters)))


def main():
    capi.load()
    out = {"lib": os.path.basename(os.environ.get("TRAVGPU_LIB", "libtravgpu.so")),
           "env": {k: v for k, v in os.environ.items() if k.startswith("TE_")}}
    d = np.load(os.path.join(ROOT, "tests", "golden", "bag_map.npz"))
    rows, cols = int(d["rows"]), int(d["cols"])
    with capi.Context(0) as c:
        c.set_params(capi.default_params())
        c.set_geometry(rows, cols, 1, float(d["resolution"]), tuple(d["position"]))
        c.upload_elevation(d["elevation"])
        out["bag_chain"] = med(c, 0, 200)
        out["bag_chain_fp"] = med(c, capi.RUN_FOOTPRINT, 200)
        if "bagonly" in sys.argv:
            print(json.dumps(out))
            return
        fp = c.download("traversability_footprint")
        out["bag_fp_zero_fraction"] = float((fp == 0).mean())
        out["bag_fp_sum"] = float(np.nansum(fp.astype(np.float64)))
    res = 0.05
    for name, n, boxes in (("steps1024", 1024, 600), ("cfg2", 1024, 0), ("steps4096", 4096, 8000)):
        e = synth.perlin_elevation(n, n, seed=1234)
        if boxes:
            e = synth.with_steps(e, boxes, seed=5)
        with capi.Context(0) as c:
            c.set_params(params(capi, synth, 5.0 if n == 1024 else 9.0, res))
            c.set_geometry(n, n, 1, res)
            c.upload_elevation(e)
            out[name + "_chain_fp"] = med(c, capi.RUN_FOOTPRINT, 50)
            fp = c.download("traversability_footprint")
            out[name + "_fp_zero_fraction"] = float((fp == 0).mean())
            out[name + "_fp_sum"] = float(np.nansum(fp.astype(np.float64)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
