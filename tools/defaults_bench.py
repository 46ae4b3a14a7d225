#!/usr/bin/env python3
"""The reference's DEFAULT parameters (robot_filter_parameter.yaml: normals / roughness radius 0.05 m, step windows 0.04 m) on
maps at the resolutions elevation_mapping is commonly run at: at res 0.05 m the 0.05 m radius is a TIE radius of exactly
one cell (the four edge neighbours are on the circle: kept or dropped per centre by the rounding of the positions), the
step windows hold the centre alone; at res 0.03 m (the bag) they are 9- and 5-point discs.  Event-timed launches, chain and
chain + footprint, fast paths and (--generic) the generic kernels.  Prints one JSON object.  Needs an MI355X."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi, synth  # noqa: E402


def one(res, n):
    capi.load()
    e = synth.perlin_elevation(n, n, seed=1234)
    with capi.Context(0) as c:
        c.set_params(capi.default_params())
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(e)
        row = {}
        seq = capi.RUN_SEQUENTIAL if "sequential" in sys.argv else 0  # (one stream: every kernel's own time under rocprofv3)
        for name, flags in (("chain", seq), ("chain+footprint", capi.RUN_FOOTPRINT | seq), ("chain generic", capi.RUN_GENERIC_KERNELS),
                            ("normals only", capi.RUN_NORMALS_ONLY), ("normals only generic", capi.RUN_NORMALS_ONLY | capi.RUN_GENERIC_KERNELS)):
            if "profile" in sys.argv and "generic" in name:
                continue
            s = c.time_chain_samples(flags, warmup=5, iters=10 if "profile" in sys.argv else 50)
            row[name] = {"ms": round(float(np.median(s)), 4), "cells_per_s": round(n * n / (float(np.median(s)) * 1e-3))}
    return row


def main():
    """Every (resolution, size) in a process of its own: in one process the 4096^2 case measured 0.29 ms where a fresh process
    measures 0.21 -- what the contexts before it left behind in the allocator decides (round 6)."""
    if "--one" in sys.argv:
        k = sys.argv.index("--one")
        print(json.dumps(one(float(sys.argv[k + 1]), int(sys.argv[k + 2]))))
        return
    import subprocess
    sizes = [int(v) for v in os.environ.get("TE_SIZES", "256,1024,4096").split(",")]
    extra = [a for a in sys.argv[1:] if a in ("profile", "sequential")]
    out = {}
    for res in (0.05, 0.03):
        for n in sizes:
            if "profile" in extra:  # (under rocprofv3: the children would not be traced)
                out[f"res {res} {n}x{n}"] = one(res, n)
                continue
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(res), str(n)] + extra, capture_output=True, text=True, check=True)
            out[f"res {res} {n}x{n}"] = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
