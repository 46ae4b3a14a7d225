#!/usr/bin/env python3
"""hipGraph replay against direct launches by map size (TE_OPT_GRAPH_REPLAY 1 / 2): where should the size switch sit?
Round 1 measured the replay 12 us SLOWER at 1024^2 / 2048^2 (ROCm 7.0, six kernels on two streams); the chain has other
kernels now and small maps run on one stream.  Event-timed launches, median of 200.  Needs an MI355X."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi, synth  # noqa: E402


def main():
    capi.load()
    out = {}
    d = np.load(os.path.join(ROOT, "tests", "golden", "bag_map.npz"))
    cases = [("bag 100x133 default YAML", int(d["rows"]), int(d["cols"]), float(d["resolution"]), None, d["elevation"])]
    for n, cells in ((256, 5.0), (512, 5.0), (1024, 5.0), (2048, 5.0), (2048, 9.0)):
        cases.append((f"{n}x{n} R{cells:g}", n, n, 0.05, cells, synth.perlin_elevation(n, n, seed=1234)))
    for name, rows, cols, res, cells, elev in cases:
        row = {}
        for mode, label in ((2, "direct"), (1, "graph")):
            with capi.Context(0) as c:
                if cells is None:
                    c.set_params(capi.default_params())
                else:
                    r = synth.benchmark_radius(cells, res)
                    c.set_params(capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                                                     fp_radius=synth.benchmark_radius(6.0, res), fp_offset=synth.benchmark_radius(3.0, res)))
                c.set_geometry(rows, cols, 1, res)
                c.set_option(capi.OPT_GRAPH_REPLAY, mode)
                c.upload_elevation(elev)
                for flags, fl in ((0, "chain"), (capi.RUN_FOOTPRINT, "chain+footprint")):
                    s = c.time_chain_samples(flags, warmup=20, iters=200)
                    row[f"{fl} {label}"] = round(float(np.median(s)) * 1e3, 1)
        out[name] = row
    print(json.dumps({"us_per_launch_median": out}, indent=1))


if __name__ == "__main__":
    main()
