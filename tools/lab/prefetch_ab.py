#!/usr/bin/env python3
"""Is PCIe full duplex for this path?  64 MB layers through pageable buffers: an upload alone, a download alone, and a
download with an upload running beside it (te_prefetch_layers), 10 repetitions each; then the three-plugin sequence
with and without the prefetches (bench.py's host_path block, torch-free)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi, synth  # noqa: E402


def best(f, reps=10):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return round(min(ts), 3), round(float(np.median(ts)), 3)


def main():
    if os.environ.get("PREFETCH_AB_TORCH"):  # (bench.py's process also holds torch's HIP context and thread pools)
        import torch
        torch.cuda.synchronize()
    if os.environ.get("PREFETCH_AB_OMP"):    # ... and has run the OpenMP oracle on 64 threads
        from oracle import oracle as O
        O.build()
        O.set_threads(64)
        g = O.geom(256, 256, 0.05)
        O.chain(g, O.default_params(), synth.perlin_elevation(256, 256, seed=1))
        O.set_threads(1)
    capi.load()
    n, res = 4096, 0.05
    elev = synth.perlin_elevation(n, n, seed=1235)
    other = np.ascontiguousarray(elev[::-1])
    out = {}
    r = synth.benchmark_radius(9, res)
    with capi.Context(0) as ctx:
        ctx.set_params(capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r))
        ctx.set_geometry(n, n, 1, res)
        ctx.upload_elevation(elev)
        ctx.run_chain(capi.RUN_KEEP_NORMALS)
        ctx.sync()
        nrm = [ctx.download(k) for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z")]
        buf = np.empty(n * n, np.float32)
        if not os.environ.get("PREFETCH_AB_SKIP_MICRO"):
            out["upload_ms"] = best(lambda: (ctx.upload_layer("surface_normal_x", nrm[0]), ctx.sync()))
            out["download_ms"] = best(lambda: ctx.download_into("traversability_slope", buf))
            out["prefetch_alone_ms"] = best(lambda: (ctx.prefetch_layers({"surface_normal_x": nrm[0]}), ctx.wait_prefetch()))

            def both():
                ctx.prefetch_layers({"surface_normal_x": nrm[0]})
                ctx.download_into("traversability_slope", buf)
                ctx.wait_prefetch()
            out["download_beside_prefetch_ms"] = best(both)

            def both2():
                ctx.prefetch_layers({"surface_normal_x": nrm[0], "surface_normal_y": nrm[1]})
                ctx.download_into("traversability_slope", buf)
                ctx.wait_prefetch()
            out["download_beside_two_prefetched_layers_ms"] = best(both2)

        def plugins(prefetch):
            ctx.upload_layer("surface_normal_z", nrm[2])
            if prefetch:
                ctx.prefetch_layers({"elevation": elev})
            ctx.run_filter("slope")
            ctx.download("traversability_slope")
            if prefetch:
                ctx.wait_prefetch()
                ctx.prefetch_layers({"surface_normal_x": nrm[0], "surface_normal_y": nrm[1]})
            else:
                ctx.upload_elevation(elev)
            ctx.run_filter("step")
            ctx.download("traversability_step")
            if prefetch:
                ctx.wait_prefetch()
            else:
                ctx.upload_layer("surface_normal_x", nrm[0])
                ctx.upload_layer("surface_normal_y", nrm[1])
            ctx.run_filter("roughness")
            ctx.download("traversability_roughness")
            ctx.sync()
        if os.environ.get("PREFETCH_AB_HOSTPATH"):  # what bench.py does before its three-plugin block
            names = ["traversability_slope", "traversability_step", "traversability_roughness", "traversability"]
            for _ in range(3):
                ctx.upload_elevation(elev)
                ctx.run_chain(0)
                outs = [ctx.download(k) for k in names]
                ctx.sync()
            bufs = [np.empty(elev.size, np.float32) for _ in names]
            for b in [elev] + bufs:
                capi.pin_host(b)
            for _ in range(3):
                ctx.upload_elevation(elev)
                ctx.run_chain(0)
                for k, b in zip(names, bufs):
                    ctx.download_into(k, b)
                ctx.sync()
            for b in [elev] + bufs:
                capi.unpin_host(b)
            del bufs, outs
            ctx.run_chain(capi.RUN_KEEP_NORMALS)
            nrm[:] = [ctx.download(k) for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z")]
        if os.environ.get("PREFETCH_AB_ORDER") == "TF":
            out["three_plugins_prefetch_ms"] = best(lambda: plugins(True), 3)
            out["three_plugins_ms"] = best(lambda: plugins(False), 3)
        else:
            out["three_plugins_ms"] = best(lambda: plugins(False), 3)
            out["three_plugins_prefetch_ms"] = best(lambda: plugins(True), 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
