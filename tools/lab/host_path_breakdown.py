#!/usr/bin/env python3
"""Where the time of the plugin-shaped host path goes (bench.py's host_path / three_plugins): every step of the
three-plugin sequence timed on its own (te_sync after each), pageable buffers, fresh and reused destinations.
Usage (GPU box): python tools/lab/host_path_breakdown.py [n]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from traversability_estimation_amd import capi
from traversability_estimation_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res = 0.05
r = synth.benchmark_radius(9.0, res)
params = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                             fp_radius=synth.benchmark_radius(6.0, res), fp_offset=synth.benchmark_radius(3.0, res))
elev = synth.perlin_elevation(n, n, seed=1235)
capi.load()
ctx = capi.Context(0)
ctx.set_params(params)
ctx.set_geometry(n, n, 1, res)
ctx.upload_elevation(elev)
ctx.run_chain(capi.RUN_KEEP_NORMALS)
nrm = [ctx.download(k) for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z")]
ctx.sync()
out = {}
def timed(name, f, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ctx.sync(); d = time.perf_counter() - t0
        best = d if best is None or d < best else best
    out[name] = round(best * 1e3, 3)
    return r
reuse = np.empty(n * n, np.float32); reuse[:] = 0
timed("upload nz (pageable, 64 MB)", lambda: ctx.upload_layer("surface_normal_z", nrm[2]))
timed("run_filter slope", lambda: ctx.run_filter("slope"))
timed("download slope into a FRESH array", lambda: ctx.download("traversability_slope"))
timed("download slope into a touched array", lambda: ctx.download_into("traversability_slope", reuse))
timed("upload elevation", lambda: ctx.upload_elevation(elev))
timed("run_filter step", lambda: ctx.run_filter("step"))
timed("download step fresh", lambda: ctx.download("traversability_step"))
timed("upload nx + ny", lambda: (ctx.upload_layer("surface_normal_x", nrm[0]), ctx.upload_layer("surface_normal_y", nrm[1])))
timed("run_filter roughness", lambda: ctx.run_filter("roughness"))
timed("download roughness fresh", lambda: ctx.download("traversability_roughness"))
timed("np.empty + touch 64 MB (page faults alone)", lambda: np.empty(n * n, np.float32).fill(0))
timed("memcpy 64 MB numpy (one thread)", lambda: np.copyto(reuse, nrm[0]))
print(json.dumps(out, indent=1))
