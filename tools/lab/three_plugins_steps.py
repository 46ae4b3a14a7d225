import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if "torch" in mode:
    import torch
    torch.cuda.synchronize()
from traversability_estimation_amd import capi, synth
capi.load()
n, res = 4096, 0.05
elev = synth.perlin_elevation(n, n, seed=1235)
stack = np.stack([elev])
r = synth.benchmark_radius(9, res)
p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=synth.benchmark_radius(6.0, res), fp_offset=synth.benchmark_radius(3.0, res))
with capi.Context(0) as ctx:
    ctx.set_params(p)
    ctx.set_geometry(n, n, 1, res)
    ctx.upload_elevation(stack)
    if "bench" in mode:  # what bench.py does first
        flags = capi.RUN_FOOTPRINT
        ctx.time_chain_samples(flags, warmup=20, iters=100)
        ctx.time_chain_samples(capi.RUN_NORMALS_ONLY, warmup=5, iters=100)
        for _ in range(26):
            ctx.run_chain(flags)
        ctx.sync()
    ctx.run_chain(capi.RUN_KEEP_NORMALS)
    nrm = [ctx.download(k) for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z")]
    for prefetch in (False, True, False, True):
        rows = []
        for _ in range(3):
            T = [time.perf_counter()]
            def tick(): T.append(time.perf_counter())
            ctx.upload_layer("surface_normal_z", nrm[2]); tick()
            if prefetch: ctx.prefetch_layers({"elevation": stack}); tick()
            ctx.run_filter("slope"); tick()
            o1 = ctx.download("traversability_slope"); tick()
            if prefetch: ctx.wait_prefetch(); tick()
            if not prefetch: ctx.upload_elevation(stack); tick()
            else: ctx.prefetch_layers({"surface_normal_x": nrm[0], "surface_normal_y": nrm[1]}); tick()
            ctx.run_filter("step"); tick()
            o2 = ctx.download("traversability_step"); tick()
            if prefetch: ctx.wait_prefetch(); tick()
            if not prefetch:
                ctx.upload_layer("surface_normal_x", nrm[0]); tick()
                ctx.upload_layer("surface_normal_y", nrm[1]); tick()
            ctx.run_filter("roughness"); tick()
            o3 = ctx.download("traversability_roughness"); tick()
            ctx.sync(); tick()
            rows.append([round((b - a) * 1e3, 2) for a, b in zip(T, T[1:])] + [round((T[-1] - T[0]) * 1e3, 2)])
        print(mode, "prefetch" if prefetch else "plain", min(rows, key=lambda r: r[-1]))
