#!/usr/bin/env python3
"""Why are downloads into fresh arrays 3 x slower in a process that has prefetched once?  Times, before and after the
first te_prefetch_layers: a pure numpy first touch of a fresh 64 MB array (one thread), te_download_layer into a fresh
array and into a preallocated one."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi, synth
capi.load()
n = 4096
elev = synth.perlin_elevation(n, n, seed=1235)
def t(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    return ts
def touch():
    a = np.empty(n * n, np.float32); a[::1024] = 1.0  # one write per page
def thp():
    try: return open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
    except OSError: return "?"
out = {"thp": thp(), "threads_before": len(os.listdir("/proc/self/task"))}
with capi.Context(0) as ctx:
    r = synth.benchmark_radius(9, 0.05)
    ctx.set_params(capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r))
    ctx.set_geometry(n, n, 1, 0.05)
    ctx.upload_elevation(elev)
    ctx.run_chain(0); ctx.sync()
    buf = np.empty(n * n, np.float32)
    def probe(tag):
        out[tag] = {"numpy_first_touch_ms": t(touch), "download_fresh_ms": t(lambda: ctx.download("traversability_slope")),
                    "download_prealloc_ms": t(lambda: ctx.download_into("traversability_slope", buf)), "threads": len(os.listdir("/proc/self/task"))}
    probe("before")
    ctx.prefetch_layers({"surface_normal_x": elev}); ctx.wait_prefetch()
    probe("after_one_prefetch")
    for _ in range(3):
        ctx.prefetch_layers({"surface_normal_x": elev}); ctx.download("traversability_slope"); ctx.wait_prefetch()
    probe("after_prefetches_beside_downloads")
    time.sleep(0.5)
    probe("after_half_a_second_idle")
print(json.dumps(out))
