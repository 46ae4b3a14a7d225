import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi, synth
from oracle import oracle
from tests.helpers import OUT_LAYERS, compare_layer
capi.load(); oracle.build()
ALL = OUT_LAYERS + ("traversability_footprint",)
n, res, tile = int(os.environ.get("N", "2048")), float(os.environ.get("RES", "0.0625")), 256
elev = np.tile(synth.perlin_elevation(1024, 1024, seed=77).reshape(1024, 1024), (n // 1024, n // 1024))
elev = (elev + np.linspace(0.0, 1.5, n, dtype=np.float32)[None, :]).astype(np.float32)
r = synth.benchmark_radius(5, res)
p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r, fp_radius=synth.benchmark_radius(6, res), fp_offset=synth.benchmark_radius(3, res))
op = oracle.default_params()
for f, _ in op._fields_: setattr(op, f, getattr(p, f))
reach = 2 * 5 + 3 + 9 + 2
margin = reach + 2 * 5 + 12
oracle.set_threads(8)
with capi.Context(0) as ctx:
    ctx.set_params(p); ctx.set_geometry(n, n, 1, res); ctx.upload_elevation(elev); ctx.run_chain(capi.RUN_FOOTPRINT)
    for tick, (r0, c0) in enumerate(((300, 500), (n - tile, 0), (700, n - tile - 3), (0, 0), (n - tile, n - tile))):
        patch = (synth.perlin_elevation(tile, tile, seed=3000 + tick).reshape(tile, tile) * np.float32(0.6)).astype(np.float32)
        elev[c0:c0 + tile, r0:r0 + tile] = patch
        ctx.upload_tile(np.ascontiguousarray(patch), 0, r0, c0)
        ctx.run_chain_region(0, r0, c0, tile, tile, flags=capi.RUN_FOOTPRINT)
        ctx.sync()
        i_lo, i_hi = max(0, r0 - reach - margin), min(n, r0 + tile + reach + margin)
        j_lo, j_hi = max(0, c0 - reach - margin), min(n, c0 + tile + reach + margin)
        crop = np.ascontiguousarray(elev[j_lo:j_hi, i_lo:i_hi])
        g = oracle.geom(i_hi - i_lo, j_hi - j_lo, res)
        want = oracle.chain(g, op, crop)
        want["traversability_footprint"] = oracle.footprint(g, op, crop, want)
        ki = slice(0 if i_lo == 0 else margin, (i_hi - i_lo) if i_hi == n else (i_hi - i_lo) - margin)
        kj = slice(0 if j_lo == 0 else margin, (j_hi - j_lo) if j_hi == n else (j_hi - j_lo) - margin)
        for k in ALL:
            a = ctx.download_tile(k, 0, i_lo, j_lo, i_hi - i_lo, j_hi - j_lo)
            b = want[k].reshape(j_hi - j_lo, i_hi - i_lo)
            bad = (np.abs(a.astype(np.float64) - b) > 1e-5) | (np.isnan(a) != np.isnan(b))
            badk = np.zeros_like(bad); badk[kj, ki] = bad[kj, ki]
            if badk.any():
                jj, ii = np.nonzero(badk)
                print("tick", tick, (r0, c0), k, "bad", badk.sum(), "cells (i,j) abs:", [(int(i + i_lo), int(j + j_lo), float(a[j, i]), float(b[j, i])) for j, i in list(zip(jj, ii))[:14]])
        # whole-map footprint again: does it fix things?
        ctx.run_footprint(); ctx.sync()
        a = ctx.download_tile("traversability_footprint", 0, i_lo, j_lo, i_hi - i_lo, j_hi - j_lo)
        b = want["traversability_footprint"].reshape(j_hi - j_lo, i_hi - i_lo)
        print("tick", tick, "after whole-map footprint: bad", int(((np.abs(a.astype(np.float64) - b) > 1e-5)[kj, ki]).sum()))
print("done")
