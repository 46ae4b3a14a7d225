#!/usr/bin/env python3
"""Measures the BASELINE.json configurations that are not the bench.py headline (which is configs[2]):

  cfg2  one 1024 x 1024 map, radius 5 cells                       (launch-latency regime)
  cfg4  a batch of 512 x 512 maps, radius 5 cells, one launch     (the batch axis, what multi-GPU shards)
  cfg5  8192 x 8192 resident map, 256 x 256 dirty tiles per tick  (te_upload_tile + te_run_chain_region)
  N2    batched circular checkFootprintPath, N3 polygon footprint layers

Prints one JSON object; the committed copies are profiles/rNN_configs.json.  Needs an MI355X.
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def params(capi, synth, cells, res):
    r = synth.benchmark_radius(cells, res)
    return capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(6.0, res), fp_offset=synth.benchmark_radius(3.0, res))


def run_bag(capi, synth, res, out):
    # ---- the reference's own map: 100 x 133 at 0.03 m, default YAML (latency of one node update) ---------------------
    d = np.load(os.path.join(ROOT, "tests", "golden", "bag_map.npz"))
    rows, cols = int(d["rows"]), int(d["cols"])
    with capi.Context(0) as c:
        c.set_params(capi.default_params())
        c.set_geometry(rows, cols, 1, float(d["resolution"]), tuple(d["position"]))
        c.upload_elevation(d["elevation"])
        for flags, name in ((0, "chain"), (capi.RUN_FOOTPRINT, "chain+footprint")):
            s = c.time_chain_samples(flags, warmup=20, iters=200)
            out[f"cfg1 bag map {rows}x{cols} default YAML {name}"] = {"ms_per_launch_median": float(np.median(s)), "ms_p90": float(np.percentile(s, 90)),
                                                                  "cells_per_s": rows * cols / (float(np.median(s)) * 1e-3)}
        t0 = time.perf_counter()
        for _ in range(200):
            c.upload_elevation(d["elevation"])
            c.run_chain(0)
            outs = [c.download(k) for k in ("traversability_slope", "traversability_step", "traversability_roughness", "traversability")]
        c.sync()
        out[f"cfg1 bag map {rows}x{cols} default YAML host in / host out"] = {"ms_per_update": (time.perf_counter() - t0) / 200 * 1e3,
                                                                          "what": "upload elevation + chain + download 4 layers, pageable buffers, mean of 200"}
        del outs


def run_cfg2(capi, synth, res, out):
    # ---- cfg2 -----------------------------------------------------------------------------------------------
    n = 1024
    with capi.Context(0) as c:
        c.set_params(params(capi, synth, 5.0, res))
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(synth.perlin_elevation(n, n, seed=1234))
        for flags, name in ((0, "chain"), (capi.RUN_FOOTPRINT, "chain+footprint")):
            ms = c.time_chain(flags, warmup=5, iters=100)
            out[f"cfg2 1024x1024 R5 {name}"] = {"ms_per_launch": ms, "cells_per_s": n * n / (ms * 1e-3)}


def run_cfg4(capi, synth, res, out):
    # ---- cfg4 -----------------------------------------------------------------------------------------------
    n, B = 512, int(os.environ.get("TE_CFG4_MAPS", "512"))
    base = synth.perlin_elevation(n, n, seed=2000)
    rng = np.random.default_rng(2000)
    maps = (base[None, :, :] + rng.normal(0.0, 0.01, size=(B, n, n)).astype(np.float32)).astype(np.float32)
    with capi.Context(0) as c:
        c.set_params(params(capi, synth, 5.0, res))
        c.set_geometry(n, n, B, res)
        c.upload_elevation(maps)
        for flags, name in ((0, "chain"), (capi.RUN_FOOTPRINT, "chain+footprint")):
            ms = c.time_chain(flags, warmup=2, iters=10)
            out[f"cfg4 {B} x 512x512 R5 {name}"] = {"ms_per_launch": ms, "cells_per_s": B * n * n / (ms * 1e-3)}
    del maps


def run_cfg5(capi, synth, res, out):
    # ---- cfg5: 8192^2 resident map, one dirty tile per tick (256^2 as BASELINE words it, and 1024^2) ---------------
    # (a) synchronous ticks: te_upload_tile + te_run_chain_region [+ footprint] + te_sync, host-timed latency per tick;
    # (b) the same plus the D2H of the tile's traversability [footprint] (te_download_tile): what a consumer sees;
    # (c) streaming: te_upload_tile_async / te_run_chain_region / te_download_tile_async on page-locked buffers, several
    #     ticks in flight (H2D of tick k+1 and D2H of tick k-1 on the copy streams beside the kernels of tick k).
    # The new content of a tile is the old one plus a smooth bump that vanishes at the tile's edge (a sensor update does
    # not tear the map; a torn edge is a cliff, i.e. untraversable cells and spiral walks in the footprint pass).
    n = 8192
    # (the generator needs a minute for 8192^2: a 2048^2 map mirrored into a seamless 4096^2 one, repeated 2 x 2 -- a plain
    # repetition has cliffs at its seams, i.e. lines of untraversable cells and spiral walks: 4.7 ms instead of 1.2)
    a = synth.perlin_elevation(2048, 2048, seed=77).reshape(2048, 2048)
    m = np.block([[a, a[:, ::-1]], [a[::-1, :], a[::-1, ::-1]]])
    elev = np.tile(m, (2, 2)).astype(np.float32)
    rng = np.random.default_rng(77)
    n_ticks = 64
    res5 = {}
    with capi.Context(0) as c:
        c.set_params(params(capi, synth, 5.0, res))
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(elev)
        ms_full = float(np.median(c.time_chain_samples(0, warmup=5, iters=20)))
        ms_full_fp = float(np.median(c.time_chain_samples(capi.RUN_FOOTPRINT, warmup=5, iters=20)))  # (the first launches capture the graph)
        res5.update({"full_map_chain_ms": ms_full, "full_map_cells_per_s": n * n / (ms_full * 1e-3),
                     "full_map_chain_footprint_ms": ms_full_fp, "full_map_chain_footprint_cells_per_s": n * n / (ms_full_fp * 1e-3)})
        for tile in (256, 1024):
            w1 = np.hanning(tile).astype(np.float32)
            bumps = [(np.outer(w1, w1) * (0.2 * synth.perlin_elevation(tile, tile, seed=1000 + k).reshape(tile, tile))).astype(np.float32) for k in range(4)]
            origins = [tuple(int(v) for v in rng.integers(0, n - tile, size=2)) for _ in range(n_ticks)]

            def new_tile(k):
                r0, c0 = origins[k]
                t = np.ascontiguousarray(elev[c0:c0 + tile, r0:r0 + tile] + bumps[k % 4])
                elev[c0:c0 + tile, r0:r0 + tile] = t
                return t

            tag = f"tile {tile}x{tile}: "
            for flags, name, layer in ((0, "chain", "traversability"), (capi.RUN_FOOTPRINT, "chain+footprint", "traversability_footprint")):
                c.run_chain(flags)  # (a region run with the footprint flag refreshes a COMPLETE footprint layer)
                for download in (False, True):
                    lat = []
                    for k, (r0, c0) in enumerate(origins):
                        t = new_tile(k)
                        t0 = time.perf_counter()
                        c.upload_tile(t, 0, r0, c0)
                        c.run_chain_region(0, r0, c0, tile, tile, flags=flags)
                        if download:
                            c.download_tile(layer, 0, r0, c0, tile, tile)
                        else:
                            c.sync()
                        lat.append((time.perf_counter() - t0) * 1e3)
                    lat = np.array(lat[8:])
                    res5[tag + f"sync ticks, {name}{', tile downloaded' if download else ''}"] = {
                        "tick_ms_median": float(np.median(lat)), "tick_ms_p95": float(np.percentile(lat, 95)), "ticks_per_s": 1e3 / float(np.median(lat))}
                # streaming with the copy streams: page-locked in / out buffers, `depth` ticks in flight
                # (the producer's tiles ARE page-locked buffers, one per tick, filled outside the timed loop like the sync ticks')
                nbuf = 8
                bin_ = [new_tile(k) for k in range(n_ticks)]
                bout = [np.empty((tile, tile), np.float32) for _ in range(nbuf)]
                for b in bin_ + bout:
                    capi.pin_host(b)
                try:
                    for depth in (1, 4):
                        c.sync()
                        t0 = time.perf_counter()
                        for k, (r0, c0) in enumerate(origins):
                            c.upload_tile_async(bin_[k], 0, r0, c0)
                            c.run_chain_region(0, r0, c0, tile, tile, flags=flags)
                            c.download_tile_async(layer, 0, r0, c0, bout[k % nbuf])
                            if k % depth == depth - 1:
                                c.sync()
                        c.sync()
                        dt = (time.perf_counter() - t0) * 1e3 / n_ticks
                        res5[tag + f"streaming ticks (copy streams, {depth} in flight), {name}, tile uploaded and downloaded"] = {
                            "ms_per_tick": dt, "ticks_per_s": 1e3 / dt, "dirty_cells_per_s": tile * tile * 1e3 / dt}
                finally:
                    for b in bin_ + bout:
                        capi.unpin_host(b)
        res5["what"] = ("host-timed; 'sync ticks' = te_upload_tile + te_run_chain_region [+ te_download_tile] + wait, one tick at a time (latency); "
                        "'streaming' = the asynchronous pair on the copy streams (H2D of the next tile and D2H of the previous result beside "
                        "the kernels), page-locked producer and consumer buffers (throughput); the BASELINE "
                        "rate to sustain is 20 ticks/s")
        out["cfg5 8192x8192 R5 resident, one dirty tile per tick"] = res5


def run_n2(capi, synth, res, out):
    # ---- N2: batched circular checkFootprintPath on the resident footprint layer -----------------------------
    from oracle import oracle as O
    n = 4096
    rng = np.random.default_rng(5)
    with capi.Context(0) as c:
        p = params(capi, synth, 9.0, res)
        c.set_params(p)
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(synth.perlin_elevation(n, n, seed=1235))
        c.run_chain(capi.RUN_FOOTPRINT)
        c.sync()
        fp = c.download("traversability_footprint")
        half = 0.5 * n * res
        paths = []
        for _ in range(200000):  # MPC-style candidates: 2..6 poses, segments up to 2 m
            k = int(rng.integers(2, 7))
            start = rng.uniform(-half + 3.0, half - 3.0, size=2)
            steps = rng.uniform(-2.0, 2.0, size=(k - 1, 2))
            paths.append(np.clip(np.vstack([start, start + np.cumsum(steps, axis=0)]), -half + 0.01, half - 0.01))
        off, xy = capi.pack_paths(paths)
        c.check_footprint_paths(paths[:1000])  # warm-up
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            safe, trav, st = c.check_footprint_paths_packed(off, xy)
            d = time.perf_counter() - t0
            best = d if best is None or d < best else best
        dt = best
        g = O.geom(n, n, res)
        m = 20000
        t0 = time.perf_counter()
        ws, wt, wst = O.check_circular_paths(g, fp, 0.3, paths[:m])
        dt_cpu = time.perf_counter() - t0
        assert np.array_equal(ws, safe[:m]) and np.array_equal(wt, trav[:m]) and np.array_equal(wst, st[:m])
        out["N2 checkFootprintPath (circular), 200000 paths of 2-6 poses on 4096x4096"] = {
            "gpu_paths_per_s": len(paths) / dt, "gpu_ms": dt * 1e3, "safe_fraction": float(safe.mean()),
            "cpu_oracle_paths_per_s": m / dt_cpu,
            "what": "te_check_footprint_paths on packed host arrays, H2D of the poses and D2H of the results included; the oracle "
                    "(1 thread, same layer) on the first 20000 paths, results bit-identical"}

def run_n2p(capi, synth, res, out):
    # ---- N2, polygonal footprints: checkPolygonalFootprintPath for a batch of candidate paths -------------------
    from oracle import oracle as O
    points = [[0.45, 0.30, 0.0], [0.45, -0.30, 0.0], [-0.45, -0.30, 0.0], [-0.45, 0.30, 0.0]]
    n = 4096
    rng = np.random.default_rng(6)
    elev = synth.with_steps(synth.perlin_elevation(n, n, seed=1235), 400, seed=9)
    with capi.Context(0) as c:
        p = params(capi, synth, 9.0, res)
        c.set_params(p)
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(elev)
        c.run_chain(capi.RUN_FOOTPRINT)
        c.sync()
        layers = {k: c.download(k) for k in ("traversability_slope", "traversability_step", "traversability_roughness",
                                             "traversability")}
        half = 0.5 * n * res
        paths = []
        for _ in range(100000):  # 2..5 poses, segments up to 1.5 m, heading along the segment
            k = int(rng.integers(2, 6))
            start = rng.uniform(-half + 3.0, half - 3.0, size=2)
            xy = np.vstack([start, start + np.cumsum(rng.uniform(-1.5, 1.5, size=(k - 1, 2)), axis=0)])
            yaw = rng.uniform(-np.pi, np.pi, k)
            paths.append(np.hstack([xy, np.zeros((k, 3)), np.sin(yaw / 2)[:, None], np.cos(yaw / 2)[:, None]]))
        cons = (np.arange(len(paths)) % 2).astype(np.uint8)
        c.check_polygon_footprint_paths(paths[:1000], points, cons[:1000])  # warm-up
        off = np.zeros(len(paths) + 1, np.int32)
        off[1:] = np.cumsum([len(q) for q in paths])
        packed = np.concatenate(paths)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            safe, trav, area, st = c.check_polygon_footprint_paths_packed(off, packed, points, cons)
            d = time.perf_counter() - t0
            best = d if best is None or d < best else best
    m = 300
    g = O.geom(n, n, res)
    op = O.default_params(**{k: getattr(p, k) for k in ("normals_radius", "rough_radius", "step_radius1", "step_radius2",
                                                        "fp_radius", "fp_offset")})
    t0 = time.perf_counter()
    ws, wt, wa, wst = O.check_polygon_paths(g, op, elev, layers["traversability_slope"], layers["traversability_step"],
                                            layers["traversability_roughness"], layers["traversability"], paths[:m], points,
                                            cons[:m])
    dt_cpu = time.perf_counter() - t0  # includes the oracle's untraversable-cell pass over the whole map ...
    t0 = time.perf_counter()
    m3 = m + 20000
    O.check_polygon_paths(g, op, elev, layers["traversability_slope"], layers["traversability_step"],
                          layers["traversability_roughness"], layers["traversability"], paths[:m3], points, cons[:m3])
    dt_cpu3 = time.perf_counter() - t0  # ... which the difference of two calls removes
    assert np.array_equal(ws, safe[:m]) and np.array_equal(wt, trav[:m]) and np.array_equal(wa, area[:m])
    out["N2 checkFootprintPath (polygonal 0.9 x 0.6 m footprint), 100000 paths of 2-5 poses on 4096x4096"] = {
        "gpu_paths_per_s": len(paths) / best, "gpu_ms": best * 1e3, "safe_fraction": float(safe.mean()),
        # (null when the difference drowns in the run-to-run variation of the mask pass)
        "cpu_oracle_paths_per_s": (m3 - m) / (dt_cpu3 - dt_cpu) if dt_cpu3 - dt_cpu > 0.05 * dt_cpu else None,
        "what": "te_check_polygon_footprint_paths on packed host arrays: hulls and areas on the host (1 thread), one launch for "
                "all segment polygons, results back; the oracle (1 thread) timed as the difference of a 20300-path and a 300-path "
                "call (each also recomputes the untraversable-cell mask of the whole map); first 300 results bit-identical"}


def run_n3(capi, synth, res, out):
    # ---- N3: traversabilityFootprint(footprintYaw): the polygon footprint layers over the whole map ------------
    from oracle import oracle as O
    footprint = [[0.45, 0.30], [0.45, -0.30], [-0.45, -0.30], [-0.45, 0.30]]  # robot_footprint_parameter.yaml:3
    yaw = math.pi / 2                                                          # footprint_yaw default
    n = 4096
    elev = synth.perlin_elevation(n, n, seed=1235)
    with capi.Context(0) as c:
        c.set_params(params(capi, synth, 9.0, res))
        c.set_geometry(n, n, 1, res)
        c.upload_elevation(elev)
        c.run_chain(capi.RUN_FOOTPRINT)
        c.run_polygon_footprint(footprint, yaw)  # warm-up (allocates the two layers)
        c.sync()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            c.run_polygon_footprint(footprint, yaw)
            c.sync()
            d = time.perf_counter() - t0
            best = d if best is None or d < best else best
        tx = c.download("traversability_x").reshape(n, n)
        # the same footprint 3 % larger: no edge passes through a cell centre, no offset is left to per-cell rounding
        off_grid = [[1.03 * x, 1.03 * y] for x, y in footprint]
        c.run_polygon_footprint(off_grid, yaw)
        c.sync()
        best_off = None
        for _ in range(3):
            t0 = time.perf_counter()
            c.run_polygon_footprint(off_grid, yaw)
            c.sync()
            d = time.perf_counter() - t0
            best_off = d if best_off is None or d < best_off else best_off
    # the oracle on a 384 x 384 crop (its own chain first; the polygon pass alone is timed)
    m = 384
    crop = np.ascontiguousarray(elev.reshape(n, n)[:m, :m])
    g = O.geom(m, m, res)
    op = O.default_params(**{k: getattr(params(capi, synth, 9.0, res), k) for k in
                             ("normals_radius", "rough_radius", "step_radius1", "step_radius2", "fp_radius", "fp_offset")})
    L = O.chain(g, op, crop)
    t0 = time.perf_counter()
    O.polygon_footprint(g, op, crop, L["traversability_slope"], L["traversability_step"], L["traversability_roughness"],
                        L["traversability"], footprint, yaw)
    dt_cpu = time.perf_counter() - t0
    out["N3 traversabilityFootprint(yaw): traversability_x + traversability_rot, 0.9 x 0.6 m footprint on 4096x4096"] = {
        "gpu_ms": best * 1e3, "gpu_cells_per_s": n * n / best, "gpu_ms_footprint_off_the_cell_centres": best_off * 1e3, "untraversable_fraction_x": float((tx == 0).mean()),
        "cpu_oracle_cells_per_s": m * m / dt_cpu,
        "what": "host-timed te_run_polygon_footprint + te_sync on resident layers (both polygons for every cell); the "
                "oracle (1 thread, untraversable mask included) on a 384 x 384 crop"}


def main():
    from traversability_estimation_amd import capi, synth
    capi.load()
    res = 0.05
    out = {}
    only = [k for k in os.environ.get("TE_CONFIGS", "").split(",") if k]  # e.g. TE_CONFIGS=N3,N3P runs those alone
    for name, fn in (("bag", run_bag), ("cfg2", run_cfg2), ("cfg4", run_cfg4), ("cfg5", run_cfg5), ("N2", run_n2), ("N2P", run_n2p), ("N3", run_n3)):
        if not only or name in only:
            fn(capi, synth, res, out)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
