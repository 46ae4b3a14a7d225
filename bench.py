#!/usr/bin/env python3
"""bench.py -- map cells/s through the full traversability filter chain on MI355X.

Workload (BASELINE.json configs[2], the HBM-roofline config): one 4096 x 4096 synthetic elevation map
per GPU, res 0.05 m, all four filter radii 9 cells (tie-free, r = 9*res*(1+1e-6)), followed by the
circular traversability_footprint pass (radius 0.30 m, offset 0.15 m).  A "step" is one pass of the
whole chain over the resident batch; the elevation is already in HBM when the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): the batch axis is sharded -- every rank
filters its own map (weak scaling, no data-path collective); rank 0's filter parameters are broadcast
once over RCCL before the timed region.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is for the whole chain launch
sequence: algorithmic bytes (20 B/cell chain, 24 B/cell with the footprint pass) divided
by the chain's duration measured with HIP events on the stream the kernels run on (median of >= 100
launches, one event pair per launch); `roofline.dominant_kernel` is the normals/slope/roughness pass timed
alone the same way (12 B/cell); `roofline_issue` prices the same launch against the double-precision
issue rate, which is what bounds these kernels.
`cpu_baseline` times the CPU oracle (our restatement of the reference; kind "port") on one host thread
over a bounded crop of the same map; `cpu_baseline_all_cores` the same oracle with OpenMP over rows on every host core.
`parity_check` (every run, rank 0): the layers the timed launches left on the device are compared with the oracle on EVERY
cell of the map (up to 4096^2; `--check-crops`: a corner crop and one full-width band); a mismatch fails the run (exit
code 1) after the line is printed.
`ranks`: how many ranks ran, over which backend, on which device each.  `--gpus N` without a launcher starts the N ranks
itself (torch.distributed.run, one per GPU); a WORLD_SIZE that differs from --gpus is an error.

Order of the run: upload -> event-timed samples of the launch (they also bring the GPU to its working clocks) -> W
warm-up steps -> barrier + sync -> K timed steps -> sync + barrier -> parity check -> host path -> CPU baselines.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
N_SIMDS, CLOCK_HZ = 1024, 2.4e9
FP64_LANE_OPS_PEAK = 39.3e12  # 78.6 TFLOP/s vector fp64 = 39.3e12 fused multiply-adds (lane operations) per second
# double-precision lane operations per cell of one launch: the sliding moments of k_normals3 (6 per disc
# column and edge + 1 per distinct run length: 118 at R = 9, scaled with 2R+1 for other radii) and its tail (31)
def fp64_lane_ops_per_cell(radius_cells, with_footprint):
    cols = 2 * int(radius_cells) + 1
    return 6.2 * cols + 31  # (the footprint's sum has been 32-bit fixed point since round 3: no fp64 work there)


def kernel_sources_sha16():
    """Identifies the kernels a committed counter profile belongs to (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "traversability_estimation_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--size", type=int, default=4096, help="map is size x size cells")
    ap.add_argument("--radius-cells", type=float, default=9.0)
    ap.add_argument("--res", type=float, default=0.05)
    ap.add_argument("--maps-per-gpu", type=int, default=1)
    ap.add_argument("--no-footprint", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive measurement (for profiler runs: "
                    "its kernels wait for a pageable H2D copy and would skew the per-kernel averages)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the OpenMP all-core run of the oracle")
    ap.add_argument("--cpu-all-cores", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle parity check of the timed result (profiler runs)")
    ap.add_argument("--check-crops", action="store_true", help="parity check on a corner crop and a full-width band (244 496 cells per layer) instead "
                    "of EVERY cell of the map (the default up to 4096^2: the OpenMP oracle takes a few seconds for it on the GPU box's host)")
    ap.add_argument("--holes", type=float, default=0.0, help="fraction of invalid (NaN) cells: speckle if < 0.5, else "
                    "solid unobserved regions covering about (value - 0.5) of the map (not the BASELINE workload)")
    ap.add_argument("--sequential", action="store_true", help="profiling aid: no two-stream overlap inside the chain")
    ap.add_argument("--config", choices=("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"), default="cfg3",
                    help="BASELINE.json configs[k-1].  cfg1: the reference's bag map (100 x 133 at 0.03 m), default YAML, chain.  cfg2: one "
                         "1024^2 map (seed 1234), radius 5, chain.  cfg3 (default, the headline): one 4096^2 map per GPU, radius 9, "
                         "footprint pass (weak scaling).  cfg4: a batch of 512 maps of 512^2, radius 5, cut into contiguous blocks "
                         "over the ranks (dist.shard_range; strong scaling, params broadcast over RCCL).  cfg5: 8192^2 resident map, "
                         "radius 5, one 256^2 dirty tile per step: tile H2D + region re-filter + tile D2H on the copy streams")
    ap.add_argument("--footprint", action="store_true", help="cfg1 / cfg2 / cfg5 run the chain alone by default (what BASELINE names): add the footprint pass")
    ap.add_argument("--yaml", help="filter parameters from a reference-format robot_filter_parameter.yaml, unchanged (instead of the "
                                   "config's tie-free benchmark radii; the file's radii are metres: mind --res)")
    ap.add_argument("--footprint-yaml", help="robot_footprint_parameter.yaml (circular_footprint_radius_inscribed / _offset, traversability_default, ...)")
    ap.add_argument("--robot-yaml", help="robot.yaml (max_gap_width)")
    ap.add_argument("--tile", type=int, default=256, help="cfg5: side of the dirty tile")
    ap.add_argument("--in-flight", type=int, default=2, help="cfg5, streamed ticks: ticks in flight on the copy streams before the host waits "
                    "(the context has two staging slots each way: more than 2 in flight wait for a slot -- 0.16 ms per tick at 1 - 2, 0.29 at 4 - 8)")
    ap.add_argument("--tick-mode", choices=("auto", "sync", "stream"), default="auto",
                    help="cfg5: sync = te_upload_tile + te_run_chain_region + te_download_tile, one blocking call after the other (lowest latency "
                         "for small tiles); stream = the asynchronous pair on the copy streams (throughput for large tiles); auto: sync up to 512^2")
    ap.add_argument("--check", action="store_true", help="(default now; kept for old command lines)")
    return ap.parse_args()


def make_params(capi, synth, args):
    """(te_params, extra run flags).  --yaml: the reference's files unchanged (params_yaml.py); cfg1: the shipped defaults
    (= robot_filter_parameter.yaml, tests/test_params_yaml.py); otherwise the config's tie-free benchmark radii."""
    if args.yaml:
        from traversability_estimation_amd import params_yaml
        return params_yaml.params_from_yaml(capi, args.yaml, args.footprint_yaml, args.robot_yaml)
    if args.config == "cfg1":
        return capi.default_params(), 0
    r = synth.benchmark_radius(args.radius_cells, args.res)
    return capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(6.0, args.res),
                               fp_offset=synth.benchmark_radius(3.0, args.res)), 0


def cpu_baseline(args, elev_full, p, with_footprint, threads=1, pos=(0.0, 0.0)):
    """Time the CPU oracle (single thread, like the reference; or OpenMP over rows) on a bounded crop of the same map.
    threads = 0: the thread count is searched first -- a ladder 1, 2, 4, ... up to os.cpu_count() on a small crop, the
    fastest rung is used (a box whose container may only use part of its cores runs SLOWER with one thread per visible
    core: round 3 reported 5.8 x one thread on 256 threads).  elev_full: [cols][rows]; a map smaller than the sample
    (cfg1) is filtered whole, as many times as the sample's seconds hold."""
    from oracle import oracle as O
    op = O.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    cols_full, rows_full = elev_full.shape
    side = min(rows_full, cols_full)
    whole = rows_full * cols_full <= 256 * 256

    def run(n):
        """one pass over an n x n corner crop (or the whole small map): (cells, seconds)"""
        if whole:
            g, crop = O.geom(rows_full, cols_full, args.res, pos), elev_full
        else:
            g, crop = O.geom(n, n, args.res), np.ascontiguousarray(elev_full[:n, :n])
        t0 = time.perf_counter()
        out = O.chain(g, op, crop)
        if with_footprint:
            O.footprint(g, op, crop, out)
        return crop.size, time.perf_counter() - t0

    ladder = None
    if threads == 0:
        ladder, t, best = {}, 1, None
        ncpu = os.cpu_count() or 1
        while True:
            O.set_threads(t)
            cells, dt = run(min(side, 512))
            ladder[t] = cells / dt
            if best is None or ladder[t] > ladder[best]:
                best = t
            if t >= ncpu:
                break
            t = min(2 * t, ncpu)
        threads = best
    O.set_threads(threads)
    cells, dt = run(128 if threads == 1 else min(side, 1024))  # (calibration sample: enough rows for every thread)
    rate = cells / dt
    # scale the sample so that it takes about cpu_seconds, at most the whole map
    seconds = args.cpu_seconds if threads == 1 else min(args.cpu_seconds, 10.0)
    if whole:
        reps = int(max(1, min(2000, rate * seconds / cells)))
        t0 = time.perf_counter()
        for _ in range(reps):
            run(0)
        dt = time.perf_counter() - t0
        cells, what = cells * reps, f"the whole {rows_full}x{cols_full} map, {reps} passes"
    else:
        n = int(min(side, max(128, (rate * seconds) ** 0.5)))
        n = max(n - n % 64, 128)
        cells, dt = run(n)
        what = f"{n}x{n} crop of the same map"
    O.set_threads(1)
    res = {"value": cells / dt, "unit": "cells/s", "cores": threads, "kind": "port",
           "sample": f"{what}, full chain{'+footprint' if with_footprint else ''}, "
                     f"{dt:.1f} s on {threads} of {os.cpu_count()} host cores (oracle/te_oracle.c, -O3"
                     f"{', OpenMP over rows' if threads > 1 else ''})"}
    if ladder is not None:
        res["thread_ladder_cells_per_s"] = {str(k): round(v) for k, v in ladder.items()}
    return res


def parity_check(args, ctx, elev, p, with_fp, rows, cols, whole=False, pos=(0.0, 0.0), around=None):
    """The layers on the device (map 0 of this rank) against the oracle: every cell of the map (whole), or a corner crop
    (two map borders) and one full-width band (every block column and strip seam of the marching kernels); cells closer to
    a cut edge of a crop than the reach of the chain (+ the footprint's) see a cut neighbourhood there and are left out.
    elev: [cols][rows] (grid_map storage order)."""
    from oracle import oracle as O
    from tests.helpers import OUT_LAYERS, TOL, compare_layer
    names = list(OUT_LAYERS) + (["traversability_footprint"] if with_fp else [])
    op = O.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    per = rows * cols
    got = {k: ctx.download(k).reshape(-1)[:per].reshape(cols, rows) for k in names}  # [col j][row i]
    reach = max(p.normals_radius, p.rough_radius, p.step_radius1 + p.step_radius2) / args.res
    R = int(reach + 1)
    margin = 2 * R + (int((p.fp_radius + p.fp_offset) / args.res) + 5 if with_fp else 0) + 2
    crop_r, crop_c = min(rows, 320), min(cols, 320)
    band = min(cols, 2 * margin + 40)
    windows = [("corner crop %dx%d" % (crop_r, crop_c), (slice(0, crop_c), slice(0, crop_r)))]
    if whole:
        windows = [("whole map %dx%d" % (rows, cols), (slice(0, cols), slice(0, rows)))]
    elif around:  # windows around given tiles (r0, c0, side): the tile and twice the margin on every side, clipped
        windows = []
        for r0, c0, side in around:
            rs = slice(max(0, r0 - 2 * margin), min(rows, r0 + side + 2 * margin))
            cs = slice(max(0, c0 - 2 * margin), min(cols, c0 + side + 2 * margin))
            windows.append(("tile at row %d, column %d" % (r0, c0), (cs, rs)))
    elif cols > crop_c:
        j0 = (cols // 2 // 64) * 64 + 17  # not aligned with anything
        j0 = min(j0, cols - band)
        windows.append(("full-width band, columns %d..%d" % (j0, j0 + band), (slice(j0, j0 + band), slice(0, rows))))
    O.set_threads(min(os.cpu_count() or 1, 64))
    rep = {"tolerance": TOL, "windows": [], "layers": {k: {"mismatches": 0, "max_abs_err": 0.0, "cells": 0} for k in names}, "ok": True}
    full = (cols, rows)
    try:
        for label, sl in windows:
            sub = np.ascontiguousarray(elev.reshape(cols, rows)[sl])
            # rows = extent along i (the fast axis); a crop has its own positions, the whole map the map's
            g = O.geom(sub.shape[1], sub.shape[0], args.res, pos if whole else (0.0, 0.0))
            want = O.chain(g, op, sub)
            if with_fp:
                want["traversability_footprint"] = O.footprint(g, op, sub, want)
            keep = []
            for ax, s1 in enumerate(sl):
                lo = 0 if s1.start == 0 else margin
                hi = (s1.stop - s1.start) if s1.stop == full[ax] else (s1.stop - s1.start) - margin
                keep.append(slice(lo, hi))
            keep = tuple(keep)
            rep["windows"].append(label)
            for k in names:
                a = got[k][sl][keep]
                b = want[k].reshape(sub.shape)[keep]
                n_bad, mx, _ = compare_layer(k, a, b)
                L = rep["layers"][k]
                L["mismatches"] += n_bad
                L["max_abs_err"] = max(L["max_abs_err"], mx)
                L["cells"] += int(a.size)
                rep["ok"] = rep["ok"] and n_bad == 0
    finally:
        O.set_threads(1)
    return rep


def main():
    args = parse()
    import torch
    from traversability_estimation_amd import capi, synth

    from traversability_estimation_amd import dist as tdist
    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same command
    # line under torch.distributed.run) -- a single process must never report an N-GPU figure
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(tdist.relaunch_under_torchrun(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    # TE_DIST_BACKEND=gloo lets a 1-GPU box exercise the N>1 code path (ranks then share device 0)
    backend_req = os.environ.get("TE_DIST_BACKEND")
    n_dev = torch.cuda.device_count()
    if args.gpus > max(1, n_dev) and (backend_req or "nccl") == "nccl":
        sys.exit(f"bench.py: --gpus {args.gpus} but this node has {n_dev} GPU(s): RCCL needs one device per rank "
                 "(TE_DIST_BACKEND=gloo lets the ranks share a device to exercise the code path; it is not a scaling figure)")
    rank, world, local_rank = tdist.init_process_group(backend_req)
    local_rank %= max(1, n_dev)
    import torch.distributed as dist
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 "(or without a launcher: bench.py starts its own ranks)")
    torch.cuda.set_device(local_rank)
    capi.load()

    total_maps = None
    rows = cols = args.size
    pos = (0.0, 0.0)
    fp_default_on = True
    bag = None
    if args.config == "cfg1":  # the reference's own map (tests/golden/bag_map.npz, decoded from TE/maps/elevation_map.bag)
        bag = np.load(os.path.join(ROOT, "tests", "golden", "bag_map.npz"))
        rows, cols, args.res, pos = int(bag["rows"]), int(bag["cols"]), float(bag["resolution"]), tuple(float(v) for v in bag["position"])
        args.radius_cells, fp_default_on = 0.05 / args.res, False
    elif args.config == "cfg2":
        rows = cols = args.size = 1024
        args.radius_cells, fp_default_on = 5.0, False
    elif args.config == "cfg4":  # (before the parameters are built: they carry the radius)
        rows = cols = args.size = 512
        args.radius_cells, total_maps = 5.0, 512
        a_, b_ = tdist.shard_range(total_maps, rank, world)
        args.maps_per_gpu = b_ - a_
    elif args.config == "cfg5":
        rows = cols = args.size = 8192
        args.radius_cells, fp_default_on = 5.0, False
    cells = rows * cols

    # filter parameters: rank 0 decides, every other rank receives the te_params blob over RCCL
    p0, yaml_flags = make_params(capi, synth, args)
    p = tdist.broadcast_params(capi, p0, src=0)
    with_fp = (fp_default_on or args.footprint) and not args.no_footprint
    flags = (capi.RUN_FOOTPRINT if with_fp else 0) | yaml_flags
    if args.sequential:
        flags |= capi.RUN_SEQUENTIAL
    if args.yaml:  # (the figures that depend on the radius -- fp64 operations per cell, parity margins -- follow the file)
        args.radius_cells = max(p.normals_radius, p.rough_radius) / args.res
    n = args.size
    B = args.maps_per_gpu
    # shard of the batch owned by this rank: maps rank*B .. rank*B+B-1 (seed = 1235 + global map index)
    if bag is not None:
        elevs = [np.ascontiguousarray(bag["elevation"], dtype=np.float32).reshape(cols, rows) for _ in range(B)]
    elif args.config == "cfg2":
        elevs = [synth.perlin_elevation(n, n, seed=1234 + rank * B + b) for b in range(B)]
    elif args.config == "cfg5":
        # (the generator needs a minute for 8192^2: a 2048^2 map mirrored into a seamless 4096^2 one, repeated 2 x 2)
        a_ = synth.perlin_elevation(2048, 2048, seed=77 + rank).reshape(2048, 2048)
        m_ = np.block([[a_, a_[:, ::-1]], [a_[::-1, :], a_[::-1, ::-1]]])
        elevs = [np.tile(m_, (2, 2)).astype(np.float32)]
        B = args.maps_per_gpu = 1
    elif total_maps is None:
        elevs = [synth.perlin_elevation(n, n, seed=1235 + rank * B + b) for b in range(B)]
    else:  # cfg4: one base map + N(0, 1 cm) perturbations, seed = global map index (the MPC-rollout shape)
        base = synth.perlin_elevation(n, n, seed=2000)
        first = tdist.shard_range(total_maps, rank, world)[0]
        elevs = [(base + np.random.default_rng(2000 + first + b).normal(0.0, 0.01, size=base.shape).astype(np.float32)).astype(np.float32)
                 for b in range(B)]
    if args.holes > 0.0:
        rng = np.random.default_rng(99)
        for b in range(B):
            if args.holes < 0.5:
                elevs[b] = synth.with_holes(elevs[b], args.holes, seed=99 + b)
            else:  # rectangles of 100..400 cells a side until the requested area is covered
                area, target = 0, (args.holes - 0.5) * cells
                while area < target:
                    h, w = (int(v) for v in rng.integers(100, 400, size=2))
                    r0, c0 = int(rng.integers(0, rows - h)), int(rng.integers(0, cols - w))
                    elevs[b][c0:c0 + w, r0:r0 + h] = np.nan
                    area += h * w
    ranks = tdist.ranks_report(local_rank)  # (collective: every rank calls it)
    ctx = capi.Context(local_rank)
    ctx.set_params(p)
    ctx.set_geometry(rows, cols, B, args.res, pos)
    ctx.upload_elevation(np.stack(elevs))

    barrier = tdist.barrier

    tick = None
    if args.config == "cfg5":
        # one step = one tick: a dirty tile arrives (H2D on copy stream A), the tile dilated by the chain's reach is filtered
        # again, the tile's result leaves (D2H on copy stream B); `--in-flight` ticks are queued before the host waits.
        # The new content of a tile is the old one plus a smooth bump that vanishes at the tile's edge (a sensor update
        # does not tear the map).  16 fixed origins (PRNG seed 77), page-locked producer / consumer buffers.
        T = args.tile
        rng5 = np.random.default_rng(77)
        w1 = np.hanning(T).astype(np.float32)
        origins = [tuple(int(v) for v in rng5.integers(0, rows - T, size=2)) for _ in range(16)]
        host_map = elevs[0]
        tiles_in = []
        for k, (r0, c0) in enumerate(origins):
            bump = np.outer(w1, w1) * (0.2 * synth.perlin_elevation(T, T, seed=1000 + k).reshape(T, T))
            tiles_in.append(np.ascontiguousarray(host_map[c0:c0 + T, r0:r0 + T] + bump.astype(np.float32)))
        tiles_out = [np.empty((T, T), np.float32) for _ in range(8)]
        for buf in tiles_in + tiles_out:
            capi.pin_host(buf)
        out_layer = "traversability_footprint" if with_fp else "traversability"
        ctx.run_chain(flags)  # the resident map's layers, once
        ctx.sync()
        tick = {"k": 0}

        tick_mode = args.tick_mode if args.tick_mode != "auto" else ("sync" if T <= 512 else "stream")

        def tick_stream():
            k = tick["k"]
            r0, c0 = origins[k % 16]
            ctx.upload_tile_async(tiles_in[k % 16], 0, r0, c0)
            ctx.run_chain_region(0, r0, c0, T, T, flags=flags)
            ctx.download_tile_async(out_layer, 0, r0, c0, tiles_out[k % 8])
            host_map[c0:c0 + T, r0:r0 + T] = tiles_in[k % 16]  # (the host's copy of the resident map, for the parity check)
            tick["k"] = k + 1
            if (k + 1) % max(1, args.in_flight) == 0:
                ctx.sync()

        def tick_sync():
            k = tick["k"]
            r0, c0 = origins[k % 16]
            ctx.upload_tile(tiles_in[k % 16], 0, r0, c0)
            ctx.run_chain_region(0, r0, c0, T, T, flags=flags)
            tick["last_out"] = ctx.download_tile(out_layer, 0, r0, c0, T, T)
            host_map[c0:c0 + T, r0:r0 + T] = tiles_in[k % 16]
            tick["k"] = k + 1

        step = tick_sync if tick_mode == "sync" else tick_stream

        # latency of ONE tick, host-timed (tile in, region run, tile out, wait), median of 48
        lat = []
        for _ in range(56):
            t0 = time.perf_counter()
            step()
            ctx.sync()
            lat.append((time.perf_counter() - t0) * 1e3)
        chain_samples = np.array(lat[8:])
        n_samples = len(chain_samples)
        ms_chain = float(np.median(chain_samples))
        ms_normals = None
        cells_timed = T * T  # per step and rank
    else:
        def step():
            ctx.run_chain(flags)

        # kernel-only duration of the chain: HIP events on the context's own stream, one pair per launch, median of >= 100.
        # Taken BEFORE the host-timed loop: the same launches, and they leave the GPU at its working clocks (the driver's
        # 5 warm-up + 20 timed steps alone start on an idle device).
        n_samples = max(100, args.steps)
        chain_samples = ctx.time_chain_samples(flags, warmup=20, iters=n_samples)
        ms_chain = float(np.median(chain_samples))
        # the dominant kernel alone (normals/slope/roughness + its fix-up pass: TE_RUN_NORMALS_ONLY), the same way
        normals_samples = ctx.time_chain_samples(capi.RUN_NORMALS_ONLY | (flags & capi.RUN_KEEP_NORMALS), warmup=5, iters=n_samples)
        ms_normals = float(np.median(normals_samples))
        ctx.run_chain(flags)  # (the normals-only launches overwrote two layers with the same values; keep the state simple)
        cells_timed = B * cells

    for _ in range(args.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dt = tdist.max_over_ranks(dt)

    # parity of what the timed launches left on the device (rank 0, map 0)
    check = None
    if rank == 0 and not args.no_check:
        if tick is not None:
            # cfg5: the region runs must have left the resident layers as a whole-map run of the final elevation would:
            # windows around the last four tiles (each dilated by twice the parity margin), against the oracle
            T = args.tile
            last = [origins[(tick["k"] - 1 - q) % 16] for q in range(4)]
            check = parity_check(args, ctx, host_map, p, with_fp, rows, cols, pos=pos, around=[(r0, c0, T) for r0, c0 in last])
        else:
            check = parity_check(args, ctx, elevs[0], p, with_fp, rows, cols, whole=(not args.check_crops) and cells <= 4096 * 4096, pos=pos)
    if tick is not None:
        other = tick_stream if tick_mode == "sync" else tick_sync  # the other form, 128 ticks, for the record
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(128):
            other()
        ctx.sync()
        tick["other_ms"] = (time.perf_counter() - t0) / 128 * 1e3
        for buf in tiles_in + tiles_out:
            capi.unpin_host(buf)

    # plugin-shaped path: host buffers in, host buffers out (PCIe both ways); reported next to, never as, `value`
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path and total_maps is None and args.config == "cfg3":  # N = 1 only: at N > 1 the other ranks would wait for it
        stack = np.stack(elevs)
        names = ["traversability_slope", "traversability_step", "traversability_roughness", "traversability"]
        if with_fp:
            names.append("traversability_footprint")
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.upload_elevation(stack)
            ctx.run_chain(flags)
            outs = [ctx.download(k) for k in names]
            ctx.sync()
            d = time.perf_counter() - t0
            best = d if best is None or d < best else best
        del outs
        host_path = {"ms": best * 1e3, "cells_per_s": B * cells / best,
                     "what": f"upload elevation + chain + download {len(names)} layers through pageable host buffers, best of 3"}
        # the same with buffers the host keeps across frames and has page-locked once (te_pin_host)
        bufs = [np.empty(stack.size, np.float32) for _ in names]
        pinned = []
        try:
            for b in [stack] + bufs:
                capi.pin_host(b)
                pinned.append(b)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.upload_elevation(stack)
                ctx.run_chain(flags)
                for k, b in zip(names, bufs):
                    ctx.download_into(k, b)
                ctx.sync()
                d = time.perf_counter() - t0
                best = d if best is None or d < best else best
            host_path["pinned_ms"] = best * 1e3
            host_path["pinned_cells_per_s"] = B * cells / best
        except capi.TeError as e:  # page-locking can be refused (ulimit -l); the pageable figure stands
            host_path["pinned_error"] = str(e)
        finally:
            for b in pinned:
                try:
                    capi.unpin_host(b)
                except capi.TeError:
                    pass
        del bufs
        # the unchanged-YAML plugin sequence (SlopeFilter -> StepFilter -> RoughnessFilter, normals from the upstream host
        # filter): per-plugin entry points, every layer uploaded once (DeviceMap keeps what is resident), pageable buffers
        if B == 1:
            try:
                ctx.run_chain(capi.RUN_KEEP_NORMALS)
                nrm = [ctx.download(k) for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z")]
                best = None
                def three_plugins(prefetch):
                    t0 = time.perf_counter()
                    # SlopeFilter::update (plugins/src/SlopeFilter.cpp)
                    ctx.upload_layer("surface_normal_z", nrm[2])
                    if prefetch:
                        ctx.prefetch_layers({"elevation": stack})
                    ctx.run_filter("slope")
                    o1 = ctx.download("traversability_slope")
                    if prefetch:
                        ctx.wait_prefetch()
                    # StepFilter::update
                    if not prefetch:
                        ctx.upload_elevation(stack)
                    else:
                        ctx.prefetch_layers({"surface_normal_x": nrm[0], "surface_normal_y": nrm[1]})
                    ctx.run_filter("step")
                    o2 = ctx.download("traversability_step")
                    if prefetch:
                        ctx.wait_prefetch()
                    # RoughnessFilter::update
                    if not prefetch:
                        ctx.upload_layer("surface_normal_x", nrm[0])
                        ctx.upload_layer("surface_normal_y", nrm[1])
                    ctx.run_filter("roughness")
                    o3 = ctx.download("traversability_roughness")
                    ctx.sync()
                    dt3 = (time.perf_counter() - t0) * 1e3
                    del o1, o2, o3
                    return dt3
                # the two forms take turns (what a download into a freshly allocated array costs -- first-touch page faults,
                # 1.7 to 5 ms per 64 MB -- drifts over the life of a process: each form should see the same weather)
                runs = {False: [], True: []}
                for _ in range(4):
                    for prefetch in (False, True):
                        runs[prefetch].append(three_plugins(prefetch))
                host_path["three_plugins_ms"] = min(runs[False])
                host_path["three_plugins_prefetch_ms"] = min(runs[True])
                host_path["three_plugins_runs_ms"] = {"one_transfer_at_a_time": [round(v, 2) for v in runs[False]],
                                                      "with_prefetches": [round(v, 2) for v in runs[True]]}
                host_path["three_plugins_what"] = ("te_run_filter(slope / step / roughness) with host layers in and out, 4 uploads "
                                                   "(elevation and surface_normal_z once), 3 downloads into freshly allocated arrays, pageable "
                                                   "buffers, best of 4 (all runs in three_plugins_runs_ms, the two forms taking turns), one transfer "
                                                   "at a time; _prefetch_: the uploads of the NEXT plugin's inputs "
                                                   "start beside each plugin's kernel and download (te_prefetch_layers, as plugins/src/DeviceMap.cpp "
                                                   "does).  A 64 MB download beside a 64 MB upload takes 1.8 ms against 1.4 + 1.6 one after the other "
                                                   "(tools/lab/prefetch_ab.py, preallocated buffers), but here the first-touch page faults of the fresh "
                                                   "output arrays (1.7 - 5 ms per layer from pass to pass of one process) decide which line is lower")
                del nrm
            except (capi.TeError, KeyError, TypeError, ValueError) as e:
                host_path["three_plugins_error"] = str(e)

    if rank == 0:
        cells_per_step = total_maps * cells if total_maps is not None else world * cells_timed
        bytes_per_cell = 24 if with_fp else 20
        achieved = cells_timed * bytes_per_cell / (ms_chain * 1e-3) / 1e9
        is_cfg3 = args.config == "cfg3" and with_fp and n == 4096 and B == 1 and args.radius_cells == 9.0 and args.holes == 0.0 and not args.yaml
        what = {"cfg1": "the reference's bag map (TE/maps/elevation_map.bag), default robot_filter_parameter.yaml: ",
                "cfg2": "", "cfg3": "", "cfg4": f"batch of {total_maps} maps cut over {world} rank(s), ",
                "cfg5": f"8192x8192 resident map, one {args.tile}x{args.tile} dirty tile per step (tile H2D + region re-filter + tile D2H; "
                        f"value counts the dirty tile's cells; the two tick forms: tick_mode): "}[args.config]
        out = {
            "metric": "map cells/s through full filter chain",
            "value": cells_per_step * args.steps / dt,
            "unit": "cells/s",
            "n_gpus": world,
            "ranks": ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # latency of ONE launch (HIP events around a single te_run_chain, median) beside the loop's throughput figure above
            "latency_ms_per_launch": ms_chain,
            # what the host-timed loop carries on top of K event-timed launches (first-launch latency, the final sync)
            "sync_overhead_ms": dt * 1e3 - args.steps * ms_chain,
            "higher_is_better": True,
            "scaling": "strong" if total_maps is not None else "weak",
            "vs_baseline": None,
            "dtype": "f64 moments and eigen-solve, f32 acos and score tail (step filter and combine: f32 compare/add, exact)",
            "data": ("the reference's fixture map (tests/golden/bag_map.npz)" if bag is not None else
                     "synthetic (gradient noise, 5 octaves, seed %s+map)" % {"cfg2": 1234, "cfg4": 2000, "cfg5": 77}.get(args.config, 1235))
                    + (f", holes {args.holes}" if args.holes else ""),
            "config": {"name": args.config,
                       "workload": what + f"{B} x {rows}x{cols} elevation map per GPU, res {args.res:g} m, radius {args.radius_cells:.3g} cells"
                                   f" (normals/roughness{'' if args.yaml or bag is not None else '/step'}), slope+roughness+step+normals+combine"
                                   f"{' + traversability_footprint pass' if with_fp else ''}"
                                   + (f"; parameters from {os.path.basename(args.yaml)}" if args.yaml else ""),
                       "maps_per_gpu": B, "map_cells": cells, "radius_cells": args.radius_cells,
                       "footprint": with_fp, "sharding": "batch axis, one map shard per rank, params broadcast over RCCL"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "chain launch sequence (all kernels of one te_run_chain)",
                         "ms_per_launch": ms_chain,
                         "ms_per_launch_stat": (f"median of {n_samples} ticks, host-timed one at a time (tile H2D + region run + tile D2H + wait; " if tick is not None
                                                else f"median of {n_samples} launches, one HIP event pair each (") +
                                               f"p10 {np.percentile(chain_samples, 10):.4f}, p90 {np.percentile(chain_samples, 90):.4f})",
                         "algorithmic_bytes_per_cell": bytes_per_cell},
        }
        if ms_normals is not None:
            out["roofline"]["dominant_kernel"] = {
                "name": "the normals / slope / roughness pass alone on the GPU (TE_RUN_NORMALS_ONLY: k_normals3s or k_normals3 + k_normals_fixup on the "
                        "headline map -- rocprofv3 lists the two separately, profiles/ --, k_normals_small on discs of at most 13 cells)",
                "ms": ms_normals, "algorithmic_bytes_per_cell": 12, "achieved": cells_timed * 12 / (ms_normals * 1e-3) / 1e9,
                "frac": cells_timed * 12 / (ms_normals * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if tick is not None:
            out["ticks_per_s"] = args.steps / dt
            out["tick_latency_ms"] = ms_chain
            out["tick_mode"] = {"timed": tick_mode, "what": "sync: te_upload_tile + te_run_chain_region + te_download_tile, blocking; stream: "
                                f"te_upload_tile_async / te_run_chain_region / te_download_tile_async on the copy streams, {args.in_flight} ticks in flight",
                                ("stream" if tick_mode == "sync" else "sync") + "_ms_per_tick": tick["other_ms"]}
            out["roofline"]["kernel"] = "one tick: tile H2D, region run of the chain (tile dilated by the reach), tile D2H -- latency-bound by construction"
        # second roofline: what actually bounds these kernels is instruction issue, most of it double precision
        ops = fp64_lane_ops_per_cell(args.radius_cells, with_fp) * cells_timed
        out["roofline_issue"] = {"bound": "valu-f64", "achieved": ops / (ms_chain * 1e-3), "peak": FP64_LANE_OPS_PEAK,
                                 "unit": "fp64 lane operations/s", "frac": ops / (ms_chain * 1e-3) / FP64_LANE_OPS_PEAK,
                                 "lane_ops_per_cell": fp64_lane_ops_per_cell(args.radius_cells, with_fp),
                                 "note": "algorithmic fp64 operations of the sliding moments, the tails and the footprint sum; "
                                         "strip warm-up, staging and the float32 work are not counted"}
        # HBM traffic of the same launch sequence: PMC counters need their own rocprofv3 passes (FETCH_SIZE and
        # WRITE_SIZE do not fit one pass), so the number comes from the committed profile of this exact workload --
        # and only if that profile was taken with the kernels of this tree (hash of csrc/)
        import glob
        tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True)  # newest round first
        if tpaths and is_cfg3:
            sha = kernel_sources_sha16()
            hit = None
            for tp in tpaths:
                t = json.load(open(tp))
                if t.get("kernel_sources_sha16") == sha:
                    hit = (tp, t)
                    break
            if hit:
                out["roofline"]["traffic"] = hit[1]["traffic_bytes"]
                out["roofline"]["traffic_unit"] = f"bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/{os.path.basename(hit[0])})"
            else:
                out["roofline"]["traffic_unit"] = f"not reported: profiles/{os.path.basename(tpaths[0])} was taken with other kernel sources"
        # What this FORMULATION can reach at most: the launch is bound by vector-instruction issue, so
        # the sum of the kernels' issue floors is a lower bound of its time, whatever the overlap.  Round 6 prices the
        # executed VALU instructions BY CLASS (tools/valu_classes.py -> profiles/rNN_valu_classes.json: fp64 as counted by
        # SQ_INSTS_VALU_*_F64 at 4.2 cycles per SIMD, the rest split by each kernel's loop mix into the 4.2-cycle class --
        # 3-operand, min / max, compare, select, convert --, the 2.2-cycle class -- plain float32 / integer -- and float32
        # transcendentals at 8.2); round 5 priced all of them at 4.2, which flattered the efficiency.  `frac` is to be read
        # against this ceiling, not against 1.
        vpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_classes.json")), reverse=True)
        if vpaths and is_cfg3:
            try:
                vc = json.load(open(vpaths[0]))
                if vc.get("kernel_sources_sha16") != kernel_sources_sha16():
                    out["roofline"]["formulation_ceiling_error"] = f"profiles/{os.path.basename(vpaths[0])} was taken with other kernel sources"
                else:
                    floor_cls, floor_all = vc["sum_issue_floor_us_by_class"], vc["sum_issue_floor_us_all_at_4.2"]
                    ceiling = cells_timed * bytes_per_cell / (floor_cls * 1e-6) / 1e9 / HBM_PEAK_GBS
                    out["roofline"]["formulation_ceiling"] = ceiling
                    out["roofline"]["frac_of_ceiling"] = out["roofline"]["frac"] / ceiling
                    out["roofline"]["formulation_ceiling_what"] = {
                        "issue_floor_us_by_class": {k: v.get("issue_floor_us_by_class") for k, v in vc["kernels"].items()},
                        "issue_floor_us_sum_by_class": floor_cls,
                        "issue_floor_us_sum_all_at_4.2": floor_all,
                        "formulation_ceiling_all_at_4.2 (round 5's model)": cells_timed * bytes_per_cell / (floor_all * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "model": "executed VALU instructions per launch (SQ counters) priced by class: fp64 / 3-operand / min-max / compare / select / convert "
                                 f"4.2 cycles per SIMD, plain float32 and 2-operand integer 2.2, float32 transcendental 8.2 (profiles/r01_valu_rates.txt), "
                                 f"/ ({N_SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz); the 2.2-cycle rate needs >= 4 waves per SIMD, so the floor is a lower bound "
                                 "of the time and the ceiling an upper bound of the formulation's reach",
                        "source": f"profiles/{os.path.basename(vpaths[0])} ({vc.get('counters')})",
                        "note": "an fp64 sliding-disc formulation at R = 9 is bound by vector-instruction issue, not by HBM: "
                                "the 50 % north-star presumes a bandwidth-bound stencil, which this is not (DESIGN.md 4)"}
            except (OSError, ValueError, KeyError) as e:
                out["roofline"]["formulation_ceiling_error"] = str(e)
        if host_path is not None:
            out["host_path"] = host_path
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, elevs[0], p, with_fp, pos=pos)
            if not args.no_cpu_all_cores:  # single thread AND OpenMP over rows on all host cores
                out["cpu_baseline_all_cores"] = cpu_baseline(args, elevs[0], p, with_fp, threads=0, pos=pos)  # (0: the fastest rung of a thread ladder)
        if check is not None:
            out["parity_check"] = check
        print(json.dumps(out), flush=True)
        if check is not None and not check["ok"]:
            print("parity check FAILED: " + json.dumps(check), file=sys.stderr)
            ctx.close()
            sys.exit(1)
    ctx.close()
    if dist.is_available() and dist.is_initialized():
        barrier()  # rank 0 is still timing the CPU baseline: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
