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
VALU_CYCLES_PER_INST, N_SIMDS, CLOCK_HZ = 4.2, 1024, 2.4e9  # measured issue rate of fp64 / 3-operand instructions per SIMD (profiles/r01_valu_rates.txt)
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
    ap.add_argument("--config", choices=("cfg3", "cfg4"), default="cfg3",
                    help="cfg3 (default): BASELINE.json configs[2], one 4096^2 map per GPU, radius 9, footprint pass (weak scaling). "
                         "cfg4: configs[3], a batch of 512 maps of 512^2, radius 5, cut into contiguous blocks over the ranks "
                         "(dist.shard_range; strong scaling, params broadcast over RCCL)")
    ap.add_argument("--check", action="store_true", help="(default now; kept for old command lines)")
    return ap.parse_args()


def make_params(capi, synth, args):
    r = synth.benchmark_radius(args.radius_cells, args.res)
    return capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r,
                               fp_radius=synth.benchmark_radius(6.0, args.res),
                               fp_offset=synth.benchmark_radius(3.0, args.res))


def cpu_baseline(args, elev_full, p, with_footprint, threads=1):
    """Time the CPU oracle (single thread, like the reference; or OpenMP over rows) on a bounded crop of the same map.
    threads = 0: the thread count is searched first -- a ladder 1, 2, 4, ... up to os.cpu_count() on a small crop, the
    fastest rung is used (a box whose container may only use part of its cores runs SLOWER with one thread per visible
    core: round 3 reported 5.8 x one thread on 256 threads)."""
    from oracle import oracle as O
    from tests.helpers import OUT_LAYERS  # noqa: F401
    op = O.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    ladder = None
    if threads == 0:
        n = min(args.size, 512)
        g = O.geom(n, n, args.res)
        crop = np.ascontiguousarray(elev_full[:n, :n])
        ladder, t, best = {}, 1, None
        ncpu = os.cpu_count() or 1
        while True:
            O.set_threads(t)
            t0 = time.perf_counter()
            out = O.chain(g, op, crop)
            if with_footprint:
                O.footprint(g, op, crop, out)
            dt = time.perf_counter() - t0
            ladder[t] = n * n / dt
            if best is None or ladder[t] > ladder[best]:
                best = t
            if t >= ncpu:
                break
            t = min(2 * t, ncpu)
        threads = best
    O.set_threads(threads)
    n = 128 if threads == 1 else min(args.size, 1024)  # (calibration sample: enough rows for every thread)
    g = O.geom(n, n, args.res)
    crop = np.ascontiguousarray(elev_full[:n, :n])
    t0 = time.perf_counter()
    out = O.chain(g, op, crop)
    if with_footprint:
        O.footprint(g, op, crop, out)
    dt = time.perf_counter() - t0
    rate = n * n / dt
    # scale the sample so that it takes about cpu_seconds, at most the whole map
    seconds = args.cpu_seconds if threads == 1 else min(args.cpu_seconds, 10.0)
    n = int(min(args.size, max(128, (rate * seconds) ** 0.5)))
    n -= n % 64
    n = max(n, 128)
    g = O.geom(n, n, args.res)
    crop = np.ascontiguousarray(elev_full[:n, :n])
    t0 = time.perf_counter()
    out = O.chain(g, op, crop)
    if with_footprint:
        O.footprint(g, op, crop, out)
    dt = time.perf_counter() - t0
    O.set_threads(1)
    res = {"value": n * n / dt, "unit": "cells/s", "cores": threads, "kind": "port",
           "sample": f"{n}x{n} crop of the same map, full chain{'+footprint' if with_footprint else ''}, "
                     f"{dt:.1f} s on {threads} of {os.cpu_count()} host cores (oracle/te_oracle.c, -O3"
                     f"{', OpenMP over rows' if threads > 1 else ''})"}
    if ladder is not None:
        res["thread_ladder_cells_per_s"] = {str(k): round(v) for k, v in ladder.items()}
    return res


def parity_check(args, ctx, elev, p, with_fp, n, whole=False):
    """The layers on the device (map 0 of this rank) against the oracle: every cell of the map (whole), or a corner crop
    (two map borders) and one full-width band (every block column and strip seam of the marching kernels); cells closer to
    a cut edge of a crop than the reach of the chain (+ the footprint's) see a cut neighbourhood there and are left out."""
    from oracle import oracle as O
    from tests.helpers import OUT_LAYERS, TOL, compare_layer
    names = list(OUT_LAYERS) + (["traversability_footprint"] if with_fp else [])
    op = O.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    per = n * n
    got = {k: ctx.download(k).reshape(-1)[:per].reshape(n, n) for k in names}  # [col j][row i]
    R = int(args.radius_cells + 1)
    margin = 2 * R + (R + 4 if with_fp else 0) + 2
    crop_n = min(n, 320)
    band = min(n, 2 * margin + 40)
    windows = [("corner crop %dx%d" % (crop_n, crop_n), (slice(0, crop_n), slice(0, crop_n)))]
    if whole:
        windows = [("whole map %dx%d" % (n, n), (slice(0, n), slice(0, n)))]
    elif n > crop_n:
        j0 = (n // 2 // 64) * 64 + 17  # not aligned with anything
        j0 = min(j0, n - band)
        windows.append(("full-width band, rows %d..%d" % (j0, j0 + band), (slice(j0, j0 + band), slice(0, n))))
    O.set_threads(min(os.cpu_count() or 1, 64))
    rep = {"tolerance": TOL, "windows": [], "layers": {k: {"mismatches": 0, "max_abs_err": 0.0, "cells": 0} for k in names}, "ok": True}
    try:
        for label, sl in windows:
            sub = np.ascontiguousarray(elev.reshape(n, n)[sl])
            g = O.geom(sub.shape[1], sub.shape[0], args.res)  # rows = extent along i (the fast axis)
            want = O.chain(g, op, sub)
            if with_fp:
                want["traversability_footprint"] = O.footprint(g, op, sub, want)
            keep = []
            for ax, s1 in enumerate(sl):
                lo = 0 if s1.start == 0 else margin
                hi = (s1.stop - s1.start) if s1.stop == n else (s1.stop - s1.start) - margin
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
    dist = None
    if world > 1:
        import torch.distributed as dist
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 "(or without a launcher: bench.py starts its own ranks)")
    torch.cuda.set_device(local_rank)
    capi.load()

    total_maps = None
    if args.config == "cfg4":  # (before the parameters are built: they carry the radius)
        args.size, args.radius_cells, total_maps = 512, 5.0, 512
        a, b = tdist.shard_range(total_maps, rank, world)
        args.maps_per_gpu = b - a

    # filter parameters: rank 0 decides, every other rank receives the te_params blob over RCCL
    p = tdist.broadcast_params(capi, make_params(capi, synth, args), src=0)
    with_fp = not args.no_footprint
    flags = capi.RUN_FOOTPRINT if with_fp else 0
    if args.sequential:
        flags |= capi.RUN_SEQUENTIAL
    n = args.size
    B = args.maps_per_gpu
    # shard of the batch owned by this rank: maps rank*B .. rank*B+B-1 (seed = 1235 + global map index)
    if total_maps is None:
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
                area, target = 0, (args.holes - 0.5) * n * n
                while area < target:
                    h, w = (int(v) for v in rng.integers(100, 400, size=2))
                    r0, c0 = int(rng.integers(0, n - h)), int(rng.integers(0, n - w))
                    elevs[b][c0:c0 + w, r0:r0 + h] = np.nan
                    area += h * w
    ranks = tdist.ranks_report(local_rank)  # (collective: every rank calls it)
    ctx = capi.Context(local_rank)
    ctx.set_params(p)
    ctx.set_geometry(n, n, B, args.res)
    ctx.upload_elevation(np.stack(elevs))

    barrier = tdist.barrier

    # kernel-only duration of the chain: HIP events on the context's own stream, one pair per launch, median of >= 100.
    # Taken BEFORE the host-timed loop: the same launches, and they leave the GPU at its working clocks (the driver's
    # 5 warm-up + 20 timed steps alone start on an idle device).
    n_samples = max(100, args.steps)
    chain_samples = ctx.time_chain_samples(flags, warmup=20, iters=n_samples)
    ms_chain = float(np.median(chain_samples))
    # the dominant kernel alone (normals/slope/roughness + its fix-up pass: TE_RUN_NORMALS_ONLY), the same way
    normals_samples = ctx.time_chain_samples(capi.RUN_NORMALS_ONLY, warmup=5, iters=n_samples)
    ms_normals = float(np.median(normals_samples))
    ctx.run_chain(flags)  # (the normals-only launches overwrote two layers with the same values; keep the state simple)

    for _ in range(args.warmup):
        ctx.run_chain(flags)
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.run_chain(flags)
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dt = tdist.max_over_ranks(dt)

    # parity of what the timed launches left on the device (rank 0, map 0)
    check = None
    if rank == 0 and not args.no_check:
        check = parity_check(args, ctx, elevs[0], p, with_fp, n, whole=(not args.check_crops) and n * n <= 4096 * 4096)

    # plugin-shaped path: host buffers in, host buffers out (PCIe both ways); reported next to, never as, `value`
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path and total_maps is None:  # N = 1 only: at N > 1 the other ranks would wait for it
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
        host_path = {"ms": best * 1e3, "cells_per_s": B * n * n / best,
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
            host_path["pinned_cells_per_s"] = B * n * n / best
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
        cells_per_step = (total_maps if total_maps is not None else world * B) * n * n
        bytes_per_cell = 24 if with_fp else 20
        achieved = B * n * n * bytes_per_cell / (ms_chain * 1e-3) / 1e9
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
            "data": "synthetic (gradient noise, 5 octaves, seed 1235+map)" + (f", holes {args.holes}" if args.holes else ""),
            "config": {"workload": (f"batch of {total_maps} maps cut over {world} rank(s), " if total_maps is not None else "") +
                                   f"{B} x {n}x{n} elevation map per GPU, res {args.res} m, radius {args.radius_cells:g} cells"
                                   f" (normals/roughness/step), slope+roughness+step+normals+combine"
                                   f"{' + traversability_footprint pass' if with_fp else ''}",
                       "maps_per_gpu": B, "map_cells": n * n, "radius_cells": args.radius_cells,
                       "footprint": with_fp, "sharding": "batch axis, one map shard per rank, params broadcast over RCCL"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "chain launch sequence (all kernels of one te_run_chain)",
                         "ms_per_launch": ms_chain, "ms_per_launch_stat": f"median of {n_samples} launches, one HIP event pair each "
                                                                          f"(p10 {np.percentile(chain_samples, 10):.4f}, p90 {np.percentile(chain_samples, 90):.4f})",
                         "algorithmic_bytes_per_cell": bytes_per_cell,
                         "dominant_kernel": {"name": "the normals pass (k_normals3s or k_normals3, + k_normals_fixup), alone on the GPU (TE_RUN_NORMALS_ONLY; rocprofv3 "
                                                     "lists the two kernels separately, profiles/)",
                                             "ms": ms_normals, "algorithmic_bytes_per_cell": 12,
                                             "achieved": B * n * n * 12 / (ms_normals * 1e-3) / 1e9,
                                             "frac": B * n * n * 12 / (ms_normals * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        # second roofline: what actually bounds these kernels is instruction issue, most of it double precision
        ops = fp64_lane_ops_per_cell(args.radius_cells, with_fp) * B * n * n
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
        if tpaths and with_fp and n == 4096 and B == 1 and args.radius_cells == 9.0 and args.holes == 0.0:
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
        # the sum of the kernels' VALU floors -- executed vector instructions (SQ_INSTS_VALU of the committed counter
        # profile) x 4.2 cycles per wavefront instruction / (1024 SIMDs x 2.4 GHz) -- is a lower bound of its time,
        # whatever the overlap.  `frac` is to be read against this ceiling, not against 1.
        spaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")), reverse=True)
        if spaths and with_fp and n == 4096 and B == 1 and args.radius_cells == 9.0 and args.holes == 0.0:
            try:
                sq = json.load(open(spaths[0]))
                floors = {}
                for kname, rec in sq.items():
                    if kname.startswith("k_combine") or kname.startswith("k_count_invalid") or kname == "kernel_sources_sha16":
                        continue  # (the sequential profile's separate combine: inside k_fp_mask in the timed launch; upload-time count)
                    v = rec.get("counters", {}).get("SQ_INSTS_VALU")
                    if v:
                        floors[kname] = v * VALU_CYCLES_PER_INST / (N_SIMDS * CLOCK_HZ) * 1e6
                floor_us = sum(floors.values())
                if floor_us > 0:
                    ceiling = B * n * n * bytes_per_cell / (floor_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                    out["roofline"]["formulation_ceiling"] = ceiling
                    out["roofline"]["frac_of_ceiling"] = out["roofline"]["frac"] / ceiling
                    out["roofline"]["formulation_ceiling_what"] = {
                        "valu_floor_us": {k: round(v, 1) for k, v in sorted(floors.items())}, "valu_floor_us_sum": round(floor_us, 1),
                        "model": f"SQ_INSTS_VALU x {VALU_CYCLES_PER_INST} cycles / ({N_SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz), every kernel of the launch; "
                                 "fp64 and 3-operand instructions issue once per 4.2 cycles per SIMD (profiles/r01_valu_rates.txt)",
                        "source": f"profiles/{os.path.basename(spaths[0])}",
                        "note": "an fp64 sliding-disc formulation at R = 9 is bound by vector-instruction issue, not by HBM: "
                                "the 50 % north-star presumes a bandwidth-bound stencil, which this is not (DESIGN.md 4)"}
            except (OSError, ValueError, KeyError) as e:
                out["roofline"]["formulation_ceiling_error"] = str(e)
        if host_path is not None:
            out["host_path"] = host_path
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, elevs[0], p, with_fp)
            if not args.no_cpu_all_cores:  # single thread AND OpenMP over rows on all host cores
                out["cpu_baseline_all_cores"] = cpu_baseline(args, elevs[0], p, with_fp, threads=0)  # (0: the fastest rung of a thread ladder)
        if check is not None:
            out["parity_check"] = check
        print(json.dumps(out), flush=True)
        if check is not None and not check["ok"]:
            print("parity check FAILED: " + json.dumps(check), file=sys.stderr)
            ctx.close()
            sys.exit(1)
    ctx.close()
    if world > 1:
        barrier()  # rank 0 is still timing the CPU baseline: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
