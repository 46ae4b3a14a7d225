"""A batch whose layers exceed 4 GiB: 4200 maps of 512 x 512 (1.10e9 cells, 4.4 GB per layer, a 63 GB slab) in ONE launch --
the batch axis sized for this device's memory rather than for BASELINE's 512 maps.  Map offsets are 64-bit, offsets within a
map 32-bit: maps on both sides of the 2^31- and 2^32-byte marks against the oracle (chain + footprint), and a map uploaded
into two slots gives the same bits in both.        python tools/dbg/large_batch.py [maps]"""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tests.helpers import OUT_LAYERS, assert_layers_match
from tests.test_gpu_fullsize import bench_params, oracle_params
from traversability_estimation_amd import capi, synth
from oracle import oracle

capi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4200
rows = cols = 512
res = 0.05
per = rows * cols
ALL = OUT_LAYERS + ("traversability_footprint",)
base = synth.perlin_elevation(rows, cols, seed=2000).reshape(cols, rows)
p = bench_params(capi, synth, 5, res)
op = oracle_params(oracle, p)
picks = sorted({0, 1, 2047, 2048, 4095, 4096, B - 1} & set(range(B)))


def the_map(b):  # the MPC-rollout shape: base map + N(0, 1 cm), seed = map index (cheap: only for the maps looked at)
    return (base + np.random.default_rng(2000 + b).normal(0.0, 0.01, size=base.shape).astype(np.float32)).astype(np.float32)


kept = {b: the_map(b) for b in picks}
with capi.Context(0) as ctx:
    ctx.set_params(p)
    ctx.set_geometry(rows, cols, B, res)
    t0 = time.time()
    blk = np.empty((64, cols, rows), np.float32)
    for b0 in range(0, B, 64):  # blocks of 64 maps: the other maps are the base map under a per-map tilt
        nb = min(64, B - b0)
        for k in range(nb):
            b = b0 + k
            blk[k] = kept[b] if b in kept else base + np.float32(1e-4 * (b % 97)) * np.arange(rows, dtype=np.float32)[None, :]
        ctx.upload_elevation(blk[:nb], map0=b0)
    ctx.upload_elevation(kept[2047][None], map0=4100 if B > 4100 else 5)  # the same map in a second slot
    ctx.run_chain(capi.RUN_FOOTPRINT)
    ctx.sync()
    print("upload + first launch", round(time.time() - t0, 2), "s", flush=True)
    t0 = time.time()
    for _ in range(3):
        ctx.run_chain(capi.RUN_FOOTPRINT)
    ctx.sync()
    dt = (time.time() - t0) / 3
    print(f"launch {dt * 1e3:.2f} ms: {B * per / dt:.3e} cells/s, {B * per * 24 / dt / 1e9:.0f} GB/s algorithmic", flush=True)
    oracle.set_threads(8)
    g = oracle.geom(rows, cols, res)
    twin = 4100 if B > 4100 else 5
    for k in ALL:
        a = ctx.download_tile(k, 2047, 0, 0, rows, cols) if 2047 < B else None
        t = ctx.download_tile(k, twin, 0, 0, rows, cols)
        if a is not None:
            assert np.array_equal(a.view(np.uint32), t.view(np.uint32)), ("the same map in two slots", k)
    for b in picks:
        want = oracle.chain(g, op, kept[b])
        want["traversability_footprint"] = oracle.footprint(g, op, kept[b], want)
        got = {k: ctx.download_tile(k, b, 0, 0, rows, cols).reshape(-1) for k in ALL}
        assert_layers_match(got, want, layers=ALL, ctx=f"map {b} of {B}")
        print(f"map {b} (byte offset {4 * b * per:#x}): checked", flush=True)
print("ok")
