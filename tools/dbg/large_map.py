"""A map of 2^28 / 2^30 cells (16384^2 / 32768^2; layers of 1 / 4 GiB: byte offsets beyond 2^31 and 2^32) through the chain
and the footprint pass, crops against the oracle -- at the corners, in the middle and around the columns whose byte offsets
cross 2^30, 2^31 and 2^32.      python tools/dbg/large_map.py 16384 [cells]"""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tests.helpers import OUT_LAYERS, compare_layer
from tests.test_gpu_fullsize import bench_params, oracle_params
from traversability_estimation_amd import capi, synth
from oracle import oracle

capi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 9
res = 0.0625
ALL = OUT_LAYERS + ("traversability_footprint",)
t0 = time.time()
a = synth.perlin_elevation(2048, 2048, seed=77).reshape(2048, 2048)
blk = np.block([[a, a[:, ::-1]], [a[::-1, :], a[::-1, ::-1]]])
elev = np.tile(blk, (n // 4096, n // 4096))
elev += np.linspace(0.0, 3.0, n, dtype=np.float32)[None, :]
elev += np.linspace(0.0, 2.0, n, dtype=np.float32)[:, None]
print("elevation", elev.shape, elev.dtype, round(time.time() - t0, 1), "s", flush=True)
p = bench_params(capi, synth, cells, res)
op = oracle_params(oracle, p)
margin = 2 * cells + 3 + 9 + 12
oracle.set_threads(8)
with capi.Context(0) as ctx:
    ctx.set_params(p)
    ctx.set_geometry(n, n, 1, res)
    t0 = time.time()
    ctx.upload_elevation(elev)
    ctx.run_chain(capi.RUN_FOOTPRINT)
    ctx.sync()
    print("upload + first launch", round(time.time() - t0, 2), "s", flush=True)
    t0 = time.time()
    for _ in range(3):
        ctx.run_chain(capi.RUN_FOOTPRINT)
    ctx.sync()
    dt = (time.time() - t0) / 3
    print(f"launch {dt * 1e3:.2f} ms: {n * n / dt:.3e} cells/s, {n * n * 24 / dt / 1e9:.0f} GB/s algorithmic", flush=True)
    size = 192
    js = sorted({0, n - size, n // 2 - size // 2, n // 4 - size // 2, (1 << 30) // (4 * n) - size // 2, (1 << 31) // (4 * n) - size // 2,
                 min(n - size, (1 << 32) // (4 * n) - size // 2)})
    js = [j for j in js if 0 <= j <= n - size]
    bad_total = 0
    for j0 in js:
        for i0 in (0, n // 2 - 77, n - size):
            i_lo, i_hi = max(0, i0 - margin), min(n, i0 + size + margin)
            j_lo, j_hi = max(0, j0 - margin), min(n, j0 + size + margin)
            crop = np.ascontiguousarray(elev[j_lo:j_hi, i_lo:i_hi])
            g = oracle.geom(i_hi - i_lo, j_hi - j_lo, res)
            want = oracle.chain(g, op, crop)
            want["traversability_footprint"] = oracle.footprint(g, op, crop, want)
            ki = slice(0 if i_lo == 0 else margin, (i_hi - i_lo) if i_hi == n else (i_hi - i_lo) - margin)
            kj = slice(0 if j_lo == 0 else margin, (j_hi - j_lo) if j_hi == n else (j_hi - j_lo) - margin)
            for k in ALL:
                got = ctx.download_tile(k, 0, i_lo, j_lo, i_hi - i_lo, j_hi - j_lo)[kj, ki]
                b = want[k].reshape(j_hi - j_lo, i_hi - i_lo)[kj, ki]
                n_bad, mx, _ = compare_layer(k, got, b)
                bad_total += n_bad
                if n_bad:
                    print(f"  MISMATCH crop i {i_lo}..{i_hi} j {j_lo}..{j_hi} layer {k}: {n_bad} cells, max {mx:.3g}", flush=True)
        print(f"columns {j0}..{j0 + size} (byte offset of the first {4 * j0 * n:#x}): checked", flush=True)
print("mismatching cells:", bad_total)
sys.exit(1 if bad_total else 0)
