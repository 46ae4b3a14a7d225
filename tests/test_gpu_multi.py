"""The batch axis across several contexts / ranks through libtravgpu (on a one-GPU box they share device 0): the C-ABI's
single-process entry points (te_bcast_params, te_run_chain_multi, te_sync_multi, te_shard_range) and two gloo ranks
driving their shards; both must reproduce the single-context result map by map."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.helpers import OUT_LAYERS, compare_layer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def _params(capi):
    return capi.default_params(normals_radius=0.21, rough_radius=0.21, step_radius1=0.16, step_radius2=0.11, step_ncrit=3,
                               slope_critical=0.8, fp_radius=0.2, fp_offset=0.1)


def test_single_process_contexts_share_a_batch(capi):
    from traversability_estimation_amd import synth
    rows, cols, res, B, n_ctx = 160, 128, 0.05, 7, 3
    elevs = np.stack([synth.perlin_elevation(rows, cols, seed=2000 + b) for b in range(B)])
    per = rows * cols
    with capi.Context(0) as one:
        one.set_params(_params(capi))
        one.set_geometry(rows, cols, B, res)
        one.upload_elevation(elevs)
        one.run_chain(capi.RUN_FOOTPRINT)
        one.sync()
        want = {k: one.download(k) for k in OUT_LAYERS + ("traversability_footprint",)}
    n_dev = capi.device_count()
    ctxs = [capi.Context(k % n_dev) for k in range(n_ctx)]
    try:
        ctxs[1].set_params(_params(capi))  # the root decides; the others start from the defaults
        capi.bcast_params(ctxs, root=1)
        for c in ctxs:
            assert capi.params_to_bytes(c.get_params()) == capi.params_to_bytes(ctxs[1].get_params())
        spans = [capi.shard_range(B, n_ctx, k) for k in range(n_ctx)]
        assert sum(n for _, n in spans) == B
        for c, (first, count) in zip(ctxs, spans):
            c.set_geometry(rows, cols, count, res)
            c.upload_elevation(elevs[first:first + count])
        capi.run_chain_multi(ctxs, capi.RUN_FOOTPRINT)
        capi.sync_multi(ctxs)
        for c, (first, count) in zip(ctxs, spans):
            for k in want:
                # (a shard is cut into other strips than the whole batch, so the sliding sums are rounded differently:
                # the same tolerance as against the oracle, not bit equality)
                n_bad, mx, _ = compare_layer(k, c.download(k), want[k][first * per:(first + count) * per], tol=2e-6)
                assert n_bad == 0, (k, first, n_bad, mx)
    finally:
        for c in ctxs:
            c.close()


WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["TE_ROOT"])
import torch
from traversability_estimation_amd import capi, dist, synth
rank, world, local_rank = dist.init_process_group("gloo")
assert world == 2
capi.load()
p = capi.default_params(normals_radius=0.21 if rank == 0 else 0.4, rough_radius=0.21, step_radius1=0.16, step_radius2=0.11,
                        step_ncrit=3 if rank == 0 else 7)
p = dist.broadcast_params(capi, p, src=0)
rows, cols, res, n_maps = 160, 128, 0.05, 5
a, b = dist.shard_range(n_maps, rank, world)
with capi.Context(local_rank % max(1, torch.cuda.device_count())) as ctx:
    ctx.set_params(p)
    ctx.set_geometry(rows, cols, b - a, res)
    ctx.upload_elevation(np.stack([synth.perlin_elevation(rows, cols, seed=2000 + m) for m in range(a, b)]))
    ctx.run_chain(0)
    ctx.sync()
    local = ctx.download("traversability").reshape(b - a, rows * cols)
full = dist.gather_shards(local, n_maps)
dist.barrier()
if rank == 0:
    np.save(os.environ["TE_OUT"], full)
'''


def test_two_gloo_ranks_drive_their_shards_through_libtravgpu(capi, tmp_path):
    from traversability_estimation_amd import synth
    out = tmp_path / "full.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, TE_ROOT=ROOT, TE_OUT=str(out), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    full = np.load(out)
    rows, cols, res, n_maps = 160, 128, 0.05, 5
    with capi.Context(0) as one:
        one.set_params(capi.default_params(normals_radius=0.21, rough_radius=0.21, step_radius1=0.16, step_radius2=0.11, step_ncrit=3))
        one.set_geometry(rows, cols, n_maps, res)
        one.upload_elevation(np.stack([synth.perlin_elevation(rows, cols, seed=2000 + m) for m in range(n_maps)]))
        one.run_chain(0)
        one.sync()
        want = one.download("traversability").reshape(n_maps, rows * cols)
    n_bad, mx, _ = compare_layer("traversability", full, want, tol=2e-6)
    assert n_bad == 0, (n_bad, mx)


def test_bcast_params_across_devices_over_rccl(capi):
    """te_bcast_params with contexts on different devices takes the RCCL branch (ncclCommInitAll + grouped ncclBroadcast
    on streams of the call's own).  Needs >= 2 GPUs: on the one-GPU boxes of the test pool this is skipped, on a
    multi-GPU node it is the first thing to run there."""
    n_dev = capi.device_count()
    if n_dev < 2:
        pytest.skip("one GPU: the RCCL branch of te_bcast_params needs two devices")
    n_dev = min(n_dev, 8)
    ctxs = [capi.Context(d) for d in range(n_dev)] + [capi.Context(0)]  # the last one shares the root's device (host copy)
    try:
        p = capi.default_params(slope_critical=0.7, step_ncrit=7, fp_radius=0.21, w_slope=0.5)
        ctxs[0].set_params(p)
        capi.bcast_params(ctxs, root=0)
        want = bytes(memoryview(ctxs[0].get_params()))
        for c in ctxs[1:]:
            assert bytes(memoryview(c.get_params())) == want
        # and from a root that is not context 0
        p2 = capi.default_params(rough_critical=0.09)
        ctxs[n_dev - 1].set_params(p2)
        capi.bcast_params(ctxs, root=n_dev - 1)
        want = bytes(memoryview(ctxs[n_dev - 1].get_params()))
        for c in ctxs:
            assert bytes(memoryview(c.get_params())) == want
    finally:
        for c in ctxs:
            c.close()


def test_bcast_params_over_rccl_on_one_device(capi):
    """TE_OPT_BCAST_RCCL on the root: te_bcast_params takes its RCCL branch although every context shares one device --
    librccl is found at run time, a communicator of one rank is created, the 144-byte block goes host -> device -> grouped
    ncclBroadcast -> host, and the other contexts are given what came back.  (Between devices the same calls run with one
    rank per device: test_bcast_params_across_devices_over_rccl, which needs two GPUs.)  In a process of its own: librccl
    prints its version banner to the C stdout of whoever initialises it, behind pytest's summary line."""
    code = (
        "from traversability_estimation_amd import capi\n"
        "capi.load()\n"
        "ctxs = [capi.Context(0) for _ in range(3)]\n"
        "p = capi.default_params(slope_critical=0.65, step_ncrit=9, fp_radius=0.23, w_rough=0.25)\n"
        "ctxs[1].set_params(p)\n"
        "ctxs[1].set_option(capi.OPT_BCAST_RCCL, 1)\n"
        "for _ in range(2):\n"  # (a communicator per call: the second call builds another)
        "    capi.bcast_params(ctxs, root=1)\n"
        "want = bytes(memoryview(ctxs[1].get_params()))\n"
        "assert want == bytes(memoryview(p))\n"
        "assert all(bytes(memoryview(c.get_params())) == want for c in ctxs)\n"
        # the default takes the host copy between contexts of one device: same result
        "ctxs[1].set_option(capi.OPT_BCAST_RCCL, 0)\n"
        "ctxs[1].set_params(capi.default_params(rough_critical=0.07))\n"
        "capi.bcast_params(ctxs, root=1)\n"
        "assert all(bytes(memoryview(c.get_params())) == bytes(memoryview(ctxs[1].get_params())) for c in ctxs)\n"
        "[c.close() for c in ctxs]\n"
        "print('rccl-one-device ok')\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-one-device ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_gpus_2_without_a_launcher_runs_two_ranks(capi):
    """`python bench.py --gpus 2` with no torchrun around it (the driver's plain form) must start two ranks itself and say so
    in the line; on a one-GPU box the two gloo ranks share device 0 (TE_DIST_BACKEND=gloo), on a node with >= 2 GPUs the
    same command runs over RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    rccl = capi.device_count() >= 2
    if not rccl:
        env["TE_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--check-crops"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["world"] == 2
    assert line["ranks"]["backend"] == ("nccl" if rccl else "gloo")
    assert line["ranks"]["devices"] == ([0, 1] if rccl else [0, 0])
    assert line["scaling"] == "weak" and line["parity_check"]["ok"]
    # two maps of 4096^2 went through the chain per step
    assert abs(line["value"] - 2 * 4096 * 4096 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-6 * line["value"]


def test_bench_under_the_drivers_launcher_over_rccl_with_one_rank(capi):
    """The driver's N > 1 form -- `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- with N = 1, the
    only world a one-GPU box admits over RCCL, and TE_DIST_WORLD1_COLLECTIVES=1 so that the rank does not take the
    world-of-one shortcuts: backend "nccl" initialises on its device, the barrier, the MAX over ranks, the gather of the
    device indices and the parameter broadcast all go through RCCL on device tensors, and the line says so."""
    import json
    from traversability_estimation_amd import dist as tdist
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TE_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TE_DIST_WORLD1_COLLECTIVES"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(tdist.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--check-crops", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks"] == {"world": 1, "backend": "nccl", "devices": [0]}
    assert line["parity_check"]["ok"]
    # the JSON line is the LAST thing on stdout: librccl's version banner (NCCL_DEBUG=VERSION on these boxes) went out when
    # the communicator came up (dist.init_process_group flushes the C stdout), not at exit
    assert [ln for ln in r.stdout.splitlines() if ln.strip()][-1].startswith("{"), r.stdout[-2000:]
