"""world_size-2 test of the multi-GPU plumbing on CPU (gloo): parameter broadcast from rank 0, batch-axis
sharding with no data-path collective, max-over-ranks timing.  The per-rank compute is the CPU oracle
here (no GPU in this container); on the GPU box the same code paths drive libtravgpu via bench.py."""
import os
import subprocess
import sys

import numpy as np

from tests.conftest import ROOT

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["TE_ROOT"])
from traversability_estimation_amd import capi, dist, synth
from oracle import oracle as O
rank, world, _ = dist.init_process_group("gloo")
assert world == 2
# rank 0 decides the parameters; rank 1 starts from different ones and must end up with rank 0's
p = capi.default_params(normals_radius=0.11 if rank == 0 else 0.5, step_ncrit=3 if rank == 0 else 9,
                        w_scale=0.25 if rank == 0 else 0.75)
p = dist.broadcast_params(capi, p, src=0)
assert abs(p.normals_radius - 0.11) < 1e-15 and p.step_ncrit == 3 and p.w_scale == 0.25, (rank, p.normals_radius)
# shard a batch of 5 maps: 3 + 2
n_maps, rows, cols, res = 5, 48, 40, 0.05
a, b = dist.shard_range(n_maps, rank, world)
assert (a, b) == ((0, 3) if rank == 0 else (3, 5))
op = O.default_params()
for f, _ in op._fields_:
    setattr(op, f, getattr(p, f))
g = O.geom(rows, cols, res)
local = np.stack([O.chain(g, op, synth.perlin_elevation(rows, cols, seed=2000 + m))["traversability"] for m in range(a, b)])
full = dist.gather_shards(local, n_maps)
t = dist.max_over_ranks(1.0 + rank)
assert t == 2.0
dist.barrier()
if rank == 0:
    np.save(os.environ["TE_OUT"], full)
'''


def test_two_rank_broadcast_and_batch_sharding(tmp_path):
    out = tmp_path / "full.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, TE_ROOT=ROOT, TE_OUT=str(out), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    full = np.load(out)
    # the sharded result equals the single-process one, map by map
    from oracle import oracle as O
    from traversability_estimation_amd import capi, synth
    p = capi.default_params(normals_radius=0.11, step_ncrit=3, w_scale=0.25)
    op = O.default_params()
    for f, _ in op._fields_:
        setattr(op, f, getattr(p, f))
    g = O.geom(48, 40, 0.05)
    for m in range(5):
        want = O.chain(g, op, synth.perlin_elevation(48, 40, seed=2000 + m))["traversability"]
        assert (full[m].view(np.uint32) == want.view(np.uint32)).all()


def test_shard_ranges_partition_the_batch():
    from traversability_estimation_amd import dist
    for n in (1, 5, 8, 512, 513):
        for world in (1, 2, 4, 8):
            spans = [dist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[k][1] == spans[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


LAUNCHED = r'''
import json, os, sys
sys.path.insert(0, os.environ["TE_ROOT"])
from traversability_estimation_amd import dist
if "WORLD_SIZE" not in os.environ:  # first entry: no launcher -> start the ranks, as bench.py --gpus N does
    sys.exit(dist.relaunch_under_torchrun(int(sys.argv[1]), [os.path.abspath(__file__)] + sys.argv[1:]))
rank, world, local_rank = dist.init_process_group("gloo")
rep = dist.ranks_report(10 + local_rank)
if rank == 0:
    open(os.environ["TE_OUT"], "w").write(json.dumps(rep))
dist.barrier()
'''


def test_a_script_without_a_launcher_starts_its_own_ranks(tmp_path):
    """bench.py --gpus N under plain `python`: N processes must run and the report must say so."""
    import json
    out = tmp_path / "ranks.json"
    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TE_ROOT=ROOT, TE_OUT=str(out))
    r = subprocess.run([sys.executable, str(script), "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads(out.read_text())
    assert rep == {"world": 3, "backend": "gloo", "devices": [10, 11, 12]}


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """A launcher that started another number of ranks than --gpus says is an error, never a silent 1-rank figure."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", TE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
    # and RCCL ranks need a device each: two ranks on a node with fewer GPUs is refused before anything is measured
    env = {k: v for k, v in os.environ.items() if k not in ("TE_DIST_BACKEND",)}
    env.update(WORLD_SIZE="64", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "one device per rank" in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]


def test_a_single_rank_can_be_made_to_run_its_collectives():
    """TE_DIST_WORLD1_COLLECTIVES=1: world 1 initialises a process group and every helper runs its collective instead of the
    shortcut (here over gloo; tests/test_gpu_multi.py runs the same switch over RCCL on the GPU box)."""
    from traversability_estimation_amd import dist as tdist
    code = ("import numpy as np\n"
            "from traversability_estimation_amd import dist as d\n"
            "assert d.init_process_group('gloo') == (0, 1, 0)\n"
            "import torch.distributed as td\n"
            "assert td.is_initialized() and td.get_world_size() == 1\n"
            "assert d.ranks_report(3) == {'world': 1, 'backend': 'gloo', 'devices': [3]}\n"
            "assert d.max_over_ranks(2.5) == 2.5 and d.broadcast_blob(b'xyz') == b'xyz'\n"
            "d.barrier()\n"
            "assert d.gather_shards(np.ones((2, 3), np.float32), 2).shape == (2, 3)\n"
            "td.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(TE_DIST_WORLD1_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(tdist.free_port()), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # without the switch a single rank never touches torch.distributed
    code2 = ("from traversability_estimation_amd import dist as d\n"
             "assert d.init_process_group('gloo') == (0, 1, 0)\n"
             "import torch.distributed as td\n"
             "assert not td.is_initialized()\n"
             "assert d.ranks_report(0) == {'world': 1, 'backend': None, 'devices': [0]}\n")
    env.pop("TE_DIST_WORLD1_COLLECTIVES")
    r = subprocess.run([sys.executable, "-c", code2], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
