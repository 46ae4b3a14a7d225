"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" on
CPU for the tests).  The filter chain shards on the BATCH axis only -- maps are independent, so there is
no data-path collective: rank 0's filter parameters (the te_params blob, < 1 KB) are broadcast once at
configure time, and timings are max-reduced.  A single large map does not shard ("replicas only").
"""
import os

import numpy as np


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def _world1_collectives():
    """TE_DIST_WORLD1_COLLECTIVES=1: a single rank initialises its process group too and every helper below runs its
    collective instead of the world-of-one shortcut -- how a one-GPU box exercises the RCCL branch of each of them."""
    return os.environ.get("TE_DIST_WORLD1_COLLECTIVES") == "1"


def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _world1_collectives())


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def init_process_group(backend=None):
    """Returns (rank, world, local_rank).  No-op for world == 1 (but see _world1_collectives)."""
    rank, world, local_rank = env_world()
    if world > 1 or _world1_collectives():
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", str(rank))  # (a single rank without a launcher: _world1_collectives)
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if not dist.is_initialized():
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
                # librccl prints its version banner (NCCL_DEBUG=VERSION, exported on the GPU boxes) to the C stdout of every
                # rank when its communicator comes up; into a pipe that is block-buffered and would appear at exit, BEHIND
                # the one JSON line rank 0 prints.  The first collective brings the communicator up; flush it out now.
                dist.barrier()
                _flush_c_stdio()
            else:
                dist.init_process_group(backend)
    return rank, world, local_rank


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch_under_torchrun(n_ranks, argv):
    """Start `argv` (script + arguments) as n_ranks processes of ONE node under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1; returns the launcher's exit code.  What `python bench.py --gpus N` does when no launcher set
    WORLD_SIZE."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + list(argv)
    return subprocess.call(cmd, env=env)


def ranks_report(device_index):
    """{"world", "backend", "devices": [device index of every rank, gathered]} -- what the bench line carries so that a
    reader can see how many ranks ran and where."""
    import torch.distributed as dist
    if not _active():
        return {"world": 1, "backend": None, "devices": [int(device_index)]}
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, int(device_index))
    return {"world": dist.get_world_size(), "backend": str(dist.get_backend()), "devices": [int(v) for v in got]}


def _device():
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_blob(blob, src=0):
    """Broadcast a bytes object of identical length on every rank (the te_params struct)."""
    import torch
    import torch.distributed as dist
    if not _active():
        return bytes(blob)
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(_device())
    if dist.get_rank() != src:
        t.zero_()
    dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def broadcast_params(capi, p, src=0):
    """Rank `src` decides the filter parameters; every rank returns the same te_params."""
    return capi.params_from_bytes(broadcast_blob(capi.params_to_bytes(p), src))


def shard_range(n_maps, rank, world):
    """Contiguous block of the batch owned by `rank`: sizes differ by at most one map."""
    base, extra = divmod(n_maps, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_over_ranks(value):
    import torch
    import torch.distributed as dist
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if _active():
        dist.barrier()


def gather_shards(local, n_maps):
    """all_gather of per-rank [n_local, cells] float32 arrays back into the full batch (tests only)."""
    import torch
    import torch.distributed as dist
    if not _active():
        return np.asarray(local)
    world = dist.get_world_size()
    cells = local.shape[1]
    per = max(shard_range(n_maps, r, world)[1] - shard_range(n_maps, r, world)[0] for r in range(world))
    pad = np.full((per, cells), np.nan, np.float32)
    pad[:local.shape[0]] = local
    out = [torch.empty((per, cells), dtype=torch.float32, device=_device()) for _ in range(world)]
    dist.all_gather(out, torch.from_numpy(pad).to(_device()))
    parts = []
    for r in range(world):
        a, b = shard_range(n_maps, r, world)
        parts.append(out[r].cpu().numpy()[:b - a])
    return np.concatenate(parts, axis=0)
