"""Data parallelism over RCCL / xGMI -- functionality the reference does not have.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  Every rank holds full
replicas of the networks and Adam state, replays the SAME host RNG protocol (so the global index
batch and noise are identical everywhere), computes rows [rank*B/W, (rank+1)*B/W) of the global
batch with 1/B_global loss scaling, and sums one flat gradient bucket per optimizer
(D: 314 401 + pad, G: 322 784 + pad fp32) with a single all-reduce each -- two collectives per D+G
step.  N ranks therefore reproduce the 1-rank run up to fp32 summation order.
"""
import os

import torch


def shard_range(B, world, rank):
    """Rows of the global batch owned by `rank` (B must divide evenly: per-rank work is fixed)."""
    if B % world != 0:
        raise ValueError("global batch %d does not divide across %d ranks" % (B, world))
    bl = B // world
    return rank * bl, (rank + 1) * bl


def env_world():
    """(world_size, rank, local_rank) from the torchrun environment."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (MASTER_ADDR defaults to 127.0.0.1)."""
    import torch.distributed as dist
    world, rank, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank, local_rank


def current():
    """(world, rank, group) of the initialised default group, or (1, 0, None)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank(), None
    return 1, 0, None


def allreduce_sum_(flat, group=None):
    """In-place SUM all-reduce of one flat bucket on the current stream (RCCL on device tensors,
    gloo on CPU tensors in the tests)."""
    import torch.distributed as dist
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat
