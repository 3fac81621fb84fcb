"""Multi-GPU sharding of the env batch.  Environments are independent, so the
path shards with NO data-path collective: rank r owns the global env ids
[r*B, (r+1)*B) (its devices are seeded by global id, so the union over ranks is
the same set of devices a single process would simulate).  torch.distributed is
used only to line the ranks up and to take the max of the elapsed time."""
from __future__ import annotations

import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_env_ids(rank: int, world: int, envs_per_rank: int):
    """(first global env id, count) owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * envs_per_rank, envs_per_rank


def init(backend: str, local_rank: int = 0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist.init_process_group(backend)
    return dist


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (elapsed seconds); identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
