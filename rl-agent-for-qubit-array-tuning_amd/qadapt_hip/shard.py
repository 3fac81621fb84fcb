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


def bucket_cost(n_dots: int, resolution: int = 64) -> float:
    """Relative cost of one env-step of an n-dot array: (n-1) channels of R*R pixels, each a 32-state
    eigenproblem plus a candidate search whose tree grows with n (measured: 8-dot pixels cost ~1.9x a 4-dot
    pixel).  Only ratios matter; used to balance mixed-N buckets over ranks."""
    return (n_dots - 1) * resolution * resolution * (0.55 + 0.18 * n_dots)


def shard_mixed(counts: dict, rank: int, world: int, resolution: int = 64):
    """BASELINE config 5: a ragged batch is bucketed by dot count and EVERY bucket is split over all
    ranks, so each GPU gets the same mix.  Global env ids run bucket after bucket (ascending n_dots); inside a
    bucket rank r owns a contiguous slice.  The remainder envs of a bucket (count % world) go, most expensive
    bucket first, to the ranks that are currently least loaded (cost-weighted), so no rank collects all the
    left-overs.  Returns {n_dots: (first_global_env_id, n_envs)} for `rank`; every global id is owned by
    exactly one rank (tests/test_shard.py)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    order = sorted(counts)
    base = {}
    off = 0
    for n in order:
        base[n] = off
        off += int(counts[n])
    share = {n: [int(counts[n]) // world] * world for n in order}
    load = [sum(share[n][r] * bucket_cost(n, resolution) for n in order) for r in range(world)]
    for n in sorted(order, key=lambda k: -bucket_cost(k, resolution)):
        rem = int(counts[n]) % world
        for r in sorted(range(world), key=lambda k: (load[k], k))[:rem]:      # one each, to the least loaded ranks
            share[n][r] += 1
            load[r] += bucket_cost(n, resolution)
    out = {}
    for n in order:
        first = base[n] + sum(share[n][:rank])
        out[n] = (first, share[n][rank])
    return out


def init(backend: str, local_rank: int = 0, timeout_s: float = 180.0):
    """Rendezvous of the ranks (used for the bench's barriers and the max of the elapsed time only).  The timeout is short
    on purpose: a rank that died before the rendezvous must not leave the others waiting for the backend's default half hour."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    to = datetime.timedelta(seconds=timeout_s)
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=to)
    else:
        dist.init_process_group(backend, timeout=to)
    return dist


def local_device_index(local_rank: int, visible_devices: int) -> int:
    """GPU a rank uses: its LOCAL_RANK when the process sees all GPUs of the node, device 0 when the launcher has
    already narrowed the visibility to one GPU per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)."""
    if visible_devices <= 0:
        raise RuntimeError("no GPU visible to this rank")
    return local_rank if local_rank < visible_devices else local_rank % visible_devices


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
