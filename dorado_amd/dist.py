"""Multi-GPU plumbing: one process per GPU, replicas only — the path has no exchange step
(SURVEY.md §8e), so the only cross-rank traffic is a barrier and one MAX-reduced scalar for the
benchmark clock.  Works with backend "nccl" (RCCL, on GPUs) and "gloo" (CPU tests)."""
from __future__ import annotations

import os


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one and the
    shards partition range(n_items) exactly (reads are independent units)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device=None) -> float:
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
