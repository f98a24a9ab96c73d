"""Episode data-parallelism: every episode's create_tasks and every sample_actions row
depends only on that episode (scripts/train.py:453-454 vmaps per sample), so ranks own disjoint episode
ranges and the data path has NO collective.  torch.distributed (RCCL on the GPU box, gloo in CPU tests)
is used only to line ranks up and to take the max of their timings."""
from __future__ import annotations

from typing import Tuple


def episode_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rank r owns episodes [start, end): contiguous, disjoint, covering, sizes differ by at most 1."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError((total, world, rank))
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_over_ranks(seconds: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank: int, world: int, steps: int, max_seconds: float) -> float:
    """bench.py's `value`: units all ranks processed / the slowest rank's time."""
    return world * units_per_rank * steps / max_seconds
