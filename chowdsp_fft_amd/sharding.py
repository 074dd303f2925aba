"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; "nccl" = RCCL).

Every frame is an independent transform / convolution: the batch is split into
contiguous blocks, one per rank, plans (twiddle tables, filter spectrum) are replicated, and there is NO
collective on the data path.  The only optional exchange is the final collection of results
(`gather_frames`, an all-gather over xGMI); it is reported separately from compute throughput.
The same functions run on CPU tensors with the gloo backend (tests/test_sharding_gloo.py).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(total_frames: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first frame, frame count) of `rank`'s contiguous block; blocks differ by at most one frame and
    cover [0, total_frames) exactly once."""
    if world_size < 1 or not (0 <= rank < world_size) or total_frames < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(total_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (elapsed time): the job is as slow as its slowest rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_frames(local, total_frames: int):
    """All-gather the per-rank result blocks into the full [total_frames, frame] tensor on every rank.
    Uneven blocks are padded to the largest block for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(total_frames, world, r)[1] for r in range(world)]
    assert local.shape[0] == counts[rank]
    width = max(counts)
    if local.shape[0] != width:
        pad = torch.zeros((width - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(c == width for c in counts):
        return out
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)], dim=0)
