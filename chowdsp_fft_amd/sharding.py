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


def _host_backend() -> bool:
    """True when the process group moves host memory only (gloo: the CPU tests, and the one-box rehearsal of bench.py's
    N > 1 branch): device tensors are then staged through the host around every collective."""
    import torch.distributed as dist

    return dist.get_backend() == "gloo"


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (elapsed time): the job is as slow as its slowest rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if _host_backend():
        device = None
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if _host_backend():
        device = None
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
    if _host_backend() and local.device.type != "cpu":
        local = local.cpu()  # gloo rehearsal on a GPU box: the collective runs on host copies (RCCL gathers device memory)
    width = max(counts)
    if local.shape[0] != width:
        pad = torch.zeros((width - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(c == width for c in counts):
        return out
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)], dim=0)


def describe_ranks(description: str):
    """Every rank's own description (device name, XCC count ...), in rank order, on every rank: the record of WHICH
    devices the process group really spans."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [description]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, description)
    return [str(d) for d in out]


def timed_gather(local, total_frames: int, device=None, synchronize=None) -> dict:
    """The one collective the north star names -- the final collection of the result shards (all-gather) -- timed on its
    own, bracketed by barriers, MAX over ranks.  `synchronize`: waits for the device (torch.cuda.synchronize on GPUs,
    nothing on CPU tensors).  Never part of the throughput `value`."""
    import time

    import torch.distributed as dist

    sync = synchronize or (lambda: None)
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    full = gather_frames(local, total_frames)
    sync()
    if world > 1:
        dist.barrier()
    seconds = max_over_ranks(time.perf_counter() - t0, device=device)
    nbytes = int(full.numel()) * int(full.element_size())
    received = nbytes * (world - 1) // world
    ok = int(full.shape[0]) == int(total_frames)
    del full
    return {"collective": "all_gather_into_tensor", "seconds": seconds, "bytes_gathered": nbytes,
            "bytes_received_per_gpu": received, "GB_s_per_gpu": (received / seconds / 1e9) if seconds > 0 else None,
            "frames_gathered": int(total_frames), "shape_ok": ok}


def multi_gpu_fields(world: int, rank_descriptions, headline_gather, extra_lines=None) -> dict:
    """What an N > 1 bench line carries besides the N = 1 fields (bench.py; driven by tests/test_sharding_gloo.py with
    CPU tensors): how many ranks the process group has, which devices they are, the gather of the headline results
    (`gather_xgmi`) and, under `multi_gpu`, the other sharded configurations with a gather of their own."""
    import torch.distributed as dist

    seen = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    fields = {"ranks_seen": int(seen), "ranks_expected": int(world), "rank_devices": list(rank_descriptions)}
    if headline_gather is not None:
        fields["gather_xgmi"] = headline_gather
    if extra_lines:
        fields["multi_gpu"] = extra_lines
    return fields
