"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising exactly the host logic
bench.py uses on GPUs -- contiguous batch sharding, barrier + MAX-over-ranks timing reduction, and the
final gather.  The per-rank transform is played by the oracle here (test infrastructure; the HIP kernels
need a GPU), so what is asserted is that sharded == unsharded, frame for frame, in the original order.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from chowdsp_fft_amd.sharding import (describe_ranks, gather_frames, max_over_ranks, multi_gpu_fields, shard_range,  # noqa: E402
                                      sum_over_ranks, timed_gather)


def test_shard_range_covers_batch_exactly_once():
    for total in (0, 1, 7, 8, 65536, 1048576, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert spans[-1][0] + spans[-1][1] == total
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_frames, n, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as ol

        oracle = ol.Oracle()
        s = oracle.new_setup(n, ol.COMPLEX)
        full = ol.uniform(total_frames * 2 * n, 1234).reshape(total_frames, 2 * n)  # same seeded batch on every rank
        start, count = shard_range(total_frames, world, rank)
        local = oracle.transform_batch(s, full[start:start + count], ol.FORWARD) if count else np.zeros((0, 2 * n), np.float32)
        dist.barrier()
        # timing reduction: the slowest rank defines the step time
        t = max_over_ranks(1.0 + rank)
        assert t == float(world)
        assert sum_over_ranks(count) == float(total_frames)
        gathered = gather_frames(torch.from_numpy(local), total_frames).numpy()
        want = oracle.transform_batch(s, full, ol.FORWARD)
        assert gathered.shape == want.shape
        assert np.array_equal(gathered, want)
        # the N > 1 branch of bench.py, same functions, CPU tensors: the timed gather (on by default there), the record of
        # which devices the process group spans, and the fields the JSON line gains
        info = timed_gather(torch.from_numpy(local), total_frames)
        assert info["collective"] == "all_gather_into_tensor" and info["shape_ok"] and info["seconds"] > 0
        assert info["frames_gathered"] == total_frames
        assert info["bytes_gathered"] == total_frames * 2 * n * 4
        assert info["bytes_received_per_gpu"] == info["bytes_gathered"] * (world - 1) // world
        ranks = describe_ranks(f"rank {rank}: cpu pid {os.getpid()}")
        assert len(ranks) == world and all(ranks[r].startswith(f"rank {r}:") for r in range(world))
        assert len({d.split("pid")[1] for d in ranks}) == world  # really `world` different processes
        conv = {"conv_real_n4096_b131072": {"value": 1.0, "gather_xgmi": info}}
        fields = multi_gpu_fields(world, ranks, info, conv)
        assert fields["ranks_seen"] == world == fields["ranks_expected"]
        assert fields["rank_devices"] == ranks
        assert fields["gather_xgmi"]["GB_s_per_gpu"] > 0
        assert fields["multi_gpu"]["conv_real_n4096_b131072"]["gather_xgmi"]["frames_gathered"] == total_frames
        import json

        json.dumps(fields)  # what bench.py prints must serialise
        oracle.destroy_setup(s)
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total_frames", [(2, 8), (2, 7), (3, 10)])
def test_sharded_equals_unsharded_gloo(tmp_path, world, total_frames):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total_frames, 256, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]
