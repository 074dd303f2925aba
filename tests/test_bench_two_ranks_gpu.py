"""bench.py's N > 1 branch, executed end to end: two ranks under torch.distributed.run on ONE box.

The driver launches the multi-GPU bench as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`;
a lease holds one GPU, so the branch (process group, barrier + MAX-over-ranks timing, the secondary configurations
under a process group, the record of the ranks' devices, the final gather and the `multi_gpu` fields) is rehearsed
with `--dist-backend gloo`: both ranks compute on device 0 through the HIP library, the collectives move host
copies.  What it proves is that the branch runs and that its one JSON line is complete -- NOT a scaling figure
(north star: "1/2/4/8 GPUs ... RCCL only for the final gather"; replicated read-only setups:
/root/reference/chowdsp_fft.h:87-91).

The file name sorts before test_c_program_gpu.py and nothing here makes a HIP call in the pytest process: the
launcher is started as a fresh child (a process that has initialised the GPU must not fork + exec on the GPU pool).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(extra, timeout):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--dist-backend", "gloo"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_two_ranks_without_a_gpu_fails_loudly():
    """CPU leg: the launcher starts both ranks, each reports the missing device and exits 3 (no CPU fallback exists)"""
    import torch

    if torch.cuda.device_count() >= 1:
        pytest.skip("a GPU is visible: the gpu-marked test below runs the branch for real")
    r = _launch(["--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=600)
    assert r.returncode != 0
    assert "no GPU visible" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_box():
    r = _launch(["--frames-per-gpu", "2048", "--secondary-frames-per-gpu", "1024", "--steps", "3", "--warmup", "1",
                 "--prewarm-seconds", "0.05"], timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0), got {len(lines)}:\n{r.stdout[-2000:]}"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["ranks_seen"] == 2 == line["ranks_expected"]
    assert line["dist_backend"] == "gloo" and "rehearsal" in line
    devs = line["rank_devices"]
    assert len(devs) == 2 and devs[0].startswith("rank 0:") and devs[1].startswith("rank 1:")
    assert len({d.split("pid")[1] for d in devs}) == 2, "two distinct processes"
    assert line["config"]["frames_per_gpu"] == 2048 and line["config"]["global_frames"] == 4096
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1
    # the final gather of the headline results and of BASELINE configs[3], each timed on its own
    g = line["gather_xgmi"]
    assert g["shape_ok"] and g["frames_gathered"] == 4096 and g["collective"] == "all_gather_into_tensor"
    conv = line["multi_gpu"]["conv_real_n4096_b131072"]
    assert conv["value"] > 0 and conv["global_frames"] == 2048 and conv["gather_xgmi"]["shape_ok"]
    # every secondary configuration ran under the process group
    sys.path.insert(0, ROOT)
    import bench

    got = [s["workload"] for s in line["secondary"]]
    assert got == list(bench.SECONDARY_WORKLOADS)
    for s in line["secondary"]:
        assert s["value"] > 0 and s["frames_per_gpu"] == 1024 and 0 < s["roofline"]["frac"] < 1
    assert "cpu_baseline" not in line  # rank 0 times the CPU baseline at N = 1 only


@pytest.mark.gpu
def test_rccl_one_rank_smoke():
    """the "nccl" (= RCCL) calls of the N > 1 branch with world size 1 on this box's GPU: communicator bound to the device, all-reduce,
    all_gather_into_tensor, all_gather_object, barrier (tools/nccl_one_rank_smoke.py) -- started as a fresh child, like the test above"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_one_rank_smoke.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "all_reduce 1.5" in r.stdout and "all_gather ok True" in r.stdout and r.stdout.strip().endswith("done"), r.stdout[-1000:]
