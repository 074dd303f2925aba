#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X, one JSON line on stdout (rank 0).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): GFFT/s, batched fp32 complex N=4096, plus achieved HBM GB/s vs the roofline.
Workload at every N: BASELINE configs[1] on EACH GPU -- N=4096 complex, batch 65536 frames per GPU,
one "step" = one FORWARD launch + one BACKWARD launch over the whole batch (fwd+inv), inputs resident
in HBM before the timed region.  Frames are independent, so N GPUs = N x the work, no data-path
collective ("weak" scaling); the per-rank elapsed time is reduced with MAX over ranks.

value        = transforms (2 per frame per step) x frames x GPUs / elapsed / 1e9
roofline     = algorithmic bytes of ONE launch (16*N bytes per frame: read 8N + write 8N) / average launch
               duration measured with HIP events on the launch stream over the timed region; peak 8000 GB/s
cpu_baseline = the REAL reference's AVX path (oracle/_ref, compiled in place from /root/reference) timed on
               this host by oracle/_ref/ref_bench; rank 0, --gpus 1 only; falls back to the plain-C oracle port.
Nothing here routes product work through oracle/: the oracle is only the CPU baseline being measured.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak

WORKLOADS = {
    # name: (N, transform, frames per GPU, description)
    "c2c_n4096_b65536_fwd_inv": (4096, "complex", 65536, "BASELINE configs[1]: N=4096 fp32 complex fwd+inv, batch=65536 per GPU"),
    "real_n1024_b262144_fwd_inv": (1024, "real", 262144, "BASELINE configs[2]: N=1024 fp32 real r2c+c2r, batch=262144 per GPU"),
    "c2c_n65536_b8192_fwd_inv": (65536, "complex", 8192, "BASELINE configs[4]: N=65536 fp32 complex (two passes in one launch, hand-off in the XCDs' L2: the library default), batch=8192 per GPU"),
    # the same configuration through the other two forms (option fused_two_pass): two launches per chunk (round 1), and the
    # ticket-queue one-launch form (round 2)
    "c2c_n65536_b8192_fwd_inv_two_launches": (65536, "complex", 8192, "BASELINE configs[4] as two launches per 256-frame chunk (fused_two_pass = 0)"),
    "c2c_n65536_b8192_fwd_inv_tickets": (65536, "complex", 8192, "BASELINE configs[4] through the superseded ticket-queue one-launch form (fused_two_pass = 1; EXPERIMENTS=1 builds only)"),
    "conv_real_n4096_b131072": (4096, "real", 131072, "BASELINE configs[3] per-GPU share: fused fwd->mul->inv frames, N=4096 real, 131072 frames per GPU"),
    # a mixed-radix size from the reference's "Other sizes" list (9216 = 2^10 * 3^2), generic kernel
    "c2c_n9216_b16384_fwd_inv": (9216, "complex", 16384, "mixed-radix N=9216 fp32 complex fwd+inv (generic kernel), batch=16384 per GPU"),
    # streaming partitioned overlap-save engine; one step = one block of 2048 samples on every channel
    "ols_engine_b2048_p16_c8192": (4096, "real", 8192, "partitioned overlap-save engine: block 2048 (N=4096 real), 16 partitions, 8192 channels per GPU"),
}


# kernels whose launches make up one library call of the workload (substrings of the demangled names): what the "kernel" field
# of the roofline object names, and what tools/pmc_refresh.py sums the PMC counters over
WORKLOAD_KERNELS = {
    "c2c_n4096_b65536_fwd_inv": ["c2c_kernel_pf"],
    "real_n1024_b262144_fwd_inv": ["r2c_kernel", "c2r_kernel"],
    "conv_real_n4096_b131072": ["conv_frames_real_tp_kernel", "conv_frames_real_kernel"],
    "c2c_n65536_b8192_fwd_inv": ["group256_kernel"],
    "c2c_n65536_b8192_fwd_inv_two_launches": ["pass_a256_kernel", "pass_b256_kernel"],
    "c2c_n65536_b8192_fwd_inv_tickets": ["fused256_kernel"],
    "c2c_n9216_b16384_fwd_inv": ["generic_inplace_kernel", "generic_kernel", "static_"],
    "ols_engine_b2048_p16_c8192": ["ols_block_kernel"],
}


def launches_per_step(workload: str) -> int:
    return 1 if (workload.startswith("conv_") or workload.startswith("ols_engine")) else 2


def committed_traffic(workload: str):
    """HBM / fabric bytes per launch from the PMC counters: NOT measured by a bench run (rocprofv3 --pmc needs passes of its
    own); read from profiles/pmc_traffic.json, which tools/pmc_refresh.sh writes together with a hash of the library sources
    it profiled.  Returns (traffic or None, extra fields): a figure taken on other sources than the ones being timed is
    withheld (`traffic` null, `traffic_stale` true) instead of being printed next to a kernel it does not describe."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None, {"traffic_source": None}
    try:
        from chowdsp_fft_amd import _buildinfo

        tj = json.load(open(tpath))
        cur_src, cur_lib = _buildinfo.source_sha256(), _buildinfo.lib_sha256()
        extra = {"traffic_source": tj.get("_taken_at"), "traffic_src_sha256": (tj.get("src_sha256") or "")[:16],
                 "timed_src_sha256": cur_src[:16], "traffic_lib_sha256_matches": tj.get("lib_sha256") == cur_lib}
        # stale = taken on other sources, OR the shared object being timed is not the one that was profiled (a stale build of the
        # right sources' predecessor must not get a figure printed beside it either; the pair of hashes says it all -- file
        # times do not survive every way a tree gets copied to a GPU box, so they are reported, not used)
        extra["lib_newer_than_sources"] = _buildinfo.lib_is_newer_than_sources()
        if tj.get("src_sha256") != cur_src or tj.get("lib_sha256") != cur_lib:
            extra["traffic_stale"] = True
            return None, extra
        entry = (tj.get("workloads") or {}).get(workload)
        if entry is None:
            return None, extra
        extra["traffic_kernels"] = entry.get("kernels")
        return entry.get("bytes_per_launch"), extra
    except Exception as e:  # a broken summary must not break the bench line
        return None, {"traffic_source": f"unreadable: {e}"}


DEFAULT_WORKLOAD = "c2c_n4096_b65536_fwd_inv"
MULTI_GPU_CONV_WORKLOAD = "conv_real_n4096_b131072"  # BASELINE configs[3]: 1M frames over 8 GPUs = 131072 per GPU
# the other BASELINE.json configurations, reported under "secondary" of the default run's ONE JSON line
SECONDARY_WORKLOADS = ("real_n1024_b262144_fwd_inv", "conv_real_n4096_b131072", "c2c_n65536_b8192_fwd_inv", "c2c_n65536_b8192_fwd_inv_two_launches")


def cpu_baseline(n: int, transform: str, seconds: float) -> dict:
    """Time the reference on the host CPU (bounded: 3 regimes x `seconds`)."""
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libchowdsp_fft_ref.so")
    ref_bench = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    if os.path.exists(ref_so) and os.path.exists(ref_bench):
        try:
            # threads = 0: ref_bench sizes its all-core pool to the CPUs this process can really use (affinity mask, cgroup
            # quota, and a measured parallel-speedup probe for containers throttled by a layer neither of those shows)
            out = subprocess.run([ref_bench, ref_so, str(n), transform, "1", f"{seconds:.2f}", "0", "512"],
                                 capture_output=True, text=True, timeout=120 + 6 * seconds, check=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            return {
                "value": d["hot_1t_transforms_per_s"] / 1e9, "unit": "GFFT/s", "cores": 1, "kind": "reference",
                "sample": f"reference AVX path (simd width {d['simd_width_bytes']} B), N={n} {transform}, in-place "
                          f"FORWARD+BACKWARD with work buffer, one cache-hot frame repeated for {seconds:.1f} s on one thread",
                "stream_1t_GFFT_s": d["stream_1t_transforms_per_s"] / 1e9,
                "allcore_GFFT_s": d["allcore_transforms_per_s"] / 1e9, "allcore_threads": d["allcore_threads"],
                "cores_effective": d.get("cores_effective"), "affinity_cpus": d.get("affinity_cpus"),
                "cgroup_cpu_quota": d.get("cgroup_cpu_quota"), "parallel_speedup_probe": d.get("parallel_speedup_probe"),
                "host_hardware_threads": d.get("host_hardware_threads"),
                "allcore_over_1t": (d["allcore_transforms_per_s"] / d["stream_1t_transforms_per_s"]) if d["stream_1t_transforms_per_s"] > 0 else None,
                "stream_sample": f"{d['stream_mib']} MiB of distinct frames, out of place, {seconds:.1f} s per regime",
            }
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"[bench] reference CPU baseline failed ({e}); using the oracle port\n")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol

    oracle = ol.Oracle()
    s = oracle.new_setup(n, ol.COMPLEX if transform == "complex" else ol.REAL)
    frame = 2 * n if transform == "complex" else n
    x = ol.uniform(64 * frame, 1234).reshape(64, frame)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        y = oracle.transform_batch(s, x, ol.FORWARD)
        oracle.transform_batch(s, y, ol.BACKWARD)
        done += 2 * 64
    dt = time.perf_counter() - t0
    del np
    return {"value": done / dt / 1e9, "unit": "GFFT/s", "cores": 1, "kind": "port",
            "sample": f"plain-C oracle (scalar restatement of the SSE path), N={n} {transform}, 64 frames fwd+bwd repeated for {seconds:.1f} s"}


def run_workload(workload, args, world, rank, dev, frames_override=0, gather=False):
    """One workload, the full protocol; returns the JSON line (dict) on rank 0, None on the other ranks, an int exit code on failure."""
    import torch
    import torch.distributed as dist

    from chowdsp_fft_amd import api, sharding

    n, transform, frames, desc = WORKLOADS[workload]
    if frames_override > 0:
        frames = frames_override
    tr = api.FFT_COMPLEX if transform == "complex" else api.FFT_REAL
    frame_floats = 2 * n if transform == "complex" else n
    default_form = api.get_option("fused_two_pass")
    if workload.endswith("_two_launches"):
        api.set_option("fused_two_pass", 0)
    elif workload.endswith("_tickets"):
        if api.get_option("experiments") != 1:
            sys.stderr.write("[bench] the ticket-queue form needs a library built with EXPERIMENTS=1\n")
            return 4
        api.set_option("fused_two_pass", 1)
    setup = api.fft_new_setup(n, tr)
    if setup is None:
        sys.stderr.write("[bench] fft_new_setup returned NULL\n")
        return 4

    # synthetic uniform(-1, 1) frames, seed 1234 (+rank): every rank owns `frames` frames (weak scaling)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.rand((frames, frame_floats), device=dev, generator=g) * 2 - 1
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    fused = workload.startswith("conv_")
    engine = None
    if workload.startswith("ols_engine"):
        block, parts = n // 2, 16
        ir = torch.rand((block * parts,), device=dev, generator=g) * 2 - 1
        engine = api.ConvEngine(block, frames, ir, shared=True)
        x = torch.rand((frames, block), device=dev, generator=g) * 2 - 1
        y = torch.empty_like(x)
    if fused:
        h = torch.rand((1, frame_floats), device=dev, generator=g) * 2 - 1
        H = torch.empty_like(h)
        api.fft_transform_unordered_batch(setup, h, H, api.FFT_FORWARD)

    def step():
        if engine is not None:
            engine.process(x, y)
        elif fused:
            api.fft_convolve_frames_batch(setup, x, H, y, 1.0 / n)
        else:
            api.fft_transform_batch(setup, x, y, api.FFT_FORWARD)
            api.fft_transform_batch(setup, y, z, api.FFT_BACKWARD)

    launches_per_step_ = launches_per_step(workload)
    transforms_per_frame_step = 2  # forward + backward (fused: both inside one kernel)

    if args.prewarm_seconds > 0:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device=dev)
    event_ms = sharding.max_over_ranks(ev0.elapsed_time(ev1), device=dev)

    # sanity on the timed data: BACKWARD(FORWARD(x)) = N x  (cheap, outside the timed region)
    if not fused and engine is None:
        err = float((z[:256] / n - x[:256]).abs().max())
        if not err < 5e-6:
            sys.stderr.write(f"[bench] round-trip check failed: {err}\n")
            return 5

    gather_info = None
    if gather and world > 1 and engine is None:
        # the final collection of the result shards over RCCL / xGMI: the ONE collective of this path, timed on its own
        # (twice: the first call also builds the communicator's channels), never part of `value`
        first = sharding.timed_gather(z if not fused else y, frames * world, device=dev, synchronize=torch.cuda.synchronize)
        gather_info = sharding.timed_gather(z if not fused else y, frames * world, device=dev, synchronize=torch.cuda.synchronize)
        gather_info["first_call_seconds"] = first["seconds"]

    line = None
    if rank == 0:
        total_transforms = transforms_per_frame_step * frames * world * args.steps
        value = total_transforms / elapsed / 1e9
        launches = launches_per_step_ * args.steps
        avg_launch_s = event_ms * 1e-3 / launches
        alg_bytes_per_launch = (4 * frame_floats * 2) * frames  # read + write one frame each, per launch
        if engine is not None:
            # per channel and block: new samples in + out (2*L), the delay line re-read (P*N) and its new slot written (N)
            alg_bytes_per_launch = 4 * frames * (2 * (n // 2) + 16 * n + n)
            transforms_per_frame_step = 2
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9
        kernel = {
            "c2c_n4096_b65536_fwd_inv": "cfft::c2c_kernel_pf<Plan<4096,256,16,16,16>, 1, INV, false> (forward and backward instantiations)",
            "real_n1024_b262144_fwd_inv": "cfft::r2c_kernel<Plan<512,64,8,8,8>, 4, false> and cfft::c2r_kernel<Plan<512,64,8,8,8>, 4, false>",
            "conv_real_n4096_b131072": "cfft::conv_frames_real_tp_kernel_pf<Plan<2048,128,16,16,8>, 1> (pair stage in registers as one linear map, transposed backward network)",
            "c2c_n65536_b8192_fwd_inv": "cfft::group256_kernel<INV> (both passes in one launch, static groups of 16 workgroups per XCD)",
            "c2c_n65536_b8192_fwd_inv_two_launches": "cfft::pass_a256_kernel<INV> + cfft::pass_b256_kernel<INV> per 256-frame chunk (two passes over the frame)",
            "c2c_n65536_b8192_fwd_inv_tickets": "cfft::fused256_kernel<INV> (both passes in one launch, per-XCD ticket queues)",
            "c2c_n9216_b16384_fwd_inv": "cfft::generic_kernel<0, INV, 1024>",
            "ols_engine_b2048_p16_c8192": "cfft::ols_block_kernel<Plan<2048,128,16,16,8>, 2>",
        }.get(workload, f"{workload} kernels")
        traffic, traffic_extra = committed_traffic(workload)
        line = {
            "metric": "GFFT/s (batched fp32 complex, N=4096) + achieved HBM GB/s vs roofline" if workload == "c2c_n4096_b65536_fwd_inv"
                      else f"GFFT/s ({workload})",
            "value": value, "unit": "GFFT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic uniform(-1,1), seed 1234",
            "config": {"workload": workload, "description": desc, "N": n, "transform": transform,
                       "frames_per_gpu": frames, "global_frames": frames * world,
                       "transforms_per_frame_per_step": transforms_per_frame_step,
                       "sharding": f"contiguous batch blocks over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, **traffic_extra, "kernel": kernel,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch, "avg_launch_ms": avg_launch_s * 1e3,
                         "launches_timed": launches, "timer": "HIP events on the launch stream"},
        }
        if gather_info:
            line["gather_xgmi"] = gather_info
    api.fft_destroy_setup(setup)
    api.set_option("fused_two_pass", default_form)
    del x, y, z, engine
    torch.cuda.empty_cache()
    return line


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="override the workload's batch (debugging only)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="seconds per CPU-baseline regime (3 regimes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default workload only: skip the secondary BASELINE configurations")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed: run the step for this long BEFORE the W warm-up steps so that the GPU has left its idle "
                         "clocks (measured: the first ~3 ms after idle run 10-25 %% slower); the timed region is still exactly K steps")
    ap.add_argument("--gather", action="store_true", help="(default when N > 1) time the final all-gather of results over xGMI, reported separately")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the final all-gather")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="N > 1: process-group backend.  nccl (= RCCL over xGMI) is what a multi-GPU node runs; gloo rehearses the "
                         "same branch on ONE GPU box (ranks share the devices that exist, collectives move host copies)")
    ap.add_argument("--secondary-frames-per-gpu", type=int, default=0,
                    help="override the batch of every secondary workload (rehearsals only: a reduced batch is not a bench result)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if not os.path.exists(os.path.join(ROOT, "chowdsp_fft_amd", "libchowdsp_fft_hip.so")):
        # source-only checkout: compile the library first (rank 0 builds, the others wait for the file)
        if local_rank == 0:
            import __graft_entry__

            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not os.path.exists(os.path.join(ROOT, "chowdsp_fft_amd", "libchowdsp_fft_hip.so")) and time.time() - t_wait < 1800:
                time.sleep(2.0)
            time.sleep(5.0)

    # The CPU baseline runs a child process (oracle/_ref/ref_bench): do it BEFORE anything touches the GPU
    # (a process that has initialised HIP must not fork+exec on this pool).
    cb = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n0, transform0 = WORKLOADS[args.workload][0], WORKLOADS[args.workload][1]
        cb = cpu_baseline(n0, transform0, args.cpu_seconds)
        try:
            with open("/proc/cpuinfo") as f:
                cb["host_cpu"] = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
        except OSError:
            pass
        cb.setdefault("host_hardware_threads", os.cpu_count())
        if cb.get("cores_effective") is None:
            try:
                cb["cores_effective"] = len(os.sched_getaffinity(0))
            except (AttributeError, OSError):
                cb["cores_effective"] = os.cpu_count()

    import torch
    import torch.distributed as dist

    from chowdsp_fft_amd import api, sharding
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1\n")
        if world == 1 and args.gpus > 1:
            return 2
    if not torch.cuda.is_available():
        sys.stderr.write("[bench] no GPU visible: the HIP path cannot run (there is no CPU fallback)\n")
        return 3
    if args.dist_backend == "gloo":
        # rehearsal: more ranks than devices is fine, the ranks then share the devices that exist
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    do_gather = world > 1 and not args.no_gather
    line = run_workload(args.workload, args, world, rank, dev, frames_override=args.frames_per_gpu, gather=do_gather)
    if isinstance(line, int):
        return line
    # BASELINE.json's other configurations, driver-timed in the same run (each is a few ms of GPU time per step): same
    # protocol (prewarm, W warm-up steps, exactly K timed steps between barriers + synchronize, MAX over ranks)
    secondary, multi_extra = [], {}
    if args.workload == DEFAULT_WORKLOAD and not args.no_secondary:
        for name in SECONDARY_WORKLOADS:
            # N > 1: BASELINE configs[3] (1M overlap-save frames over 8 GPUs = 131072 per GPU) gets a gather of its own
            sec = run_workload(name, args, world, rank, dev, frames_override=args.secondary_frames_per_gpu,
                               gather=(do_gather and name == MULTI_GPU_CONV_WORKLOAD))
            if isinstance(sec, int):
                return sec
            if rank == 0:
                secondary.append({"workload": name, "description": sec["config"]["description"], "value": sec["value"], "unit": sec["unit"],
                                  "ms_per_step": sec["ms_per_step"], "steps": sec["steps"], "warmup": sec["warmup"],
                                  "frames_per_gpu": sec["config"]["frames_per_gpu"], "roofline": sec["roofline"]})
                if world > 1 and name == MULTI_GPU_CONV_WORKLOAD:
                    multi_extra[name] = {"description": sec["config"]["description"], "value": sec["value"], "unit": sec["unit"],
                                         "ms_per_step": sec["ms_per_step"], "global_frames": sec["config"]["global_frames"],
                                         "roofline_frac_per_gpu": sec["roofline"]["frac"], "gather_xgmi": sec.get("gather_xgmi")}
    if world > 1:
        # proof of what the process group spans: every rank's device, collected over the process group itself
        props = torch.cuda.get_device_properties(local_rank)
        desc = f"rank {rank}: cuda:{local_rank} {props.name} {getattr(props, 'gcnArchName', '')} {props.multi_processor_count} CUs pid {os.getpid()}"
        rank_desc = sharding.describe_ranks(desc)
        if rank == 0:
            line.update(sharding.multi_gpu_fields(world, rank_desc, line.pop("gather_xgmi", None), multi_extra))
            line["dist_backend"] = args.dist_backend
            if args.dist_backend != "nccl":
                line["rehearsal"] = "gloo process group: ranks may share a device and the gather moves host copies -- not a scaling measurement"
    if rank == 0:
        if secondary:
            line["secondary"] = secondary
        if cb is not None:
            cb["gpu_over_cpu"] = line["value"] / cb["value"] if cb["value"] > 0 else None
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
