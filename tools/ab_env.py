#!/usr/bin/env python3
"""A/B of ONE library build under two (or more) values of an environment switch that the library reads PER CALL, interleaved
rounds in one process (same buffers, same workspace, same tables: nothing but the switch differs -- tools/time_variants.py
compares differently built libraries, each with its own workspace, whose placement alone moves the long frames' rates).

usage: python tools/ab_env.py --env CHOWDSP_FFT_REAL4_CARRY=1,0 --workloads r524288,r1048576 [--rounds 7] [--iters 30]
A workload is c<N> / r<N>: ~512 MiB of frames, forward + backward per step.  Prints ms/step and the fraction of 8 TB/s.
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chowdsp_fft_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", required=True, help="NAME=v1,v2[,v3...]")
    ap.add_argument("--workloads", default="r524288,r1048576")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--direction", choices=["both", "fwd", "bwd"], default="both")
    args = ap.parse_args()
    name, vals = args.env.split("=", 1)
    vals = vals.split(",")
    dev = torch.device("cuda:0")
    for w in args.workloads.split(","):
        n, cplx = int(w[1:]), w[0] == "c"
        ff = 2 * n if cplx else n
        frames = max(1, (512 << 20) // (4 * ff))
        x = torch.rand((frames, ff), device=dev) * 2 - 1
        y, z = torch.empty_like(x), torch.empty_like(x)
        s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)
        assert s is not None

        def step():
            if args.direction != "bwd":
                api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
            if args.direction != "fwd":
                api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

        times = {v: [] for v in vals}
        for v in vals:
            os.environ[name] = v
            step()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v in vals:
                os.environ[name] = v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    step()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.iters)
        launches = 2 if args.direction == "both" else 1
        alg = launches * 2 * 4 * ff * frames
        print(f"--- {w}: frames={frames} direction={args.direction}")
        for v in vals:
            med = statistics.median(times[v])
            print(f"{name}={v:8s} median {med:8.4f} ms/step  min {min(times[v]):8.4f}  -> {alg / med / 1e6:7.0f} GB/s  {alg / med / 1e6 / 8000:.3f} of 8 TB/s")
        api.fft_destroy_setup(s)


if __name__ == "__main__":
    main()
