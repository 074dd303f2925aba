#!/usr/bin/env python3
"""Rate of the long two-pass frames against WHERE the two workspaces lie (experiment build: tools/build_tu_variant.sh wsexp capi
"-DCFFT_WS_EXPERIMENT=1"; the library then allocates its workspaces with 64 MiB of slack and uses them CHOWDSP_FFT_WS_SHIFT0 /
_SHIFT1 bytes in, read per call).  Same buffers, same tables, same kernels: only the workspace addresses move.

usage: python tools/ws_shift_sweep.py [--workloads r1048576,c524288] [--shifts 0,65536,...] [--lib chowdsp_fft_amd/variants/lib_wsexp.so]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="r1048576,c524288")
    ap.add_argument("--shifts", default="0,4096,65536,262144,1048576,2097152,3145728,4194304,6291456,8388608,16777216,33554432")
    ap.add_argument("--lib", default=os.path.join(ROOT, "chowdsp_fft_amd", "variants", "lib_wsexp.so"))
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--pre-alloc-mib", type=int, default=0, help="allocate (and keep) this much before the library's first call: moves every later allocation")
    args = ap.parse_args()
    os.environ["CHOWDSP_FFT_LIB"] = args.lib
    import torch
    from chowdsp_fft_amd import api

    dev = torch.device("cuda:0")
    keep = torch.empty((args.pre_alloc_mib << 20,), dtype=torch.uint8, device=dev) if args.pre_alloc_mib else None
    shifts = [int(v) for v in args.shifts.split(",")]
    for w in args.workloads.split(","):
        n, cplx = int(w[1:]), w[0] == "c"
        ff = 2 * n if cplx else n
        frames = max(1, (512 << 20) // (4 * ff))
        x = torch.rand((frames, ff), device=dev) * 2 - 1
        y, z = torch.empty_like(x), torch.empty_like(x)
        s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)

        def step():
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
            api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

        os.environ["CHOWDSP_FFT_WS_PRINT"] = "1"
        os.environ["CHOWDSP_FFT_WS_SHIFT0"] = os.environ["CHOWDSP_FFT_WS_SHIFT1"] = "0"
        api.fft_transform_batch(s, x[:2], y[:2], api.FFT_FORWARD)  # prints the base addresses (one chunk: lane 0)
        step()
        torch.cuda.synchronize()
        del os.environ["CHOWDSP_FFT_WS_PRINT"]
        print(f"--- {w}: frames={frames}  x {x.data_ptr():#x} y {y.data_ptr():#x} z {z.data_ptr():#x}", flush=True)
        alg = 2 * 2 * 4 * ff * frames
        for mode in ("both", "lane1"):
            for sh in shifts:
                os.environ["CHOWDSP_FFT_WS_SHIFT0"] = str(sh if mode == "both" else 0)
                os.environ["CHOWDSP_FFT_WS_SHIFT1"] = str(sh)
                step()
                torch.cuda.synchronize()
                ts = []
                for _ in range(args.rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        step()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / args.iters)
                med = statistics.median(ts)
                print(f"shift {mode:5s} {sh:>10d}  {med:8.4f} ms/step  {alg / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
        api.fft_destroy_setup(s)


if __name__ == "__main__":
    main()
