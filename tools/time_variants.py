#!/usr/bin/env python3
"""A/B timing of differently built libchowdsp_fft_hip variants in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Each variant is its own dlopen handle (RTLD_LOCAL), so the same
symbol names coexist.

usage: python tools/time_variants.py [--workloads c4096,r1024,c65536,conv4096,c64,c256,c1024] [--rounds 7] lib1.so lib2.so ...
       (any c<N> / r<N>: fwd + bwd of complex / real frames; v<N> / w<N>: fused convolution of real / complex frames)
       (default libs: chowdsp_fft_amd/libchowdsp_fft_hip.so + chowdsp_fft_amd/variants/*.so)
"""
import argparse
import glob
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chowdsp_fft_amd import _lib  # noqa: E402

WORK = {
    # name: (N, transform(1=complex), frames, fused)
    "c4096": (4096, 1, 65536, False),
    "r1024": (1024, 0, 262144, False),
    "c65536": (65536, 1, 8192, False),
    "conv4096": (4096, 0, 131072, True),
    "conv1024": (1024, 0, 524288, True),
    "conv32": (32, 0, 16777216, True),
    "conv64": (64, 0, 8388608, True),
    "conv128": (128, 0, 4194304, True),
    "conv256": (256, 0, 2097152, True),
    "conv2048": (2048, 0, 262144, True),
    "conv8192": (8192, 0, 65536, True),
    "cconv4096": (4096, 1, 65536, True),
    "cconv1024": (1024, 1, 262144, True),
    "cconv2048": (2048, 1, 131072, True),
    "cconv512": (512, 1, 524288, True),
    "cconv256": (256, 1, 1048576, True),
    "cconv128": (128, 1, 2097152, True),
    "cconv64": (64, 1, 4194304, True),
    "cconv32": (32, 1, 8388608, True),
    "cconv16": (16, 1, 16777216, True),
    "c64": (64, 1, 4194304, False),
    "c256": (256, 1, 1048576, False),
    "c1024": (1024, 1, 262144, False),
    "c2048": (2048, 1, 131072, False),
    "r4096": (4096, 0, 131072, False),
    "r8192": (8192, 0, 65536, False),
    "c16": (16, 1, 16777216, False),
    "c32": (32, 1, 8388608, False),
    "c128": (128, 1, 2097152, False),
    "c512": (512, 1, 524288, False),
    "r32": (32, 0, 16777216, False),
    "r64": (64, 0, 8388608, False),
    "r256": (256, 0, 2097152, False),
    "r2048": (2048, 0, 262144, False),
    "ols": (4096, 0, 8192, False),
    "c8192": (8192, 1, 32768, False),
    "r16384": (16384, 0, 32768, False),
    "c16384": (16384, 1, 16384, False),
    "c32768": (32768, 1, 8192, False),
    "r32768": (32768, 0, 16384, False),
    "c9216": (9216, 1, 16384, False),
    "c3072": (3072, 1, 65536, False),
    "c768": (768, 1, 262144, False),
    "c480": (480, 1, 524288, False),
    "c384": (384, 1, 524288, False),
    "c640": (640, 1, 262144, False),
    "c1536": (1536, 1, 131072, False),
    "c2000": (2000, 1, 131072, False),
    "c6400": (6400, 1, 32768, False),
    "r960": (960, 0, 524288, False),
    "r9216": (9216, 0, 32768, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c4096,r1024,conv4096,c65536")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    libs = args.libs or ([os.path.join(ROOT, "chowdsp_fft_amd", "libchowdsp_fft_hip.so")]
                         + sorted(glob.glob(os.path.join(ROOT, "chowdsp_fft_amd", "variants", "*.so"))))
    handles = [(os.path.basename(p).replace("lib_", "").replace(".so", ""), _lib.load(p, strict=False)) for p in libs]
    dev = torch.device("cuda:0")
    stream = int(torch.cuda.current_stream().cuda_stream)
    for wname in args.workloads.split(","):
        if wname not in WORK and wname[0] in "vw" and wname[1:].isdigit():
            # fused convolution of any size: v<N> real frames, w<N> complex frames, ~512 MiB of frames
            nn, trr = int(wname[1:]), (1 if wname[0] == "w" else 0)
            WORK[wname] = (nn, trr, max(1, (512 << 20) // (4 * (2 * nn if trr else nn))), True)
        if wname not in WORK and wname[0] in "cr" and wname[1:].isdigit():
            # any size: c<N> / r<N>, ~512 MiB of frames
            nn, trr = int(wname[1:]), (1 if wname[0] == "c" else 0)
            WORK[wname] = (nn, trr, max(1, (512 << 20) // (4 * (2 * nn if trr else nn))), False)
        n, tr, frames, fused = WORK[wname]
        ff = 2 * n if tr == 1 else n
        x = torch.rand((frames, ff), device=dev) * 2 - 1
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        H = torch.rand((1, ff), device=dev)
        if os.environ.get("TV_SPECTRUM_H"):  # a real filter spectrum (random signs, |H| ~ sqrt(N)) instead of uniform [0,1)
            H = (torch.randn((1, ff), device=dev) * (n / 3.0) ** 0.5).contiguous()
        setups = [(name, L, L.fft_new_setup(n, tr, True)) for name, L in handles]
        engines = {}
        if wname.startswith("ols"):
            # partitioned overlap-save engine: `frames` channels, block n/2, 16 partitions, shared impulse response
            block, parts = n // 2, 16
            ir = torch.rand((block * parts,), device=dev) * 2 - 1
            xb = torch.rand((frames, block), device=dev) * 2 - 1
            yb = torch.empty_like(xb)
            for name, L in handles:
                engines[name] = L.fft_conv_engine_create(block, frames, ir.data_ptr(), block * parts, 1, stream)
                assert engines[name], L.fft_gpu_last_error()

        def step(L, s):
            if engines:
                name = [nm for nm, LL, _ in setups if LL is L][0]
                rc = L.fft_conv_engine_process(engines[name], xb.data_ptr(), yb.data_ptr(), stream)
                assert rc == 0, L.fft_gpu_last_error()
                return
            if fused:
                rc = L.fft_convolve_frames_batch(s, x.data_ptr(), H.data_ptr(), y.data_ptr(), 1.0 / n, frames, stream)
            else:
                rc = L.fft_transform_batch(s, x.data_ptr(), y.data_ptr(), frames, 0, stream)
                rc |= L.fft_transform_batch(s, y.data_ptr(), z.data_ptr(), frames, 1, stream)
            assert rc == 0, L.fft_gpu_last_error()

        times = {name: [] for name, _, _ in setups}
        for name, L, s in setups:  # warm-up
            step(L, s)
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for name, L, s in setups:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    step(L, s)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / args.iters)
        launches = 1 if (fused or engines) else 2
        alg = 2 * 4 * ff * frames  # bytes per launch
        if engines:
            alg = 4 * (2 * (n // 2) + (16 + 1) * n) * frames  # in + out blocks, delay line read, one spectrum write
        print(f"--- {wname}: N={n} {'complex' if tr else 'real'} frames={frames} {'fused conv' if fused else 'fwd+bwd'}")
        for name, ts in times.items():
            med, best = statistics.median(ts), min(ts)
            print(f"{name:28s} median {med:8.4f} ms/step  min {best:8.4f}  -> {alg * launches / med * 1e-6:7.0f} GB/s  "
                  f"{2 * frames / med * 1e-6:8.4f} GFFT/s")
        for name, L, s in setups:
            if engines:
                L.fft_conv_engine_destroy(engines[name])
            L.fft_destroy_setup(s)
        del x, y, z
        torch.cuda.empty_cache()
        sys.stdout.flush()


if __name__ == "__main__":
    main()
