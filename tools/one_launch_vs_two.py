#!/usr/bin/env python3
"""The one-launch group forms (options fused_two_pass / group_fast / group_real) against the two-launch chunked form of the same
sizes, toggled inside one process (same buffers): c65536, c131072, c262144, r131072; fwd + bwd per step, 512 MiB batches."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chowdsp_fft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
for w, opts in (("c65536", ("fused_two_pass", 2, 0)), ("c131072", ("group_fast", 1, 0)), ("c262144", ("group_fast", 1, 0)), ("r131072", ("group_real", 1, 0))):
    n, cplx = int(w[1:]), w[0] == "c"
    ff = 2 * n if cplx else n
    frames = (512 << 20) // (4 * ff)
    x = torch.rand((frames, ff), device=dev) * 2 - 1
    y, z = torch.empty_like(x), torch.empty_like(x)
    name, on, off = opts
    res = {}
    s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)
    for rnd in range(5):
        for v in (on, off):
            api.set_option(name, v)
            for _ in range(2):
                api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
                api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
                api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
    api.set_option(name, on)
    alg = 2 * 2 * 4 * ff * frames
    for v in (on, off):
        med = statistics.median(res[v])
        print(f"{w:9s} {name}={v}  {med:8.4f} ms/step  {alg / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    api.fft_destroy_setup(s)
