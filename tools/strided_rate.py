#!/usr/bin/env python3
"""Throughput of fft_transform_batch_strided next to the packed call: frames embedded at stride
2 x frame (the JUCE adapter's buffer shape) and at frame + 16 bytes."""
import sys

import torch

sys.path.insert(0, ".")
from chowdsp_fft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
for n, tr, frames in ((4096, api.FFT_COMPLEX, 32768), (1024, api.FFT_REAL, 262144), (64, api.FFT_COMPLEX, 2097152), (65536, api.FFT_COMPLEX, 2048)):
    nf = 2 * n if tr == api.FFT_COMPLEX else n
    s = api.fft_new_setup(n, tr)
    for stride in (nf, nf + 4, 2 * nf):
        x = torch.rand((frames, stride), device=dev)
        y = torch.empty_like(x)

        def step():
            api.fft_transform_batch_strided(s, x, stride, y, stride, frames, api.FFT_FORWARD)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"N={n:6d} {'complex' if tr else 'real   '} stride={stride:7d} floats ({'packed' if stride == nf else 'strided'}): "
              f"{ms:7.3f} ms  {2 * 4 * nf * frames / ms * 1e-6:6.0f} GB/s algorithmic  {frames / ms * 1e-6:8.4f} GFFT/s")
    api.fft_destroy_setup(s)
