#!/usr/bin/env python3
"""fft_transform_2d_batch rates: forward + backward over device-resident batches of about 1 GiB, HIP events.
Algorithmic bytes = 16 * rows * cols per array and direction (one read, one write); the two passes move twice that.
usage: tools/rate_2d.py [256x256,480x640,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chowdsp_fft_amd import api  # noqa: E402
from _timing import measure, spread  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(256, 256), (512, 512), (1024, 1024), (2048, 2048), (480, 640), (720, 1280), (1080, 1920), (144, 192), (600, 800)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for rows, cols in shapes:
    s = api.fft_new_setup(cols, api.FFT_COMPLEX)
    if s is None:
        print(f"{rows}x{cols}: not supported")
        continue
    batch = max(1, (1 << 27) // (rows * cols))
    x = torch.rand((batch, rows, 2 * cols), device=dev) * 2 - 1
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    def step():
        api.fft_transform_2d_batch(s, rows, x, y, batch, api.FFT_FORWARD)
        api.fft_transform_2d_batch(s, rows, y, z, batch, api.FFT_BACKWARD)

    ms2, lo, hi = measure(step, launches=10)  # median of 5 repetitions of 10 forward + 10 backward calls (tools/_timing.py)
    ms = ms2 / 2
    err = float((z[:1] / (rows * cols) - x[:1]).abs().max())
    gbs = batch * rows * cols * 16 / ms / 1e6
    print(f"{rows:5d} x {cols:<5d} batch {batch:<6d} {ms:8.4f} ms/launch ({spread(ms2, lo, hi)})  {gbs:7.0f} GB/s  frac {gbs / 8000:5.3f}   round-trip err {err:.1e}")
    api.fft_destroy_setup(s)
    del x, y, z
