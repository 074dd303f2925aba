#!/usr/bin/env python3
"""Rate of one size against the batch size (footprint): forward + backward, device-resident, HIP events (tools/_timing.py).
usage: tools/batch_sweep.py c4096 8192,16384,32768,65536,131072"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chowdsp_fft_amd import api  # noqa: E402
from _timing import measure, spread  # noqa: E402

dev = torch.device("cuda:0")
sz = sys.argv[1]
cplx, n = sz[0] == "c", int(sz[1:])
nf = 2 * n if cplx else n
s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)
for batch in [int(b) for b in sys.argv[2].split(",")]:
    x = torch.rand((batch, nf), device=dev) * 2 - 1
    y, z = torch.empty_like(x), torch.empty_like(x)

    def step():
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

    ms2, lo, hi = measure(step)
    ms = ms2 / 2
    gbs = batch * nf * 8 / ms / 1e6
    print(f"{sz} batch {batch:<8d} {batch * nf * 4 / 2**20:8.0f} MiB per buffer  {ms:8.4f} ms/launch ({spread(ms2, lo, hi)})  {gbs:7.0f} GB/s  frac {gbs / 8000:5.3f}", flush=True)
    del x, y, z
