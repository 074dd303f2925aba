#!/usr/bin/env python3
"""Placement of the buffers, two factors apart: ONE pool (fixed address), x at pool + xoff, y = x + size + gap, z = y + size + gap.
N = 4096 complex, 65536 frames (2 GiB per buffer), forward x -> y, backward y -> z.   usage: tools/offset_sweep2.py [c4096] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chowdsp_fft_amd import api  # noqa: E402
from _timing import measure, spread  # noqa: E402

dev = torch.device("cuda:0")
sz = sys.argv[1] if len(sys.argv) > 1 else "c4096"
cplx, n = sz[0] == "c", int(sz[1:])
nf = 2 * n if cplx else n
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)
elems = batch * nf
MiB = 1 << 20
pool = torch.empty(3 * elems + (64 * MiB) // 4, device=dev)
print(f"pool at 0x{pool.data_ptr():x}")
for xoff in (0, 2 * MiB, 4 * MiB, 8 * MiB):
    for gap in (0, 1 * MiB, 2 * MiB, 4 * MiB, 8 * MiB):
        a, g = xoff // 4, gap // 4
        x = pool[a:a + elems].view(batch, nf)
        y = pool[a + elems + g:a + 2 * elems + g].view(batch, nf)
        z = pool[a + 2 * elems + 2 * g:a + 3 * elems + 2 * g].view(batch, nf)
        x.uniform_(-1, 1)

        def step():
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
            api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

        ms2, lo, hi = measure(step, launches=20, reps=3, prewarm_s=0.2)
        ms = ms2 / 2
        gbs = batch * nf * 8 / ms / 1e6
        print(f"{sz} xoff {xoff // MiB:2d} MiB gap {gap // MiB:2d} MiB  {ms:8.4f} ms/launch ({spread(ms2, lo, hi)})  {gbs:7.0f} GB/s  frac {gbs / 8000:5.3f}", flush=True)
