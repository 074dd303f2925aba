#!/usr/bin/env python3
"""Rate against the ABSOLUTE placement of the buffers: x at an address = r (mod 8 MiB), y = x + size, z = y + size (same residue).
N = 4096 complex, 65536 frames.  The library comes from CHOWDSP_FFT_LIB (tools/time_variants.py's variants).   usage: tools/offset_sweep3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chowdsp_fft_amd import api  # noqa: E402
from _timing import measure  # noqa: E402

dev = torch.device("cuda:0")
n, batch = 4096, 65536
nf = 2 * n
s = api.fft_new_setup(n, api.FFT_COMPLEX)
elems = batch * nf
MiB = 1 << 20
pool = torch.empty(3 * elems + (32 * MiB) // 4, device=dev)
base = pool.data_ptr()
out = []
for rx, ry in ((0, 0), (2, 2), (4, 4), (6, 6), (0, 2), (2, 0), (1, 1), (3, 3)):
    offx = ((rx * MiB - base) % (8 * MiB)) // 4
    x = pool[offx:offx + elems].view(batch, nf)
    offy = offx + elems + (((ry - rx) * MiB) % (8 * MiB)) // 4
    y = pool[offy:offy + elems].view(batch, nf)
    offz = offy + elems + (((rx - ry) * MiB) % (8 * MiB)) // 4  # z back at x's residue (the backward transform reads y, writes z)
    z = pool[offz:offz + elems].view(batch, nf)
    assert x.data_ptr() % (8 * MiB) == rx * MiB and y.data_ptr() % (8 * MiB) == ry * MiB
    x.uniform_(-1, 1)

    def step():
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

    ms2, lo, hi = measure(step, launches=20, reps=3, prewarm_s=0.2)
    out.append(f"in {rx} out {ry}: {batch * nf * 8 / (ms2 / 2) / 1e6 / 8000:5.3f}")
print(os.path.basename(os.environ.get("CHOWDSP_FFT_LIB", "default")), " | ".join(out), flush=True)
