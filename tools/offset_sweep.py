#!/usr/bin/env python3
"""Does the rate depend on where the output buffer sits relative to the input?  N = 4096 complex, 65536 frames (2 GiB per buffer), x -> y -> z,
the three buffers carved out of ONE allocation with `pad` bytes between them.   usage: tools/offset_sweep.py [c4096] [batch]"""
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
pads = [int(p) for p in os.environ["OFFSET_SWEEP_PADS"].split(",")] if os.environ.get("OFFSET_SWEEP_PADS") else [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, (16 << 20) + 65536, (64 << 20) + 256]
for pad in pads:
    padf = pad // 4
    pool = torch.empty(3 * elems + 2 * padf + 1024, device=dev)
    base = (-(pool.data_ptr() // 4)) % 64  # 256-byte aligned start
    x = pool[base:base + elems].view(batch, nf)
    y = pool[base + elems + padf:base + 2 * elems + padf].view(batch, nf)
    z = pool[base + 2 * elems + 2 * padf:base + 3 * elems + 2 * padf].view(batch, nf)
    x.uniform_(-1, 1)

    def step():
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

    ms2, lo, hi = measure(step)
    ms = ms2 / 2
    gbs = batch * nf * 8 / ms / 1e6
    print(f"{sz} batch {batch} pad {pad:>10d} B  x at 0x{x.data_ptr():x}  {ms:8.4f} ms/launch ({spread(ms2, lo, hi)})  {gbs:7.0f} GB/s  frac {gbs / 8000:5.3f}", flush=True)
    del x, y, z, pool
    torch.cuda.empty_cache()
