#!/usr/bin/env python3
"""fft_convolve_frames_batch rates (forward -> * H * scaling -> backward per frame), device-resident frames, HIP events.
Algorithmic bytes per frame: the frame read once + written once (16 N complex, 8 N real; the shared filter stays in L2).
usage: conv_frames_rate.py [c|r]N ...   e.g. c16384 c32768 r32768 r65536; option fused_conv_split=0 times the three-launch form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chowdsp_fft_amd import api  # noqa: E402
from _timing import measure, spread  # noqa: E402

dev = torch.device("cuda:0")
opts = [a for a in sys.argv[1:] if "=" in a]
sys.argv = [a for a in sys.argv if "=" not in a]
for o in opts:
    api.set_option(o.split("=")[0], int(o.split("=")[1]))
sizes = sys.argv[1:] or ["c4096", "r8192", "c8192", "r16384", "c16384", "r32768", "c32768", "r65536"]
# (per size: 0.3 s of the timed call itself untimed, then the median of 5 repetitions of 20 launches -- tools/_timing.py)
for form, opt in (("one launch where fused", 1), ("three launches (fused_conv_split = fused_conv_generic = fused_conv_group = 0)", 0)):
    api.set_option("fused_conv_split", opt)
    api.set_option("fused_conv_generic", opt)
    api.set_option("fused_conv_group", opt)
    print(f"# {form}")
    for sz in sizes:
        real, n = sz[0] == "r", int(sz[1:])
        nf = n if real else 2 * n
        batch = max(1, (1 << (30 if n >= 65536 else 28)) // nf)  # 1 GiB in + 1 GiB out (4 GiB each for the two-pass sizes)
        s = api.fft_new_setup(n, api.FFT_REAL if real else api.FFT_COMPLEX)
        x = torch.rand((batch, nf), device=dev) * 2 - 1
        h = torch.rand((1, nf), device=dev) * 2 - 1
        H = torch.empty_like(h)
        api.fft_transform_unordered_batch(s, h, H, api.FFT_FORWARD)
        y = torch.empty_like(x)
        ms, lo, hi = measure(lambda: api.fft_convolve_frames_batch(s, x, H, y, 1.0 / n))
        nbytes = 2 * batch * nf * 4
        print(f"{sz:>7s} batch={batch:<7d} {ms:8.4f} ms ({spread(ms, lo, hi)})  {nbytes / ms / 1e6:7.0f} GB/s algorithmic  {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s")
        api.fft_destroy_setup(s)
