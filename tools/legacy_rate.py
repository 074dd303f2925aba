#!/usr/bin/env python3
"""PCIe-inclusive rate of the LEGACY single-frame symbols with host pointers (one frame staged per call).
This is the drop-in compatibility path, never the number reported as `value` by bench.py."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from chowdsp_fft_amd import api  # noqa: E402

for n, tr, name in ((4096, api.FFT_COMPLEX, "complex"), (1024, api.FFT_REAL, "real"), (65536, api.FFT_COMPLEX, "complex")):
    nf = 2 * n if tr == api.FFT_COMPLEX else n
    x, xp = api.aligned_array(nf)
    y, yp = api.aligned_array(nf)
    x[:] = np.random.default_rng(0).uniform(-1, 1, nf)
    s = api.fft_new_setup(n, tr)
    for _ in range(20):
        api.fft_transform(s, x, y, None, api.FFT_FORWARD)
    calls = 2000 if n <= 4096 else 300
    t0 = time.perf_counter()
    for _ in range(calls):
        api.fft_transform(s, x, y, None, api.FFT_FORWARD)
    dt = time.perf_counter() - t0
    print(f"legacy fft_transform, host pointers, N={n} {name}: {dt / calls * 1e6:8.1f} us/call  {calls / dt:10.0f} FFT/s  "
          f"({2 * nf * 4 * calls / dt / 1e9:.3f} GB/s over PCIe incl. launch + sync)")
    api.fft_destroy_setup(s)
    api.aligned_free(xp)
    api.aligned_free(yp)
