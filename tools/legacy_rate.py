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
    # call coalescing: the same single-frame calls on independent buffers, recorded and flushed as one batch
    nq = 512 if n <= 4096 else 64
    xs = [api.aligned_array(nf) for _ in range(nq)]
    ys = [api.aligned_array(nf) for _ in range(nq)]
    for xa, _ in xs:
        xa[:] = x
    for _ in range(2):
        t0 = time.perf_counter()
        api.defer_begin()
        for (xa, _), (ya, _) in zip(xs, ys):
            api.fft_transform(s, xa, ya, None, api.FFT_FORWARD)
        api.defer_flush()
        dtq = time.perf_counter() - t0
    print(f"coalesced legacy calls (defer_begin/flush), N={n} {name}: {dtq / nq * 1e6:8.2f} us/call {nq / dtq:10.0f} FFT/s  ({nq} calls per flush, incl. the Python call overhead)")
    for _, pp in xs + ys:
        api.aligned_free(pp)
    # the streaming mode: the same frames in ONE call, pinned in place, two-stream pipeline
    nb = max(64, (512 << 20) // (4 * nf))
    X = np.random.default_rng(1).uniform(-1, 1, (nb, nf)).astype(np.float32)
    Y = np.zeros_like(X)  # touched: the first call would otherwise also pay the page faults of a fresh output buffer
    api.fft_transform_host_batch(s, X[:64], Y[:64], api.FFT_FORWARD)
    t0 = time.perf_counter()
    api.fft_transform_host_batch(s, X, Y, api.FFT_FORWARD)
    dt_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    api.fft_transform_host_batch(s, X, Y, api.FFT_FORWARD)
    dt = time.perf_counter() - t0
    print(f"   (first call on these buffers: {X.nbytes / dt_first / 1e9:.1f} GB/s each way; steady state below)")
    print(f"fft_transform_host_batch,        N={n} {name}: {dt / nb * 1e6:8.2f} us/frame {nb / dt:10.0f} FFT/s  "
          f"({X.nbytes / dt / 1e9:.1f} GB/s each way over PCIe, {nb} frames, pin + copy + transform + copy)")
    api.fft_destroy_setup(s)
    api.aligned_free(xp)
    api.aligned_free(yp)
