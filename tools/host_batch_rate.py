#!/usr/bin/env python3
"""PCIe-inclusive rate of fft_transform_host_batch by how the caller's memory reaches the DMA engines (host_stream.cpp):
pinned in place (whole pages inside the buffer; partial head / tail pages through bounce pages) for large buffers -- page
aligned and as an unaligned view (+20 bytes) of a larger array -- and the library-owned staging ring for small ones.
Never the number reported as `value` by bench.py."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from chowdsp_fft_amd import api  # noqa: E402

n = 4096
nf = 2 * n
s = api.fft_new_setup(n, api.FFT_COMPLEX)
rng = np.random.default_rng(1)


def rate(label, X, Y, reps=3):
    api.fft_transform_host_batch(s, X, Y, api.FFT_FORWARD)  # first call: page faults of fresh buffers + pinning
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        api.fft_transform_host_batch(s, X, Y, api.FFT_FORWARD)
        best = min(best, time.perf_counter() - t0)
    print(f"{label:62s} {X.nbytes / 2**20:8.1f} MiB  {X.nbytes / best / 1e9:6.2f} GB/s each way  {X.shape[0] / best:10.0f} FFT/s", flush=True)


nb = (512 << 20) // (4 * nf)
X = rng.uniform(-1, 1, (nb, nf)).astype(np.float32)
Y = np.zeros_like(X)
rate("512 MiB, arrays of their own (pinned in place)", X, Y)
# unaligned views: 5 floats (20 bytes) into a larger array -- first and last page shared with the rest of the parent
big_in = np.zeros(nb * nf + 4096, np.float32)
big_out = np.zeros(nb * nf + 4096, np.float32)
Xv = big_in[5:5 + nb * nf].reshape(nb, nf)
Yv = big_out[5:5 + nb * nf].reshape(nb, nf)
Xv[:] = X
rate("512 MiB, views at +20 bytes of larger arrays (interior pinned)", Xv, Yv)
assert np.array_equal(Yv, Y), "view result differs from the aligned result"
assert not big_out[:5].any() and not big_out[5 + nb * nf:].any(), "bytes outside the view were written"
for mib in (64, 16, 4, 1):
    k = (mib << 20) // (4 * nf)
    rate(f"{mib} MiB slice" + (" (staging ring)" if mib < 8 else " (pinned in place)"), X[:k], Y[:k], reps=5)
api.fft_destroy_setup(s)
