"""Coarse rate floors: a guard against GROSS regressions of the hot path's kernels (a spill, a lost occupancy step, a plan that
silently falls to a slower family, a chunking default that halves the grid) -- not a benchmark.  Each floor is about 55-60 % of what
the size measures on the slowest MI355X box seen so far (profiles/r05_all_sizes_rates.txt, r05_lanes_sweep.txt), so box-to-box
variance (+-5 %) and a cold device cannot trip it; bench.py and profiles/ hold the real numbers.

Rate = algorithmic bytes (read + write one frame per transform) / time, as a fraction of the 8 TB/s HBM peak, 256 MiB batches,
forward + backward per step (fused convolution: one call), median of 3 x 10 steps after 0.2 s of the same call.
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu

# (workload, floor): c<N> / r<N> = transforms of complex / real frames, v<N> = fused convolution of real frames
FLOORS = [("c64", 0.38), ("c4096", 0.40), ("r1024", 0.38), ("v4096", 0.35), ("c65536", 0.27), ("c131072", 0.22), ("r131072", 0.19),
          ("c524288", 0.19), ("r1048576", 0.15), ("c9216", 0.33), ("c46080", 0.19), ("r41472", 0.17), ("r96000", 0.15)]


@pytest.mark.parametrize("workload,floor", FLOORS)
def test_rate_floor(gpu, workload, floor):
    from _timing import measure

    api, torch, dev = gpu
    kind, n = workload[0], int(workload[1:])
    cplx = kind == "c"
    ff = 2 * n if cplx else n
    frames = max(2, (256 << 20) // (4 * ff))
    x = torch.rand((frames, ff), device=dev) * 2 - 1
    y = torch.empty_like(x)
    s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)
    assert s is not None
    if kind == "v":
        H = torch.rand((1, ff), device=dev)
        step, launches = (lambda: api.fft_convolve_frames_batch(s, x, H, y, 1.0 / n)), 1
    else:
        z = torch.empty_like(x)

        def step():
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
            api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

        launches = 2
    ms, lo, hi = measure(step, launches=10, reps=3, prewarm_s=0.2)
    frac = launches * 2 * 4 * ff * frames / (ms * 1e-3) / 8e12
    api.fft_destroy_setup(s)
    print(f"{workload}: {frac:.3f} of 8 TB/s (floor {floor})")
    assert frac >= floor, f"{workload}: {frac:.3f} of 8 TB/s, floor {floor} ({ms:.3f} ms per step, {frames} frames)"
