#!/usr/bin/env python3
"""JUCE-engine adapter rates (fft_juce_real_*_batch), device-resident buffers of 2N floats per frame, HIP events.
Algorithmic bytes per frame: forward N*4 read + (N+2)*4 (or 2N*4 with the negative frequencies) written; inverse the reverse
of the first.  Run with CHOWDSP_FFT_JUCE_FUSED=0 for the two-launch form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chowdsp_fft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
orders = [int(a) for a in sys.argv[1:]] or [9, 10, 12, 14]
for order in orders:
    n = 1 << order
    batch = max(1, (1 << 27) // n)  # 1 GiB of buffers
    eng = api.JuceFFT(order)
    buf = torch.rand((batch, 2 * n), device=dev) * 2 - 1
    for name, fn, nbytes in (
        ("forward, half", lambda: eng.perform_real_only_forward_transform(buf, True), batch * (2 * n + 2) * 4),
        ("forward, full", lambda: eng.perform_real_only_forward_transform(buf, False), batch * 3 * n * 4),
        ("inverse", lambda: eng.perform_real_only_inverse_transform(buf), batch * (2 * n + 2) * 4),
    ):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"N={n:<6d} batch={batch:<8d} {name:14s} {ms:8.4f} ms  {nbytes / ms / 1e6:7.0f} GB/s algorithmic")
    eng.close()
