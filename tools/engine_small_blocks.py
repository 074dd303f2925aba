#!/usr/bin/env python3
"""Per-block cost of the partitioned convolution engine for SHORT blocks (low-latency audio): the two fused kernels versus
the five-kernel path (CHOWDSP_FFT_ENGINE_FUSED=0 is read at engine creation)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from chowdsp_fft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
for block, channels, parts in ((32, 8, 32), (64, 8, 16), (64, 1024, 16), (64, 65536, 16), (128, 8, 16)):
    ir = np.random.default_rng(0).uniform(-1, 1, block * parts).astype(np.float32)
    x = torch.rand((channels, block), device=dev)
    y = torch.empty_like(x)
    res = {}
    for fused in ("1", "0"):
        os.environ["CHOWDSP_FFT_ENGINE_FUSED"] = fused
        eng = api.ConvEngine(block, channels, ir, shared=True)
        for _ in range(20):
            eng.process(x, y)
        torch.cuda.synchronize()
        n = 1000
        t0 = time.perf_counter()
        for _ in range(n):
            eng.process(x, y)
        torch.cuda.synchronize()
        res[fused] = (time.perf_counter() - t0) / n
        eng.close()
    traffic = 4 * channels * (2 * block + (parts + 1) * 2 * block)
    print(f"block={block:5d} channels={channels:6d} partitions={parts:3d}: fused {res['1'] * 1e6:8.1f} us/block "
          f"({traffic / res['1'] / 1e9:7.1f} GB/s algorithmic), five kernels {res['0'] * 1e6:8.1f} us/block")
