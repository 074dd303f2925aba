#!/usr/bin/env python3
"""Per-block cost of the partitioned convolution engine for SMALL jobs (launch-bound): eager process() calls versus one
captured process() replayed as a hipGraph (the engine keeps its position in device memory for exactly this)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from chowdsp_fft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
for block, channels, parts in ((256, 8, 8), (512, 64, 16), (2048, 64, 16)):
    ir = np.random.default_rng(0).uniform(-1, 1, block * parts).astype(np.float32)
    side = torch.cuda.Stream()
    x = torch.rand((channels, block), device=dev)
    y = torch.empty_like(x)
    with torch.cuda.stream(side):
        eng = api.ConvEngine(block, channels, ir, shared=True)
        for _ in range(10):
            eng.process(x, y)
    side.synchronize()
    n = 2000
    with torch.cuda.stream(side):
        t0 = time.perf_counter()
        for _ in range(n):
            eng.process(x, y)
        side.synchronize()
        eager = (time.perf_counter() - t0) / n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.process(x, y)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n
    print(f"engine block={block} channels={channels} partitions={parts}: eager {eager * 1e6:6.1f} us/block, "
          f"hipGraph replay {graph * 1e6:6.1f} us/block  (block = {block / 48000 * 1e6:.0f} us of audio at 48 kHz)")
    eng.close()
