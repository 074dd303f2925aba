#!/usr/bin/env python3
"""Does the 256 MiB memory-side cache (Infinity Cache) serve a buffer that was just WRITTEN, or just READ?  torch kernels only:
a = write W MiB (fill_), then read it (sum) -- against the same read after 2 GiB of other traffic has passed through.
Also read-after-read, and copy (read + write) of a just-written source.  Prints GB/s of the second operation."""
import sys

import torch

dev = torch.device("cuda:0")
big = torch.empty((2 << 30) // 4, device=dev)
other = torch.empty((2 << 30) // 4, device=dev)


def timed(fn, reps=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def flush():
    other.add_(1.0)  # 2 GiB read + 2 GiB written: whatever the cache held is gone


print(f"{'MiB':>6} {'sum after fill':>15} {'sum after flush':>16} {'sum after sum':>14} {'copy after fill':>16} {'copy after flush':>17}   (GB/s; copy counts read + write)")
for mib in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = (mib << 20) // 4
    a = big[:n]
    b = big[n:2 * n] if 2 * n <= big.numel() else None
    res = []
    for mode in ("fill", "flush", "sum"):
        ts = []
        for _ in range(7):
            a.fill_(1.0)
            if mode == "flush":
                flush()
            if mode == "sum":
                flush()
                a.sum()
            torch.cuda.synchronize()
            ts.append(timed(lambda: a.sum()))
        ts.sort()
        res.append(mib * 1.048576 / ts[len(ts) // 2])
    for mode in ("fill", "flush"):
        ts = []
        for _ in range(7):
            a.fill_(1.0)
            if mode == "flush":
                flush()
            torch.cuda.synchronize()
            ts.append(timed(lambda: b.copy_(a)))
        ts.sort()
        res.append(2 * mib * 1.048576 / ts[len(ts) // 2])
    print(f"{mib:6d} {res[0]:15.0f} {res[1]:16.0f} {res[2]:14.0f} {res[3]:16.0f} {res[4]:17.0f}", flush=True)
