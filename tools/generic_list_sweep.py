#!/usr/bin/env python3
"""Stage-list sweep of the one-pass mixed-radix plans: every candidate list of a length through the RUN-TIME-list kernel
(CHOWDSP_FFT_GENERIC_STATIC=0, so that the planner's own list gets no compile-time advantage), one process, plans rebuilt per candidate
(developer override CHOWDSP_FFT_GENERIC_LIST, table cache off).  Complex frames, fwd + bwd, 256 MiB batches; fraction of 8 TB/s.

usage: CHOWDSP_FFT_GENERIC_STATIC=0 python tools/generic_list_sweep.py 192 240 ...      (candidates: every ordered list of up to 4 radices <= 16)
"""
import itertools
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chowdsp_fft_amd import api  # noqa: E402

RADICES = (16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2)


def lists(n, max_stages=4):
    out = set()

    def rec(rest, cur):
        if rest == 1:
            out.add(tuple(cur))
            return
        if len(cur) == max_stages:
            return
        for r in RADICES:
            if rest % r == 0:
                rec(rest // r, cur + [r])

    rec(n, [])
    # drop lists with a radix-2 stage unless nothing else exists, and more than 3 stages where a 2- or 3-stage list exists
    good = [l for l in out if 2 not in l and len(l) >= 2]
    short = [l for l in good if len(l) <= 3]
    cand = short or good
    # long frames have hundreds of ordered lists: keep every order of the MULTISETS with the smallest radix sums (the planner's criterion) only
    sets = sorted({tuple(sorted(l)) for l in cand}, key=lambda m: (len(m), sum(m)))[: int(os.environ.get("SWEEP_MULTISETS", "1000"))]
    cand = [l for l in cand if tuple(sorted(l)) in sets]
    return sorted(cand, key=lambda l: (len(l), l))


def rate(n, real=False):
    dev = torch.device("cuda:0")
    ff = n if real else 2 * n
    frames = (256 << 20) // (4 * ff)
    x = torch.rand((frames, ff), device=dev) * 2 - 1
    y, z = torch.empty_like(x), torch.empty_like(x)
    s = api.fft_new_setup(n, api.FFT_REAL if real else api.FFT_COMPLEX)
    assert s is not None

    def step():
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    err = float((z[:64] / n - x[:64]).abs().max())
    api.fft_destroy_setup(s)
    ms = statistics.median(ts)
    return 2 * 2 * 4 * ff * frames / (ms * 1e-3) / 8e12, err


def main():
    api.set_option("table_cache_mib", 0)
    for n in [int(a) for a in sys.argv[1:]]:
        os.environ.pop("CHOWDSP_FFT_GENERIC_LIST", None)
        base, err = rate(n)
        d = api.plan_describe(n, api.FFT_COMPLEX)
        print(f"--- n = {n}: planner's list {d['radices']}  {base:.3f}  (round-trip err {err:.1e})", flush=True)
        res = []
        for l in lists(n):
            os.environ["CHOWDSP_FFT_GENERIC_LIST"] = f"{n}:" + ",".join(map(str, l))
            f, err = rate(n)
            assert err < 1e-4, (n, l, err)
            res.append((f, l))
        res.sort(reverse=True)
        for f, l in res[:6]:
            print(f"    {list(l)}  {f:.3f}  ({100 * (f / base - 1):+.1f} %)", flush=True)
        os.environ.pop("CHOWDSP_FFT_GENERIC_LIST", None)


if __name__ == "__main__":
    main()
