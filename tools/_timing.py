"""Timing helper shared by the rate tools (round 4): the secondary tables have to be trustworthy to +-3 %.

Round 3's tools warmed the clocks once per PROCESS and timed one run of 8-10 launches per size: rows measured first
under-reported by up to 20 % (the same kernel read 0.480 and 0.602 in two sections of one table).  Now, per size:
  * the timed call itself runs untimed for `prewarm_s` seconds first (as bench.py does: from idle clocks the first
    milliseconds run 10-25 % slow),
  * `reps` repetitions (>= 5) of `launches` calls (>= 20) are timed separately with HIP events on torch's current stream
    (the stream the tools launch on),
  * the MEDIAN repetition is reported, with min and max beside it.
"""
import time

import torch


def measure(fn, launches: int = 20, reps: int = 5, prewarm_s: float = 0.3):
    """fn() enqueues ONE unit of work on the current stream.  Returns (median, min, max) milliseconds per fn()."""
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= prewarm_s:
                break
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / launches)
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


def spread(med: float, lo: float, hi: float) -> str:
    """'-1.2/+0.8 %' : how far the fastest / slowest repetition is from the median"""
    return f"{100 * (lo / med - 1):+.1f}/{100 * (hi / med - 1):+.1f} %"
