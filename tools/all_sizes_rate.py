#!/usr/bin/env python3
"""Rate table: every power of two (complex 2^4..2^21, real 2^5..2^21) and a few mixed-radix lengths, forward + backward,
device-resident batches of about 1 GiB, HIP events.  Algorithmic bytes = 16 N per complex transform, 8 N per real one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chowdsp_fft_amd import api  # noqa: E402
from _timing import measure, spread  # noqa: E402

dev = torch.device("cuda:0")
sizes = [(1 << lg, True) for lg in range(4, 22)] + [(1 << lg, False) for lg in range(5, 22)]
# mixed radix: the reference's "other sizes" (test/test.cpp:279-285) complex and real, then a spread of others
sizes += [(n, True) for n in (96, 192, 384, 480, 640, 768, 9216)] + [(n, False) for n in (96, 192, 384, 480, 640, 768, 9216)]
sizes += [(n, True) for n in (1536, 1920, 5120, 19200, 20736, 46080, 245760)] + [(n, False) for n in (1920, 3840, 18432, 41472)]
# round 3: more of the long plans (run-time factors, 48 kHz seconds, factors beyond 639 points, two-pass beyond 2^21)
sizes += [(n, True) for n in (34992, 48000, 655360, 1327104, 1 << 22)] + [(n, False) for n in (96000,)]
# round 4: frames beyond 2^22 complex points (strided columns x contiguous rows + one transpose)
sizes += [(n, True) for n in (1 << 23, 3 << 21, 1 << 24)] + [(n, False) for n in (1 << 24,)]
if len(sys.argv) > 1:  # subset: "c480,c768,r9216"; further arguments: library options, "split8192=1"
    sizes = [(int(t[1:]), t[0] == "c") for t in sys.argv[1].split(",")]
for opt in sys.argv[2:]:
    api.set_option(opt.split("=")[0], int(opt.split("=")[1]))
for n, cplx in sizes:
    nf = 2 * n if cplx else n
    batch = max(1, (1 << 28) // nf)  # 1 GiB of fp32 per buffer
    s = api.fft_new_setup(n, api.FFT_COMPLEX if cplx else api.FFT_REAL)
    if s is None:
        print(f"{'c' if cplx else 'r'}{n}: not supported")
        continue
    # Two placements of the three buffers, carved out of one pool: every buffer at 0 and at 2 MiB (mod 4 MiB).  The memory system's address hash
    # makes the same kernel up to 8 % faster at one residue than at the other (profiles/r05_placement_residue.txt); which one an allocator hands
    # out is luck, so the row reports the better placement and lists both.
    elems = batch * nf
    MiB = 1 << 20
    pool = torch.empty(3 * elems + (24 * MiB) // 4, device=dev)
    rows = []
    for res in (0, 2):
        off = ((res * MiB - pool.data_ptr()) % (4 * MiB)) // 4
        step_elems = elems + (-elems) % MiB  # keeps y and z at x's residue (1 Mi floats = 4 MiB)
        x = pool[off:off + elems].view(batch, nf)
        y = pool[off + step_elems:off + step_elems + elems].view(batch, nf)
        z = pool[off + 2 * step_elems:off + 2 * step_elems + elems].view(batch, nf)
        x.uniform_(-1, 1)

        def step():
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
            api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)

        # median of 5 repetitions of 10 forward + 10 backward launches after 0.3 s of the same calls untimed (tools/_timing.py)
        ms2, lo, hi = measure(step, launches=10)
        err = float((z[:4] / n - x[:4]).abs().max())
        rows.append((ms2, lo, hi, err))
    ms2, lo, hi, err = min(rows)
    ms = ms2 / 2
    gbs = batch * nf * 8 / ms / 1e6
    both = " / ".join(f"{batch * nf * 8 / (r[0] / 2) / 1e6 / 8000:5.3f}" for r in rows)
    print(f"{'c' if cplx else 'r'}{n:<8d} batch {batch:<9d} {ms:8.4f} ms/launch ({spread(ms2, lo, hi)})  {gbs:7.0f} GB/s  frac {gbs / 8000:5.3f}   round-trip err {max(r[3] for r in rows):.1e}   (buffers at 0 / 2 MiB mod 4 MiB: {both})")
    api.fft_destroy_setup(s)
    del x, y, z, pool
