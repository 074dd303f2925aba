#!/usr/bin/env python3
"""Developer GPU check (not part of the test-suite): numerics vs float64 numpy and raw kernel timing.

usage: python tools/gpu_check.py [--sizes 16,64,...] [--time-n 4096 --time-batch 65536]
"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from chowdsp_fft_amd import api  # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def packed_real(spec):  # numpy rfft output [batch, N/2+1] -> pffft packed layout [batch, N]
    b, m1 = spec.shape
    n = 2 * (m1 - 1)
    out = np.empty((b, n), dtype=np.float64)
    out[:, 0] = spec[:, 0].real
    out[:, 1] = spec[:, -1].real
    out[:, 2::2] = spec[:, 1:-1].real
    out[:, 3::2] = spec[:, 1:-1].imag
    return out


def check_complex(n, batch, dev):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    if s is None:
        return f"N={n}: setup NULL"
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD)
    torch.cuda.synchronize()
    y = yd.cpu().numpy().astype(np.float64)
    ref = np.fft.fft(x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2], axis=1)
    got = y[:, 0::2] + 1j * y[:, 1::2]
    e_f = max(rel_l2(got[i], ref[i]) for i in range(batch))
    zd = torch.empty_like(xd)
    api.fft_transform_batch(s, yd, zd, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    z = zd.cpu().numpy() / n
    e_b = float(np.abs(z - x).max())
    # in-place
    api.fft_transform_batch(s, xd, xd, api.FFT_FORWARD)
    torch.cuda.synchronize()
    e_ip = float((xd - yd).abs().max())
    api.fft_destroy_setup(s)
    return f"complex N={n:6d} batch={batch:5d} fwd rel-L2 {e_f:.2e}  roundtrip max-abs {e_b:.2e}  in-place diff {e_ip:.1e}"


def check_real(n, batch, dev):
    rng = np.random.default_rng(n + 1)
    x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    s = api.fft_new_setup(n, api.FFT_REAL)
    if s is None:
        return f"real N={n}: setup NULL"
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty_like(xd)
    try:
        api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD)
    except api.FFTError as e:
        return f"real N={n}: {e}"
    torch.cuda.synchronize()
    y = yd.cpu().numpy().astype(np.float64)
    ref = packed_real(np.fft.rfft(x.astype(np.float64), axis=1))
    e_f = max(rel_l2(y[i], ref[i]) for i in range(batch))
    zd = torch.empty_like(xd)
    api.fft_transform_batch(s, yd, zd, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    e_b = float(np.abs(zd.cpu().numpy() / n - x).max())
    api.fft_destroy_setup(s)
    return f"real    N={n:6d} batch={batch:5d} fwd rel-L2 {e_f:.2e}  roundtrip max-abs {e_b:.2e}"


def time_transform(n, batch, transform, direction, dev, iters=20):
    s = api.fft_new_setup(n, transform)
    if s is None:
        return None
    frame = 2 * n if transform == api.FFT_COMPLEX else n
    x = torch.rand((batch, frame), device=dev) * 2 - 1
    y = torch.empty_like(x)
    try:
        for _ in range(3):
            api.fft_transform_batch(s, x, y, direction)
    except api.FFTError as e:
        return str(e)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        api.fft_transform_batch(s, x, y, direction)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gbs = 2 * x.numel() * 4 / ms * 1e-6
    api.fft_destroy_setup(s)
    return ms, gbs, batch / ms * 1e-3  # ms, GB/s, MFFT/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="16,32,64,128,256,512,1024,2048,4096")
    ap.add_argument("--real-sizes", default="")
    ap.add_argument("--time", default="4096:65536,1024:262144,256:1048576,64:4194304")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print("torch", torch.__version__, "device", torch.cuda.get_device_name(0), flush=True)
    for n in [int(v) for v in args.sizes.split(",") if v]:
        print(check_complex(n, 37, dev), flush=True)
    for n in [int(v) for v in args.real_sizes.split(",") if v]:
        print(check_real(n, 37, dev), flush=True)
    for item in [v for v in args.time.split(",") if v]:
        n, b = (int(v) for v in item.split(":"))
        for tr, nm in ((api.FFT_COMPLEX, "c2c"), (api.FFT_REAL, "real")):
            for d, dn in ((api.FFT_FORWARD, "fwd"), (api.FFT_BACKWARD, "bwd")):
                r = time_transform(n, b, tr, d, dev)
                if isinstance(r, tuple):
                    print(f"time {nm} {dn} N={n} batch={b}: {r[0]:.3f} ms  {r[1]:.0f} GB/s  {r[2]:.2f} MFFT/s", flush=True)
                else:
                    print(f"time {nm} {dn} N={n}: {r}", flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"done in {time.time() - t0:.1f}s")
