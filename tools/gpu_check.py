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


def check_conv(n, transform, batch, dev):
    """unfused (3 API calls) and fused convolution vs float64 circular convolution"""
    rng = np.random.default_rng(7 * n + transform)
    frame = 2 * n if transform == api.FFT_COMPLEX else n
    x = rng.uniform(-1, 1, (batch, frame)).astype(np.float32)
    h = rng.uniform(-1, 1, (1, frame)).astype(np.float32)
    s = api.fft_new_setup(n, transform)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    X, H = torch.empty_like(xd), torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, xd, X, api.FFT_FORWARD)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    acc = torch.zeros_like(xd)
    api.fft_convolve_unordered_batch(s, X, H, acc, 1.0 / n, b_is_shared=True)
    y = torch.empty_like(xd)
    api.fft_transform_unordered_batch(s, acc, y, api.FFT_BACKWARD)
    msg = ""
    try:
        y2 = torch.empty_like(xd)
        api.fft_convolve_frames_batch(s, xd, H, y2, 1.0 / n)
        torch.cuda.synchronize()
        fused = y2.cpu().numpy().astype(np.float64)
    except api.FFTError as e:
        fused = None
        msg = f" fused: {e}"
    torch.cuda.synchronize()
    got = y.cpu().numpy().astype(np.float64)
    if transform == api.FFT_COMPLEX:
        xc = x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2]
        hc = h[:, 0::2].astype(np.float64) + 1j * h[:, 1::2]
        r = np.fft.ifft(np.fft.fft(xc, axis=1) * np.fft.fft(hc, axis=1), axis=1)
        ref = np.empty((batch, frame)); ref[:, 0::2] = r.real; ref[:, 1::2] = r.imag
    else:
        ref = np.fft.irfft(np.fft.rfft(x.astype(np.float64), axis=1) * np.fft.rfft(h.astype(np.float64), axis=1), n=n, axis=1)
    e1 = max(rel_l2(got[i], ref[i]) for i in range(batch))
    e2 = max(rel_l2(fused[i], ref[i]) for i in range(batch)) if fused is not None else float("nan")
    api.fft_destroy_setup(s)
    kind = "complex" if transform == api.FFT_COMPLEX else "real   "
    return f"conv {kind} N={n:6d} unfused rel-L2 {e1:.2e}  fused rel-L2 {e2:.2e}{msg}"


def check_legacy(n, transform):
    """the eleven-symbol single-frame API with HOST pointers (reference test/test.cpp call sequence)"""
    frame = 2 * n if transform == api.FFT_COMPLEX else n
    x, xp = api.aligned_array(frame)
    w, wp = api.aligned_array(frame)
    i = np.arange(n, dtype=np.float32)
    if transform == api.FFT_COMPLEX:
        x[0::2] = np.sin(np.float32(3.14) * np.float32(100.0 / 48000.0) * i)
        x[1::2] = np.cos(np.float32(3.14) * np.float32(100.0 / 48000.0) * i)
    else:
        x[:] = np.sin(np.float32(3.14) * np.float32(100.0 / 48000.0) * i)
    x0 = x.copy()
    s = api.fft_new_setup(n, transform, False)
    assert api.fft_simd_width_bytes(s) == 16
    api.fft_transform(s, x, x, w, api.FFT_FORWARD)  # in place, like the reference test
    if transform == api.FFT_COMPLEX:
        ref = np.fft.fft(x0[0::2].astype(np.float64) + 1j * x0[1::2])
        got = x[0::2].astype(np.float64) + 1j * x[1::2]
        e_f = rel_l2(got, ref)
    else:
        ref = packed_real(np.fft.rfft(x0.astype(np.float64))[None, :])[0]
        e_f = rel_l2(x.astype(np.float64), ref)
    api.fft_transform(s, x, x, w, api.FFT_BACKWARD)
    e_b = float(np.abs(x / n - x0).max())
    api.fft_destroy_setup(s)
    api.aligned_free(xp); api.aligned_free(wp)
    kind = "complex" if transform == api.FFT_COMPLEX else "real   "
    return f"legacy host-pointer {kind} N={n:6d} fwd rel-L2 {e_f:.2e} roundtrip max-abs {e_b:.2e}"


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
    ap.add_argument("--real-sizes", default="32,64,128,256,512,1024,2048,4096,8192")
    ap.add_argument("--big-sizes", default="8192,16384,65536,262144,524288,1048576")
    ap.add_argument("--big-real-sizes", default="16384,65536,1048576")
    ap.add_argument("--time", default="4096:65536,1024:262144,256:1048576,64:4194304,65536:4096")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print("torch", torch.__version__, "device", torch.cuda.get_device_name(0), flush=True)
    for n in [int(v) for v in args.sizes.split(",") if v]:
        print(check_complex(n, 37, dev), flush=True)
    for n in [int(v) for v in args.real_sizes.split(",") if v]:
        print(check_real(n, 37, dev), flush=True)
    for n in [int(v) for v in args.big_sizes.split(",") if v]:
        print(check_complex(n, 3, dev), flush=True)
    for n in [int(v) for v in args.big_real_sizes.split(",") if v]:
        print(check_real(n, 3, dev), flush=True)
    for n in (16, 64, 1024, 4096):
        print(check_conv(n, api.FFT_COMPLEX, 5, dev), flush=True)
    for n in (32, 64, 1024, 4096, 8192):
        print(check_conv(n, api.FFT_REAL, 5, dev), flush=True)
    for n in (32, 1024, 4096, 65536):
        print(check_legacy(n, api.FFT_COMPLEX), flush=True)
        print(check_legacy(n, api.FFT_REAL), flush=True)
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
