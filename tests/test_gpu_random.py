"""Randomised differential test of the batched entry points: random supported size, batch count, direction, packed /
strided / in-place call shape, checked frame by frame against numpy's float64 FFT (the oracle pins the algorithm
elsewhere; this one hunts indexing mistakes in ragged groups, staging areas, strides and plan boundaries).
Deterministic: fixed seeds, 240 cases.
"""
import numpy as np
import pytest

from oracle_lib import COMPLEX, REAL

pytestmark = pytest.mark.gpu

POW2 = [1 << k for k in range(4, 18)]
MIXED = [48, 96, 160, 192, 240, 384, 480, 640, 768, 960, 1536, 2000, 3072, 5120, 9216, 12288, 20480, 24576]


def _exact_forward(x, n, tr):
    if tr == COMPLEX:
        z = x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2]
        f = np.fft.fft(z, axis=1)
        out = np.empty_like(x, dtype=np.float64)
        out[:, 0::2], out[:, 1::2] = f.real, f.imag
        return out
    f = np.fft.rfft(x.astype(np.float64), axis=1)
    out = np.empty_like(x, dtype=np.float64)
    out[:, 0], out[:, 1] = f[:, 0].real, f[:, n // 2].real
    out[:, 2::2], out[:, 3::2] = f[:, 1:n // 2].real, f[:, 1:n // 2].imag
    return out


def _exact_backward(spec, n, tr):
    """unscaled inverse of the layouts above (BACKWARD(FORWARD(x)) = N x)"""
    if tr == COMPLEX:
        z = spec[:, 0::2].astype(np.float64) + 1j * spec[:, 1::2]
        f = np.fft.ifft(z, axis=1) * n
        out = np.empty_like(spec, dtype=np.float64)
        out[:, 0::2], out[:, 1::2] = f.real, f.imag
        return out
    half = np.empty((spec.shape[0], n // 2 + 1), dtype=np.complex128)
    half[:, 0], half[:, -1] = spec[:, 0], spec[:, 1]
    half[:, 1:-1] = spec[:, 2::2] + 1j * spec[:, 3::2]
    return np.fft.irfft(half, n=n, axis=1) * n


@pytest.mark.parametrize("seed", range(24))
def test_random_call_shapes(gpu, seed):
    api, torch, dev = gpu
    rng = np.random.RandomState(1000 + seed)
    for _ in range(10):
        tr = REAL if rng.rand() < 0.5 else COMPLEX
        n = int(rng.choice(POW2 if rng.rand() < 0.7 else MIXED))
        if tr == REAL and (n % 32 != 0):
            n *= 2
        if not api.size_supported(n, tr):
            continue
        nf = 2 * n if tr == COMPLEX else n
        batch = int(rng.choice([1, 2, 3, 5, 17, 31, 64, 129])) if n <= 4096 else int(rng.choice([1, 2, 3, 5]))
        backward = rng.rand() < 0.5
        shape = rng.choice(["packed", "packed_inplace", "strided", "strided_inplace"])
        x = rng.uniform(-1, 1, (batch, nf)).astype(np.float32)
        if backward:
            src = _exact_forward(x, n, tr).astype(np.float32)  # a valid spectrum (matters for REAL)
            want = _exact_backward(src, n, tr)
        else:
            src = x
            want = _exact_forward(x, n, tr)
        direction = api.FFT_BACKWARD if backward else api.FFT_FORWARD
        s = api.fft_new_setup(n, tr)
        assert s is not None
        pad = int(rng.choice([4, 8, 64, nf]))
        stride = nf + pad if shape.startswith("strided") else nf
        buf = torch.full((batch, stride), 123.0, device=dev)
        buf[:, :nf] = torch.from_numpy(src).to(dev)
        out = buf if shape.endswith("inplace") else torch.full((batch, stride), 321.0, device=dev)
        if shape.startswith("strided"):
            api.fft_transform_batch_strided(s, buf, stride, out, stride, batch, direction)
        else:
            api.fft_transform_batch(s, buf, out, direction, batch=batch)
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        err = np.linalg.norm(got[:, :nf] - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
        what = f"seed {seed}: N={n} {'real' if tr == REAL else 'complex'} batch={batch} {'bwd' if backward else 'fwd'} {shape} stride={stride}"
        assert err.max() <= 1.0e-6, f"{what}: relative L2 {err.max():.3e}"
        if stride > nf:
            fill = 123.0 if shape.endswith("inplace") else 321.0
            assert np.all(got[:, nf:] == fill), f"{what}: padding between frames was written"
        api.fft_destroy_setup(s)


@pytest.mark.parametrize("seed", range(12))
def test_random_fused_convolutions(gpu, seed):
    """fft_convolve_frames_batch over random supported sizes (every fused form: LDS-resident incl. the round-4 kernels -- pair
    stage in registers, transposed backward network, staged narrow teams --, split-plane, mixed radix, two-pass), random batch
    counts (ragged groups, fewer frames than resident workgroups), in place or not, against the float64 circular convolution
    (the reference's forward -> convolve -> backward sequence, test/test.cpp:161-166, computes exactly that times N * scaling)."""
    api, torch, dev = gpu
    rng = np.random.RandomState(7000 + seed)
    for _ in range(8):
        tr = REAL if rng.rand() < 0.5 else COMPLEX
        n = int(rng.choice([16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384] if rng.rand() < 0.75 else [96, 480, 768, 2400, 6000, 9216, 65536]))
        if tr == REAL and (n % 32 != 0):
            n *= 2
        if not api.size_supported(n, tr):
            continue
        nf = 2 * n if tr == COMPLEX else n
        batch = int(rng.choice([1, 2, 3, 7, 33, 130, 517])) if n <= 4096 else int(rng.choice([1, 2, 5]))
        x = rng.uniform(-1, 1, (batch, nf)).astype(np.float32)
        h = rng.uniform(-1, 1, (1, nf)).astype(np.float32)
        if tr == COMPLEX:
            xz = x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2]
            hz = h[:, 0::2].astype(np.float64) + 1j * h[:, 1::2]
            yz = np.fft.ifft(np.fft.fft(xz, axis=1) * np.fft.fft(hz, axis=1), axis=1)
            want = np.empty((batch, nf))
            want[:, 0::2], want[:, 1::2] = yz.real, yz.imag
        else:
            want = np.fft.irfft(np.fft.rfft(x.astype(np.float64), axis=1) * np.fft.rfft(h.astype(np.float64), axis=1), n=n, axis=1)
        s = api.fft_new_setup(n, tr)
        xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
        H = torch.empty_like(hd)
        api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
        inplace = rng.rand() < 0.5
        yd = xd if inplace else torch.full_like(xd, 55.0)
        api.fft_convolve_frames_batch(s, xd, H, yd, 1.0 / n)
        api.stream_synchronize()
        got = yd.cpu().numpy().astype(np.float64)
        err = np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
        assert err.max() <= 1.5e-6, f"seed {seed}: N={n} {'real' if tr == REAL else 'complex'} batch={batch} inplace={inplace}: relative L2 {err.max():.3e}"
        api.fft_destroy_setup(s)
