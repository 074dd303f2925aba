"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties
(the oracle would need minutes per configuration at these sizes):

  * BACKWARD(FORWARD(x)) = N x                 (reference chowdsp_fft.h:128-129)
  * linearity: F(a x + y) = a F(x) + F(y)
  * Parseval: sum |X|^2 = N sum |x|^2, per frame
  * every K-th frame spot-checked against the oracle frame by frame (1e-6 bound)
  * fused convolution == the three-call sequence
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


def _rand(torch, dev, shape, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.rand(shape, device=dev, generator=g) * 2 - 1


def _spot_check(gpu, oracle, n, tr, x, y, frames):
    so = oracle.new_setup(n, tr)
    idx = np.array(frames)
    want = oracle.transform_batch(so, x[idx].cpu().numpy(), FORWARD)
    assert_frames_close(y[idx].cpu().numpy(), want, what=f"spot frames N={n}")
    oracle.destroy_setup(so)


def test_config2_c2c_4096_batch_65536(gpu, oracle):
    """BASELINE configs[1]: N=4096 fp32 complex fwd+inv, batch=65536"""
    api, torch, dev = gpu
    n, batch = 4096, 65536
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    x = _rand(torch, dev, (batch, 2 * n), 1234)
    y = torch.empty_like(x)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    _spot_check(gpu, oracle, n, COMPLEX, x, y, [0, 1, 255, 256, 4097, 32768, 65534, 65535])
    # Parseval per frame
    ex = (x.double() ** 2).sum(dim=1)
    ey = (y.double() ** 2).sum(dim=1)
    assert float(((ey / (n * ex)) - 1).abs().max()) < 1e-5
    # round trip
    z = torch.empty_like(x)
    api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    err = (z / n - x).abs().amax(dim=1)
    assert float(err.max()) < 2e-6
    # linearity on a slice of the batch
    a = 0.37
    x2 = _rand(torch, dev, (1024, 2 * n), 99)
    lhs = torch.empty_like(x2)
    api.fft_transform_batch(s, a * x[:1024] + x2, lhs, api.FFT_FORWARD)
    y2 = torch.empty_like(x2)
    api.fft_transform_batch(s, x2, y2, api.FFT_FORWARD)
    rhs = a * y[:1024] + y2
    rel = (lhs - rhs).norm(dim=1) / rhs.norm(dim=1)
    assert float(rel.max()) < 1e-6
    api.fft_destroy_setup(s)


def test_config3_real_1024_batch_262144(gpu, oracle):
    """BASELINE configs[2]: N=1024 fp32 real (r2c / c2r), batch=262144"""
    api, torch, dev = gpu
    n, batch = 1024, 262144
    s = api.fft_new_setup(n, api.FFT_REAL)
    x = _rand(torch, dev, (batch, n), 4321)
    y = torch.empty_like(x)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    _spot_check(gpu, oracle, n, REAL, x, y, [0, 3, 4, 1023, 100000, 262143])
    # Parseval for the packed half spectrum: |X0|^2 + |X(N/2)|^2 + 2 sum_{0<k<N/2} |Xk|^2 = N sum x^2
    yd = y.double()
    e = yd[:, 0] ** 2 + yd[:, 1] ** 2 + 2 * (yd[:, 2:] ** 2).sum(dim=1)
    assert float((e / (n * (x.double() ** 2).sum(dim=1)) - 1).abs().max()) < 1e-5
    z = torch.empty_like(x)
    api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    assert float((z / n - x).abs().max()) < 2e-6
    api.fft_destroy_setup(s)


def test_config4_overlap_save_frames_4096(gpu, oracle):
    """BASELINE configs[3], one GPU's share: N=4096 real frames, 131072 frames, one shared filter;
    fused kernel == three-call sequence, and every 10000th frame == oracle pipeline"""
    api, torch, dev = gpu
    n, batch = 4096, 131072
    s = api.fft_new_setup(n, api.FFT_REAL)
    x = _rand(torch, dev, (batch, n), 777)
    h = _rand(torch, dev, (1, n), 778)
    H = torch.empty_like(h)
    api.fft_transform_unordered_batch(s, h, H, api.FFT_FORWARD)
    y = torch.empty_like(x)
    api.fft_convolve_frames_batch(s, x, H, y, 1.0 / n)
    # three-call sequence on a slice
    sl = slice(0, 4096)
    X = torch.empty_like(x[sl])
    api.fft_transform_unordered_batch(s, x[sl].contiguous(), X, api.FFT_FORWARD)
    acc = torch.zeros_like(X)
    api.fft_convolve_unordered_batch(s, X, H, acc, 1.0 / n, b_is_shared=True)
    y3 = torch.empty_like(X)
    api.fft_transform_unordered_batch(s, acc, y3, api.FFT_BACKWARD)
    rel = (y[sl] - y3).norm(dim=1) / y3.norm(dim=1)
    assert float(rel.max()) < 1e-6
    # oracle pipeline on a few frames
    so = oracle.new_setup(n, REAL)
    hb = oracle.transform(so, h[0].cpu().numpy(), FORWARD, False)
    idx = [0, 10000, 70000, 131071]
    want = np.stack([
        oracle.transform(so, oracle.convolve(so, oracle.transform(so, x[i].cpu().numpy(), FORWARD, False), hb,
                                             np.zeros(n, np.float32), 1.0 / n), ol.BACKWARD, False)
        for i in idx])
    assert_frames_close(y[idx].cpu().numpy(), want, what="fused overlap-save frames")
    oracle.destroy_setup(so)
    api.fft_destroy_setup(s)


def test_config5_c2c_65536_batch_8192(gpu, oracle):
    """BASELINE configs[4]: N=65536 fp32 complex forward (two-pass, HBM-resident twiddles), batch=8192"""
    api, torch, dev = gpu
    n, batch = 65536, 8192
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    x = _rand(torch, dev, (batch, 2 * n), 55)
    y = torch.empty_like(x)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    _spot_check(gpu, oracle, n, COMPLEX, x, y, [0, 255, 256, 8191])  # 256 frames = one workspace chunk boundary
    ex = (x.double() ** 2).sum(dim=1)
    ey = (y.double() ** 2).sum(dim=1)
    assert float(((ey / (n * ex)) - 1).abs().max()) < 1e-5
    api.fft_transform_batch(s, y, y, api.FFT_BACKWARD)  # in place
    assert float((y / n - x).abs().max()) < 3e-6
    api.fft_destroy_setup(s)


@pytest.mark.parametrize("n,tr,batch", [(8192, COMPLEX, 2600), (16384, REAL, 2600), (16384, COMPLEX, 1300), (32768, REAL, 1300),
                                        (32768, COMPLEX, 700), (65536, REAL, 700)])
def test_fused_convolution_on_the_split_plane_kernels_large_batches(gpu, n, tr, batch):
    """fft_convolve_frames_batch on the frame-in-registers kernels (one launch) with more frames than resident workgroups,
    so that every workgroup goes round its persistent loop several times: the one-launch result equals the three-launch
    composition (option fused_conv_split = 0) on every frame, and an all-pass filter returns the input"""
    api, torch, dev = gpu
    nf = 2 * n if tr == COMPLEX else n
    s = api.fft_new_setup(n, api.FFT_COMPLEX if tr == COMPLEX else api.FFT_REAL)
    x = _rand(torch, dev, (batch, nf), 4000 + n)
    h = _rand(torch, dev, (1, nf), 4001 + n)
    H = torch.empty_like(h)
    api.fft_transform_unordered_batch(s, h, H, api.FFT_FORWARD)
    y1 = torch.empty_like(x)
    api.fft_convolve_frames_batch(s, x, H, y1, 1.0 / n)
    api.set_option("fused_conv_split", 0)
    try:
        y3 = torch.empty_like(x)
        api.fft_convolve_frames_batch(s, x, H, y3, 1.0 / n)
    finally:
        api.set_option("fused_conv_split", 1)
    torch.cuda.synchronize()
    rel = (y1 - y3).norm(dim=1) / y3.norm(dim=1)
    assert float(rel.max()) < 1e-6
    # identity filter: the spectrum of a unit impulse at sample 0 is all ones (packed real: DC and Nyquist both 1)
    d = torch.zeros((1, nf), device=dev)
    d[0, 0] = 1.0
    D = torch.empty_like(d)
    api.fft_transform_unordered_batch(s, d, D, api.FFT_FORWARD)
    y = x.clone()
    api.fft_convolve_frames_batch(s, y, D, y, 1.0 / n)  # in place
    torch.cuda.synchronize()
    assert float((y - x).abs().max()) < 3e-6
    api.fft_destroy_setup(s)
