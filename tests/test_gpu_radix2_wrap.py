"""Complex 2^19 .. 2^21 points (and the real frames of 2^21 / 2^22 samples on those cores) as ONE radix-2 decimation in time around
the register-direct two-pass transform of the two half-length sub-frames (csrc/tile_fast_kernels.hpp pass_*_fast_r2_kernel): parity with
the oracle in both directions, strided and in place, ragged batches (fewer tiles than workgroups / several chunks), and
fft_convolve_frames_batch (the product riding on the forward transform's stores) against the three-call composition and the float64
circular convolution.  Reference: cfftf1_ps plans any even N with a radix-2 pass (simd/chowdsp_fft_impl_sse.cpp:372-432, passf2_ps
:98-137); test/test.cpp:270-277 runs complex sizes up to 2^19.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


def _gpu(gpu, s, x, direction):
    api, torch, dev = gpu
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    yd = torch.full_like(xd, -4.0)
    api.fft_transform_batch(s, xd, yd, direction)
    api.stream_synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("lg,batch", [(19, 1), (19, 5), (20, 1), (20, 3), (21, 1), (21, 2), (21, 9)])
def test_complex_vs_oracle(gpu, oracle, lg, batch):
    api = gpu[0]
    n = 1 << lg
    d = api.plan_describe(n, api.FFT_COMPLEX)
    assert d["passes"] == 2
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    x = ol.uniform(batch * 2 * n, 5200 + lg + batch).reshape(batch, 2 * n)
    pick = sorted({0, batch - 1})
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        got = _gpu(gpu, s, x, direction)
        want = oracle.transform_batch(so, x[pick], odir)
        assert_frames_close(got[pick], want, what=f"complex 2^{lg} batch {batch} direction {direction}")
    # every frame of the batch is the same transform (tiles of all frames share the persistent loop)
    rt = _gpu(gpu, s, _gpu(gpu, s, x, api.FFT_FORWARD), api.FFT_BACKWARD)
    assert np.max(np.abs(rt / n - x)) < 4e-6
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("lg", [21, 22])
def test_real_frames_on_these_cores(gpu, oracle, lg):
    api = gpu[0]
    n = 1 << lg
    s, so = api.fft_new_setup(n, api.FFT_REAL), oracle.new_setup(n, REAL)
    x = ol.uniform(2 * n, 61 + lg).reshape(2, n)
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        assert_frames_close(_gpu(gpu, s, x, direction), oracle.transform_batch(so, x, odir), what=f"real 2^{lg} direction {direction}")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_in_place_and_strided(gpu, oracle):
    api, torch, dev = gpu
    n, batch, pad = 1 << 21, 3, 32
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    x = ol.uniform(batch * 2 * n, 99).reshape(batch, 2 * n)
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        want = oracle.transform_batch(so, x[[0, 2]], odir)
        xd = torch.from_numpy(x).to(dev)
        api.fft_transform_batch(s, xd, xd, direction)
        api.stream_synchronize()
        assert_frames_close(xd.cpu().numpy()[[0, 2]], want, what="in place")
        stride = 2 * n + pad
        buf = torch.full((batch, stride), 11.0, device=dev)
        buf[:, :2 * n] = torch.from_numpy(x).to(dev)
        out = torch.full((batch, stride), -7.0, device=dev)
        api.fft_transform_batch_strided(s, buf, stride, out, stride, batch, direction)
        api.stream_synchronize()
        o = out.cpu().numpy()
        assert_frames_close(o[[0, 2], :2 * n], want, what="strided")
        assert np.all(o[:, 2 * n:] == -7.0), "padding between output frames untouched"
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("lg,batch", [(19, 3), (20, 2), (21, 1), (21, 3)])
def test_fused_convolution(gpu, lg, batch):
    """one call (the product on the forward transform's pass-B stores) == the composition == the float64 circular convolution"""
    api, torch, dev = gpu
    n = 1 << lg
    rng = np.random.RandomState(300 + lg + batch)
    x = rng.uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    h = rng.uniform(-1, 1, (1, 2 * n)).astype(np.float32)
    xz = x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2]
    hz = h[:, 0::2].astype(np.float64) + 1j * h[:, 1::2]
    yz = np.fft.ifft(np.fft.fft(xz, axis=1) * np.fft.fft(hz, axis=1), axis=1)
    want = np.empty((batch, 2 * n))
    want[:, 0::2], want[:, 1::2] = yz.real, yz.imag
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    one = torch.full_like(xd, 3.0)
    api.fft_convolve_frames_batch(s, xd, H, one, 1.0 / n)
    api.stream_synchronize()
    api.set_option("fused_conv_group", 0)
    try:
        three = torch.full_like(xd, 5.0)
        api.fft_convolve_frames_batch(s, xd, H, three, 1.0 / n)
        api.stream_synchronize()
    finally:
        api.set_option("fused_conv_group", 1)
    got = one.cpu().numpy().astype(np.float64)
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    assert err.max() <= 1.5e-6, f"2^{lg}: relative L2 {err.max():.3e}"
    assert_frames_close(one.cpu().numpy(), three.cpu().numpy(), tol=4e-7, maxabs_tol=2e-6, what="one call vs composition")
    api.fft_destroy_setup(s)
