"""fft_transform_batch_strided -- frames embedded in larger buffers.

The reference's counterpart is the caller's own loop over fft_transform with advancing pointers, so the expected
result of a strided call is, frame for frame, the oracle's transform of that frame (1e-6), the floats between the
frames must stay untouched, and it must agree with the packed call to the last bits (same algorithm).
Every execution path is covered: LDS-staged narrow teams, register-direct teams, the software-pipelined
one-frame-per-workgroup kernels, the two-pass kernels (generic tiles and the 256 x 256 fast path), long real frames
and the mixed-radix kernel.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu

SENTINEL = -77.25

CASES = [
    # (N, transform, batch)
    (16, COMPLEX, 301), (64, COMPLEX, 77), (128, COMPLEX, 45), (256, COMPLEX, 37), (1024, COMPLEX, 19),
    (2048, COMPLEX, 7), (4096, COMPLEX, 9), (8192, COMPLEX, 5), (16384, COMPLEX, 3), (32768, COMPLEX, 3),
    (65536, COMPLEX, 3), (480, COMPLEX, 21), (3072, COMPLEX, 5), (12288, COMPLEX, 3), (24576, COMPLEX, 3),
    (9216, COMPLEX, 4),
    (96, COMPLEX, 45), (160, COMPLEX, 29),   # lengths whose PACKED batches take the staged-I/O path (generic_kernels.hpp): strided ones must not
    (32, REAL, 301), (64, REAL, 131), (256, REAL, 37), (1024, REAL, 21), (4096, REAL, 7), (8192, REAL, 5),
    (16384, REAL, 3), (32768, REAL, 3), (65536, REAL, 2), (131072, REAL, 2), (480, REAL, 13), (960, REAL, 9), (9216, REAL, 5), (49152, REAL, 2),
    (192, REAL, 45), (288, REAL, 29),
]


def _embed(torch, dev, frames, stride):
    """[batch, nf] numpy -> device buffer of batch*stride floats, frames at multiples of `stride`, sentinel between"""
    batch, nf = frames.shape
    buf = torch.full((batch, stride), SENTINEL, device=dev, dtype=torch.float32)
    buf[:, :nf] = torch.from_numpy(frames).to(dev)
    return buf


@pytest.mark.parametrize("n,tr,batch", CASES)
@pytest.mark.parametrize("direction", [FORWARD, BACKWARD])
def test_strided_equals_packed_and_oracle(gpu, oracle, n, tr, batch, direction):
    api, torch, dev = gpu
    nf = 2 * n if tr == COMPLEX else n
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    assert s is not None
    x = ol.uniform(batch * nf, 31 + n + direction).reshape(batch, nf)
    if tr == REAL and direction == BACKWARD:
        x = oracle.transform_batch(so, x, FORWARD) / np.float32(n)  # a valid packed spectrum
        if n % 25 == 0:
            pytest.skip("reference real forward defect (tests/test_gpu_other_sizes.py)")
    in_stride, out_stride = nf + 4, 2 * nf + 8
    xin = _embed(torch, dev, x, in_stride)
    yout = torch.full((batch, out_stride), SENTINEL, device=dev, dtype=torch.float32)
    api.fft_transform_batch_strided(s, xin, in_stride, yout, out_stride, batch, direction)
    # packed call on the same frames
    xp = torch.from_numpy(x).to(dev)
    yp = torch.empty_like(xp)
    api.fft_transform_batch(s, xp, yp, direction)
    torch.cuda.synchronize()
    got = yout.cpu().numpy()
    # same algorithm; where the packed call takes the software-pipelined kernel variant the compiler may contract
    # multiply-adds differently, so allow last-bit differences (far inside the 1e-6 parity bound)
    # the strided and the packed instantiation of one kernel: same algorithm, the compiler may contract differently
    assert_frames_close(got[:, :nf], yp.cpu().numpy(), tol=2e-7, maxabs_tol=4e-7, what=f"N={n} strided vs packed")
    assert np.all(got[:, nf:] == SENTINEL), "floats between the output frames were written"
    assert np.all(xin.cpu().numpy()[:, nf:] == SENTINEL)
    idx = np.unique(np.concatenate([np.arange(0, batch, max(1, batch // 9)), [batch - 1]]))
    assert_frames_close(got[idx, :nf], oracle.transform_batch(so, x[idx], direction), what=f"N={n} strided")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("n,tr,batch", [(64, COMPLEX, 50), (4096, COMPLEX, 6), (65536, COMPLEX, 3), (1024, REAL, 33),
                                        (16384, REAL, 3), (480, REAL, 10)])
def test_strided_in_place(gpu, oracle, n, tr, batch):
    """the JUCE adapter's shape: one 2N-float buffer per frame, transformed in place (chowdsp_fft_juce.cpp:47-55)"""
    api, torch, dev = gpu
    nf = 2 * n if tr == COMPLEX else n
    stride = 2 * nf
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    x = ol.uniform(batch * nf, 3 + n).reshape(batch, nf)
    buf = _embed(torch, dev, x, stride)
    api.fft_transform_batch_strided(s, buf, stride, buf, stride, batch, api.FFT_FORWARD)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.all(got[:, nf:] == SENTINEL)
    want = oracle.transform_batch(so, x, FORWARD)
    assert_frames_close(got[:, :nf], want, what=f"N={n} in place fwd")
    api.fft_transform_batch_strided(s, buf, stride, buf, stride, batch, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    back = buf.cpu().numpy()
    assert np.all(back[:, nf:] == SENTINEL)
    assert np.abs(back[:, :nf] / n - x).max() <= 3e-6
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_strided_argument_checks(gpu):
    api, torch, dev = gpu
    n = 256
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    buf = torch.zeros(8 * 1024, device=dev)
    out = torch.zeros(8 * 1024, device=dev)
    with pytest.raises(api.FFTError):  # stride shorter than a frame
        api.fft_transform_batch_strided(s, buf, 2 * n - 4, out, 2 * n, 4, api.FFT_FORWARD)
    with pytest.raises(api.FFTError):  # not a multiple of 4 floats
        api.fft_transform_batch_strided(s, buf, 2 * n + 2, out, 2 * n, 4, api.FFT_FORWARD)
    with pytest.raises(api.FFTError):  # in place with different strides
        api.fft_transform_batch_strided(s, buf, 2 * n, buf, 2 * n + 4, 4, api.FFT_FORWARD)
    with pytest.raises(api.FFTError):  # partial overlap
        api.fft_transform_batch_strided(s, buf, 2 * n, buf[2 * n:], 2 * n, 4, api.FFT_FORWARD)
    api.fft_transform_batch_strided(s, buf, 2 * n, out, 2 * n, 0, api.FFT_FORWARD)  # empty batch is fine
    api.fft_destroy_setup(s)
