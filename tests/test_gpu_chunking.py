"""Chunking invariance of the two-pass plans: a batch whose intermediate exceeds option "workspace_mib" is worked through in
equal chunks, one after the other on the caller's stream (capi.cpp big_c2c_chunked / big_real4_chunked).  Every frame is
transformed on its own, so the results must be BIT-IDENTICAL for every cap -- one chunk, a few, one frame per chunk -- and equal
to the oracle's (reference: one pffft_transform_ordered call per frame, /root/reference/chowdsp_fft.h:117-134).

Covers each chunked family: register-direct tiles (complex 2^19, real four-step 2^19 / 2^20), the radix-2 split (2^20 complex is
not: tiles), the long mixed-radix complex plan, the real four-step on generic cores, the two-pass + pair-stage real form (2^21),
and the three-sweep / product-on-the-stores fused convolutions of the same plans.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import MIXED_RADIX_MAXABS_TOL, assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu

# (n, transform, batch, caps in MiB): frame sizes 0.3 ... 8 MiB, so the small caps give 3 ... `batch` chunks
CASES = [(1 << 19, COMPLEX, 5, (4, 9)), (1 << 19, REAL, 9, (3, 8)), (1 << 20, REAL, 5, (9, 18)), (46080, COMPLEX, 41, (1, 3)),
         (41472, REAL, 50, (1, 2)), (96000, REAL, 7, (1, 2)), (1 << 21, REAL, 3, (16, 32))]


@pytest.fixture
def cap(gpu):
    api = gpu[0]
    before = api.get_option("workspace_mib")
    assert before == 256 or before > 0
    yield lambda mib: api.set_option("workspace_mib", mib)
    api.set_option("workspace_mib", before)


@pytest.mark.parametrize("n,tr,batch,caps", CASES)
def test_results_do_not_depend_on_the_chunking(gpu, oracle, cap, n, tr, batch, caps):
    api, torch, dev = gpu
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(batch * nf, 4000 + batch).reshape(batch, nf)
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    xd = torch.from_numpy(x).to(dev)
    ref_f = ref_b = None
    for mib in (256,) + tuple(caps):
        cap(mib)
        assert api.get_option("workspace_mib") == mib
        yd, zd = torch.empty_like(xd), torch.empty_like(xd)
        api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD)
        api.fft_transform_batch(s, yd, zd, api.FFT_BACKWARD)
        torch.cuda.synchronize()
        f, b = yd.cpu().numpy(), zd.cpu().numpy()
        if ref_f is None:
            ref_f, ref_b = f, b
            mt = MIXED_RADIX_MAXABS_TOL if n % 3 == 0 or n % 5 == 0 else None
            if tr == REAL and n % 25 == 0:
                # the reference's real radix-5 forward pass is wrong where N % 25 == 0 (tests/test_gpu_other_sizes.py): float64 instead
                c = np.fft.rfft(x.astype(np.float64), axis=1)
                want = np.empty((batch, n))
                want[:, 0], want[:, 1] = c[:, 0].real, c[:, n // 2].real
                want[:, 2::2], want[:, 3::2] = c[:, 1 : n // 2].real, c[:, 1 : n // 2].imag
            else:
                want = oracle.transform_batch(so, x, FORWARD)
            assert_frames_close(f, want, what=f"fwd N={n}", maxabs_tol=mt)
            assert_frames_close(b / n, x, what=f"round trip N={n}")
        else:
            assert np.array_equal(f, ref_f), f"forward differs at workspace_mib={mib}"
            assert np.array_equal(b, ref_b), f"backward differs at workspace_mib={mib}"
    # in place, at the smallest cap: a chunk's column sweep has read the chunk's frames before its row sweep overwrites them
    w = xd.clone()
    api.fft_transform_batch(s, w, w, api.FFT_FORWARD)
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy(), ref_f), "in-place forward differs"
    api.fft_transform_batch(s, w, w, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy(), ref_b), "in-place backward differs"
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("n,tr,batch,caps", [(1 << 19, COMPLEX, 5, (4,)), (1 << 19, REAL, 9, (3,)), (46080, COMPLEX, 41, (1,)), (41472, REAL, 50, (1,))])
def test_fused_convolution_does_not_depend_on_the_chunking(gpu, cap, n, tr, batch, caps):
    """fft_convolve_frames_batch of the chunked plans (reference call sequence test/test.cpp:212-218) at several caps == itself at
    256 MiB, bit for bit, and == the float64 circular convolution"""
    api, torch, dev = gpu
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(batch * nf, 4100 + batch).reshape(batch, nf)
    h = ol.uniform(nf, 4200).reshape(1, nf)
    s = api.fft_new_setup(n, tr)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)  # the filter spectrum in the library's own layout
    ref = None
    for mib in (256,) + tuple(caps):
        cap(mib)
        yd = torch.empty_like(xd)
        api.fft_convolve_frames_batch(s, xd, H, yd, 1.0 / n)
        torch.cuda.synchronize()
        y = yd.cpu().numpy()
        if ref is None:
            ref = y
            # float64: circular convolution of the first and the last frame with h
            pick = [0, batch - 1]
            if tr == COMPLEX:
                xc = x[pick].astype(np.float64).view(np.complex128)
                want = np.fft.ifft(np.fft.fft(xc, axis=1) * np.fft.fft(h.astype(np.float64).view(np.complex128), axis=1), axis=1).view(np.float64)
            else:
                want = np.fft.irfft(np.fft.rfft(x[pick].astype(np.float64), axis=1) * np.fft.rfft(h.astype(np.float64), axis=1), n=n, axis=1)
            assert_frames_close(y[pick], want, what=f"fused convolution N={n} vs float64 circular convolution", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
        else:
            assert np.array_equal(y, ref), f"fused convolution differs at workspace_mib={mib}"
    api.fft_destroy_setup(s)


def test_two_launch_form_of_65536_on_its_two_lanes(gpu, oracle, cap):
    """N = 65536 complex with option fused_two_pass = 0: two launches per chunk through the LDS-image 256 x 256 passes -- the one family
    whose chunks still alternate between two lanes by default (capi.cpp lane_count) -- one chunk, a few, many: bit-identical, and equal
    to the one-launch form's result within the parity bound and to the oracle's"""
    api, torch, dev = gpu
    n, batch = 65536, 40
    x = ol.uniform(batch * 2 * n, 4300).reshape(batch, 2 * n)
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    xd = torch.from_numpy(x).to(dev)
    before = api.get_option("fused_two_pass")
    api.set_option("fused_two_pass", 0)
    try:
        ref = None
        for mib in (256, 6, 1):
            cap(mib)
            yd, zd = torch.empty_like(xd), torch.empty_like(xd)
            api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD)
            api.fft_transform_batch(s, yd, zd, api.FFT_BACKWARD)
            torch.cuda.synchronize()
            f, b = yd.cpu().numpy(), zd.cpu().numpy()
            if ref is None:
                ref = (f, b)
                assert_frames_close(f, oracle.transform_batch(so, x, FORWARD), what="two-launch 65536 fwd")
                assert_frames_close(b / n, x, what="two-launch 65536 round trip")
            else:
                assert np.array_equal(f, ref[0]) and np.array_equal(b, ref[1]), f"differs at workspace_mib={mib}"
    finally:
        api.set_option("fused_two_pass", before)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
