"""SURVEY section 8 row f3, "call coalescing": single-frame legacy calls with host pointers recorded between
fft_gpu_defer_begin / fft_gpu_defer_flush and executed as batched transforms.  The observable result must be that of the
calls executed one after the other -- including calls that depend on each other, mixed plans and directions, in-place
calls, and calls the recorder cannot take (convolve, accumulate)."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


def test_independent_frames_are_coalesced_and_match_the_oracle(gpu, oracle):
    api = gpu[0]
    plans = [(4096, COMPLEX), (1024, REAL), (480, COMPLEX), (65536, COMPLEX)]
    setups = [(api.fft_new_setup(n, tr), oracle.new_setup(n, tr), n, tr) for n, tr in plans]
    bufs = []
    api.defer_begin()
    for rep in range(6):
        for s, so, n, tr in setups:
            nf = 2 * n if tr == COMPLEX else n
            x, xp = api.aligned_array(nf)
            y, yp = api.aligned_array(nf)
            x[:] = ol.uniform(nf, 1000 * rep + n)
            y[:] = -1.0
            d = api.FFT_FORWARD if rep % 2 == 0 else api.FFT_BACKWARD
            api.fft_transform(s, x, y, None, d)
            assert np.all(y == -1.0), "a recorded call has not run yet"
            bufs.append((so, x.copy(), y, d, xp, yp))
    api.defer_flush()
    for so, x, y, d, xp, yp in bufs:
        assert_frames_close(y, oracle.transform(so, x, FORWARD if d == api.FFT_FORWARD else BACKWARD), what="coalesced call")
        api.aligned_free(xp)
        api.aligned_free(yp)
    for s, so, n, tr in setups:
        api.fft_destroy_setup(s)
        oracle.destroy_setup(so)


def test_dependent_calls_keep_their_order(gpu, oracle):
    """the reference's own usage (test/test.cpp:47-62): forward then backward IN PLACE on one buffer, then a convolution"""
    api = gpu[0]
    n = 1024
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    a, ap = api.aligned_array(2 * n)
    b, bp = api.aligned_array(2 * n)
    acc, accp = api.aligned_array(2 * n)
    a0, b0 = ol.uniform(2 * n, 1), ol.uniform(2 * n, 2)
    a[:], b[:], acc[:] = a0, b0, 0.0
    api.defer_begin()
    api.fft_transform(s, a, a, None, api.FFT_FORWARD)   # recorded
    api.fft_transform(s, b, b, None, api.FFT_FORWARD)   # recorded (independent of the first)
    api.fft_transform(s, a, a, None, api.FFT_BACKWARD)  # reads the first call's output: flushes, then recorded
    api.fft_convolve_unordered(s, a, b, acc, 0.5)       # not recordable: flushes, then runs
    api.defer_flush()
    A = oracle.transform(so, oracle.transform(so, a0, FORWARD), BACKWARD)
    B = oracle.transform(so, b0, FORWARD)
    assert_frames_close(a, A, what="forward then backward in place")
    assert_frames_close(b, B, what="independent call")
    # this backend's "unordered" layout is the canonical interleaved one: the product is an element-wise complex multiply
    prod = (A[0::2].astype(np.float64) + 1j * A[1::2]) * (B[0::2].astype(np.float64) + 1j * B[1::2]) * 0.5
    want = np.empty(2 * n)
    want[0::2], want[1::2] = prod.real, prod.imag
    assert_frames_close(acc, want, what="convolve after the flush")
    # outside begin / flush the calls are synchronous again
    a[:] = a0
    api.fft_transform(s, a, a, None, api.FFT_FORWARD)
    assert_frames_close(a, oracle.transform(so, a0, FORWARD), what="synchronous again")
    for p in (ap, bp, accp):
        api.aligned_free(p)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_plan_destroyed_before_the_flush(gpu, oracle):
    """The reference caller's new_setup -> fft_transform -> fft_destroy_setup (test/test.cpp:30-67) inside a deferred region:
    recorded calls hold their (immutable) plan by value, so they still run correctly after the handle has been destroyed --
    or, for a pre-allocated plan, after the caller has overwritten and freed the memory its header lived in."""
    api = gpu[0]
    n = 2048
    so = oracle.new_setup(n, COMPLEX)
    x, xp = api.aligned_array(2 * n)
    y, yp = api.aligned_array(2 * n)
    x2, x2p = api.aligned_array(2 * n)
    y2, y2p = api.aligned_array(2 * n)
    x[:], x2[:] = ol.uniform(2 * n, 5), ol.uniform(2 * n, 6)
    y[:], y2[:] = -1.0, -1.0
    api.defer_begin()
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    api.fft_transform(s, x, y, None, api.FFT_FORWARD)
    api.fft_destroy_setup(s)  # before the flush
    nbytes = api.fft_bytes_required(n, api.FFT_COMPLEX)
    block, blockp = api.aligned_array((nbytes + 3) // 4)
    sp = api.fft_new_setup_preallocated(n, api.FFT_COMPLEX, blockp)
    assert sp is not None
    api.fft_transform(sp, x2, y2, None, api.FFT_BACKWARD)
    block[:] = 12345.0  # the caller owns that memory (chowdsp_fft.h:98-118) and is done with the plan
    api.aligned_free(blockp)
    assert np.all(y == -1.0) and np.all(y2 == -1.0), "recorded calls have not run yet"
    api.defer_flush()
    assert_frames_close(y, oracle.transform(so, x, FORWARD), what="call recorded on a plan destroyed before the flush")
    assert_frames_close(y2, oracle.transform(so, x2, BACKWARD), what="call recorded on a pre-allocated plan freed before the flush")
    for p in (xp, yp, x2p, y2p):
        api.aligned_free(p)
    oracle.destroy_setup(so)
