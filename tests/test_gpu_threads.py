"""The setup is read-only and may be shared by concurrent threads, each with its own buffers
(reference chowdsp_fft.h:87-91).  Host threads hammer ONE plan through the legacy host-pointer symbols
(default stream, per-thread staging) and through the batched extension on per-thread streams; every result must
equal the oracle's.  Sizes cover the LDS-resident, the two-pass (shared workspace, enqueue lock) and the generic path.
"""
import threading

import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,tr", [(1024, COMPLEX), (4096, REAL), (16384, COMPLEX), (65536, COMPLEX), (480, COMPLEX)])
def test_legacy_calls_from_four_threads_share_one_setup(gpu, oracle, n, tr):
    api = gpu[0]
    nf = 2 * n if tr == COMPLEX else n
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    nthreads, reps = 4, 12
    inputs = [ol.uniform(nf, 100 + t) for t in range(nthreads)]
    wants = [oracle.transform(so, inputs[t], FORWARD) for t in range(nthreads)]
    errors = []

    def worker(t):
        try:
            x, xp = api.aligned_array(nf)
            y, yp = api.aligned_array(nf)
            x[:] = inputs[t]
            for _ in range(reps):
                y[:] = 0
                api.fft_transform(s, x, y, None, api.FFT_FORWARD)
                assert_frames_close(y, wants[t], what=f"thread {t}")
            api.aligned_free(xp)
            api.aligned_free(yp)
        except Exception as e:  # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_batched_calls_from_threads_on_their_own_streams(gpu, oracle):
    api, torch, dev = gpu
    n = 32768  # two-pass: per-stream workspaces
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    nthreads = 3
    xs = [ol.uniform(6 * 2 * n, 200 + t).reshape(6, 2 * n) for t in range(nthreads)]
    wants = [oracle.transform_batch(so, xs[t], FORWARD) for t in range(nthreads)]
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            xd = torch.from_numpy(xs[t]).to(dev)
            yd = torch.empty_like(xd)
            torch.cuda.synchronize()
            for _ in range(8):
                api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD, stream=st)
            st.synchronize()
            assert_frames_close(yd.cpu().numpy(), wants[t], what=f"stream thread {t}")
        except Exception as e:
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_legacy_call_from_a_fresh_thread_on_another_device(oracle):
    """A setup created on device 1, called through the legacy host-pointer symbols from a thread that has never selected
    a device (its current device is 0): the staging memory, the copies, the launch and the final synchronisation all have to
    address the PLAN's device.  Needs two devices (skipped on a one-GPU lease)."""
    from chowdsp_fft_amd import _lib, api

    lib = _lib.lib()
    if lib.fft_gpu_device_count() < 2:
        pytest.skip("needs two HIP devices")
    n, tr = 4096, COMPLEX
    assert lib.fft_gpu_set_device(1) == 0
    s = api.fft_new_setup(n, tr)
    assert s is not None and lib.fft_gpu_setup_device(s) == 1
    assert lib.fft_gpu_set_device(0) == 0
    so = oracle.new_setup(n, tr)
    x = ol.uniform(2 * n, 5)
    want = oracle.transform(so, x, FORWARD)
    errors = []

    def worker():
        try:
            xb, xp = api.aligned_array(2 * n)
            yb, yp = api.aligned_array(2 * n)
            xb[:] = x
            for _ in range(20):
                yb[:] = 0
                api.fft_transform(s, xb, yb, None, api.FFT_FORWARD)
                assert_frames_close(yb, want, what="device-1 plan from a fresh thread")
            api.aligned_free(xp)
            api.aligned_free(yp)
        except Exception as e:
            errors.append(e)

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert not errors, errors[0]
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
