"""fft_transform_host_batch -- frames in HOST memory, pinned in place, pipelined over two
streams.  Must equal `batch` single-frame legacy calls, i.e. the oracle, frame for frame (1e-6), for batches that
span several pipeline chunks, ragged batch counts, in-place buffers and every execution path (LDS-resident,
two-pass, generic)."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,tr,batch", [(4096, COMPLEX, 2500), (1024, REAL, 20001), (64, COMPLEX, 7), (65536, COMPLEX, 150), (480, REAL, 333)])
def test_host_batch_equals_oracle(gpu, oracle, n, tr, batch):
    api = gpu[0]
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(batch * nf, 5 + n).reshape(batch, nf)  # 2500 x 32 KiB = 78 MiB: three 32 MiB chunks
    y = np.empty_like(x)
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)
    idx = np.unique(np.concatenate([np.arange(0, batch, max(1, batch // 37)), [batch - 1]]))
    want = oracle.transform_batch(so, x[idx], FORWARD)
    assert_frames_close(y[idx], want, what="host batch fwd")
    # in place, backward
    z = y.copy()
    api.fft_transform_host_batch(s, z, z, api.FFT_BACKWARD)
    assert_frames_close(z[idx], oracle.transform_batch(so, y[idx], BACKWARD), what="host batch bwd in place")
    assert np.abs(z[idx] / n - x[idx]).max() <= 3e-6
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_host_batch_with_library_pinned_memory(gpu, oracle):
    """buffers from fft_gpu_host_malloc are already pinned: registering them again must not break the call"""
    import ctypes as C

    api = gpu[0]
    from chowdsp_fft_amd import _lib

    n, batch = 1024, 64
    nbytes = batch * 2 * n * 4
    p_in, p_out = _lib.lib().fft_gpu_host_malloc(nbytes), _lib.lib().fft_gpu_host_malloc(nbytes)
    assert p_in and p_out
    x = np.ctypeslib.as_array((C.c_float * (batch * 2 * n)).from_address(p_in)).reshape(batch, 2 * n)
    y = np.ctypeslib.as_array((C.c_float * (batch * 2 * n)).from_address(p_out)).reshape(batch, 2 * n)
    x[:] = ol.uniform(batch * 2 * n, 9).reshape(batch, 2 * n)
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)
    assert_frames_close(y, oracle.transform_batch(so, x, FORWARD), what="pinned host batch")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
    _lib.lib().fft_gpu_host_free(p_in)
    _lib.lib().fft_gpu_host_free(p_out)
