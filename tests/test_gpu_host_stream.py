"""fft_transform_host_batch -- frames in HOST memory, pipelined over two streams; large buffers
pinned in place (only the pages that lie entirely inside them; partial head / tail pages through bounce pages), small
ones through the library's pinned staging ring.  Must equal `batch` single-frame legacy calls, i.e. the oracle, frame
for frame (1e-6), for batches that span several pipeline chunks, ragged batch counts, in-place buffers, unaligned views
into larger arrays, and every execution path (LDS-resident, two-pass, generic)."""
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


@pytest.mark.parametrize("inplace", [False, True])
def test_unaligned_view_of_a_larger_array(gpu, oracle, inplace):
    """A 64 MiB+ slice starting 20 bytes into a larger array (the caller pattern `ptr + offset`): its first and last page
    are shared with the rest of the parent, so only the pages inside the slice may be pinned; the bytes around the slice
    must come back untouched."""
    api = gpu[0]
    n, batch = 4096, 2100  # 2100 x 32 KiB = 65.6 MiB
    nf = 2 * n
    guard = 4099  # floats before and after, not a multiple of a page
    parent_in = np.full(batch * nf + 2 * guard, 7.0, np.float32)
    parent_out = parent_in if inplace else np.full(batch * nf + 2 * guard, -9.0, np.float32)
    x = parent_in[5:5 + batch * nf].reshape(batch, nf)
    assert x.ctypes.data % 4096 != 0 and (x.ctypes.data - parent_in.ctypes.data) == 20
    x[:] = ol.uniform(batch * nf, 321).reshape(batch, nf)
    x0 = x.copy()
    y = parent_out[5:5 + batch * nf].reshape(batch, nf)
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)
    idx = np.array([0, 1, 2, 1023, 1024, 1025, 2047, 2048, batch - 2, batch - 1])  # chunk edges, first and last page
    assert_frames_close(y[idx], oracle.transform_batch(so, x0[idx], FORWARD), what="unaligned view")
    fill_in, fill_out = 7.0, (7.0 if inplace else -9.0)
    assert np.all(parent_in[:5] == fill_in) and np.all(parent_in[5 + batch * nf:] == fill_in)
    assert np.all(parent_out[:5] == fill_out) and np.all(parent_out[5 + batch * nf:] == fill_out), "bytes outside the view were written"
    if not inplace:
        assert np.array_equal(x, x0), "the input view was modified"
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_small_heap_blocks_next_to_pageable_copies(gpu, oracle):
    """The sequence that faulted in round 2 (session r02j): small heap blocks -- which share their first and last page with
    other heap objects -- streamed through the library while the runtime makes pageable copies from and to NEIGHBOURING
    objects of the same pages, and the addresses are recycled.  Small buffers go through the staging ring now (nothing of
    the caller's is pinned), so neighbouring pageable copies must stay valid."""
    api, torch, dev = gpu
    for rounds in range(3):
        for n, batch in ((64, 7), (480, 333), (1024, 200), (4096, 23), (4096, 250)):
            nf = 2 * n
            s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
            left = np.random.rand(1000 + 37 * rounds).astype(np.float32)   # heap neighbours
            x = ol.uniform(batch * nf, 11 * rounds + n).reshape(batch, nf)
            right = np.random.rand(777).astype(np.float32)
            y = np.empty_like(x)
            dl = torch.from_numpy(left).to(dev)       # pageable H2D from a neighbour before ...
            api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)
            dr = torch.from_numpy(right).to(dev)      # ... and after
            dx = torch.from_numpy(x).to(dev)          # and from the very addresses the call just used
            torch.cuda.synchronize()
            assert np.array_equal(dl.cpu().numpy(), left) and np.array_equal(dr.cpu().numpy(), right)
            assert np.array_equal(dx.cpu().numpy(), x)
            pick = np.array([0, batch // 2, batch - 1])
            assert_frames_close(y[pick], oracle.transform_batch(so, x[pick], FORWARD), what="small heap block")
            api.fft_destroy_setup(s)
            oracle.destroy_setup(so)
            del x, y, left, right, dl, dr, dx  # freed: the next round's arrays reuse the addresses


@pytest.mark.parametrize("inplace", [False, True])
def test_tail_page_shared_by_the_last_two_chunks(gpu, oracle, inplace):
    """Frames shorter than a page (N = 64 complex: 512 B), batch = one pipeline chunk + 1 frame, buffer starting 16 bytes
    into a page: the partial tail page (528 B) belongs to BOTH last chunks -- 16 bytes to the full chunk, 512 to the
    one-frame chunk -- which run on different streams.  Each tail byte has its own place in the bounce page; round 3
    put both pieces at offset 0 (the earlier chunk's input overwritten before its DMA ran, its output never delivered)."""
    api = gpu[0]
    n = 64
    nf = 2 * n
    chunk_frames = (32 << 20) // (nf * 4)
    for batch in (chunk_frames + 1, 2 * chunk_frames + 3):
        raw_in = np.zeros(batch * nf + 2048, np.float32)
        off = (-(raw_in.ctypes.data) % 4096) // 4 + 4  # first float at page + 16 bytes
        x = raw_in[off:off + batch * nf].reshape(batch, nf)
        assert x.ctypes.data % 4096 == 16
        x[:] = ol.uniform(batch * nf, 77).reshape(batch, nf)
        x0 = x.copy()
        if inplace:
            y = x
        else:
            raw_out = np.full(batch * nf + 2048, -3.0, np.float32)
            offo = (-(raw_out.ctypes.data) % 4096) // 4 + 4
            y = raw_out[offo:offo + batch * nf].reshape(batch, nf)
        s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
        api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)
        idx = np.unique(np.concatenate([np.arange(0, 4), np.arange(chunk_frames - 3, chunk_frames + 2), np.arange(batch - 4, batch)]))
        idx = idx[idx < batch]
        assert_frames_close(y[idx], oracle.transform_batch(so, x0[idx], FORWARD), what="tail page split over two chunks")
        # every frame: forward then backward in place returns N x
        z = y.copy() if not inplace else y
        api.fft_transform_host_batch(s, z, z, api.FFT_BACKWARD)
        assert np.abs(z / n - x0).max() <= 2e-6
        api.fft_destroy_setup(s)
        oracle.destroy_setup(so)
