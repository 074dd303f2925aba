"""The batched entry points are plain asynchronous launches on the caller's stream, so a sequence of them can be
captured into a hipGraph and replayed (what a block-by-block audio host would do to shed per-launch overhead).
Covers the LDS-resident kernels, the fused convolution, the partitioned engine, and the two-pass plans (chunks one after
the other on the capturing stream; with CHOWDSP_FFT_LANES=2 their two-lane pipeline forks from and joins back into it with events).
Everything is called once before the capture (plans, occupancy queries and workspaces are created on first use).
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,tr,batch", [(256, COMPLEX, 16), (4096, COMPLEX, 8), (1024, REAL, 32), (16384, COMPLEX, 4),
                                        (65536, COMPLEX, 600), (131072, REAL, 5)])
def test_forward_backward_in_a_graph(gpu, oracle, n, tr, batch):
    api, torch, dev = gpu
    nf = 2 * n if tr == COMPLEX else n
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    x = torch.empty((batch, nf), device=dev)
    y, z = torch.empty_like(x), torch.empty_like(x)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        # warm-up on the stream that will capture: first-use allocations must not happen inside the capture
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    idx = np.unique(np.linspace(0, batch - 1, min(batch, 5)).astype(int))
    for rep in range(3):  # replay with fresh input each time
        xh = ol.uniform(batch * nf, 40 + rep + n).reshape(batch, nf)
        x.copy_(torch.from_numpy(xh))
        g.replay()
        torch.cuda.synchronize()
        want = oracle.transform_batch(so, xh[idx], FORWARD)
        assert_frames_close(y.cpu().numpy()[idx], want, what=f"graph replay {rep} fwd")
        assert np.abs(z.cpu().numpy()[idx] / n - xh[idx]).max() <= 3e-6
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_engine_block_in_a_graph(gpu):
    api, torch, dev = gpu
    block, channels, ir_len = 256, 6, 700
    rng = np.random.RandomState(5)
    ir = rng.uniform(-1, 1, ir_len).astype(np.float32)
    xs = rng.uniform(-1, 1, (channels, 6 * block)).astype(np.float32)
    side = torch.cuda.Stream()
    xin, out = torch.zeros((channels, block), device=dev), torch.empty((channels, block), device=dev)
    with torch.cuda.stream(side):
        eng = api.ConvEngine(block, channels, ir, shared=True)
        eng.process(xin, out)  # warm-up (silence in, engine state stays zero)
        eng.reset()
    side.synchronize()
    # The engine keeps its delay-line position in device memory (blocks >= 128), so ONE captured process() call is
    # replayed for every following block -- per block the host only refreshes the input buffer and launches the graph.
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.process(xin, out)
    y = np.empty_like(xs)
    for t in range(6):
        xin.copy_(torch.from_numpy(np.ascontiguousarray(xs[:, t * block:(t + 1) * block])))
        g.replay()
        torch.cuda.synchronize()
        y[:, t * block:(t + 1) * block] = out.cpu().numpy()
    for c in range(channels):
        d = np.convolve(xs[c].astype(np.float64), ir.astype(np.float64))[: xs.shape[1]]
        assert np.abs(y[c] - d).max() <= 5e-6 * np.abs(d).max()
    eng.close()


def test_fused_convolution_in_a_graph(gpu, oracle):
    api, torch, dev = gpu
    n, batch = 2048, 12
    s, so = api.fft_new_setup(n, api.FFT_REAL), oracle.new_setup(n, REAL)
    h = ol.uniform(n, 77).reshape(1, n)
    hd = torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    x, y = torch.empty((batch, n), device=dev), torch.empty((batch, n), device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        api.fft_convolve_frames_batch(s, x, H, y, 1.0 / n)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        api.fft_convolve_frames_batch(s, x, H, y, 1.0 / n)
    xh = ol.uniform(batch * n, 78).reshape(batch, n)
    x.copy_(torch.from_numpy(xh))
    g.replay()
    torch.cuda.synchronize()
    hb = oracle.transform(so, h[0], FORWARD, False)
    want = np.stack([oracle.transform(so, oracle.convolve(so, oracle.transform(so, xh[i], FORWARD, False), hb, np.zeros(n, np.float32), 1.0 / n), BACKWARD, False) for i in range(batch)])
    assert_frames_close(y.cpu().numpy(), want, what="fused convolution graph replay")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
