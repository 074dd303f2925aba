"""the streaming partitioned overlap-save convolution engine, against
  (a) the same algorithm assembled from the oracle's operations (unordered forward, P x convolve-accumulate,
      unordered backward) -- parity bound 1e-6 relative L2 per output block (relative to the running signal), and
  (b) direct time-domain convolution in float64 (what the engine must equal mathematically).
"""
import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import BACKWARD, FORWARD, REAL

pytestmark = pytest.mark.gpu


def _oracle_engine(oracle, block, x, ir):
    """x: [channels, T*block], ir: [ir_len] shared; returns y of the same shape"""
    n = 2 * block
    s = oracle.new_setup(n, REAL)
    P = (ir.size + block - 1) // block
    H = []
    for p in range(P):
        h = np.zeros(n, np.float32)
        taps = ir[p * block:(p + 1) * block]
        h[:taps.size] = taps
        H.append(oracle.transform(s, h, FORWARD, False))
    C, total = x.shape
    y = np.zeros_like(x)
    for c in range(C):
        prev = np.zeros(block, np.float32)
        line = [np.zeros(n, np.float32) for _ in range(P)]  # newest first
        for t in range(total // block):
            new = x[c, t * block:(t + 1) * block]
            X = oracle.transform(s, np.concatenate([prev, new]), FORWARD, False)
            line = [X] + line[:-1]
            acc = np.zeros(n, np.float32)
            for p in range(P):
                acc = oracle.convolve(s, line[p], H[p], acc, 1.0 / n)
            y[c, t * block:(t + 1) * block] = oracle.transform(s, acc, BACKWARD, False)[block:]
            prev = new.copy()
    oracle.destroy_setup(s)
    return y


# fused: blocks <= 4096 run as two fused kernels per block; CHOWDSP_FFT_ENGINE_FUSED=0 (read at engine creation) forces
# the five-kernel path built from the public batched operations, which longer blocks always take
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("block,ir_len,channels,nblocks", [(16, 40, 3, 7), (64, 64, 2, 5), (128, 300, 37, 5), (256, 1000, 4, 6),
                                                           (1024, 1500, 5, 4), (2048, 5000, 3, 4), (4096, 4096, 2, 3),
                                                           (8192, 20000, 1, 3)])
def test_engine_vs_oracle_and_direct(gpu, oracle, monkeypatch, fused, block, ir_len, channels, nblocks):
    api, torch, dev = gpu
    monkeypatch.setenv("CHOWDSP_FFT_ENGINE_FUSED", "1" if fused else "0")
    rng = np.random.RandomState(block + ir_len)
    ir = (rng.uniform(-1, 1, ir_len) * np.exp(-np.arange(ir_len) / (ir_len / 4.0))).astype(np.float32)
    x = rng.uniform(-1, 1, (channels, nblocks * block)).astype(np.float32)
    eng = api.ConvEngine(block, channels, ir, shared=True)
    assert eng.partitions == (ir_len + block - 1) // block
    y = np.empty_like(x)
    out = torch.empty((channels, block), device=dev)
    for t in range(nblocks):
        xin = torch.from_numpy(np.ascontiguousarray(x[:, t * block:(t + 1) * block])).to(dev)
        eng.process(xin, out)
        torch.cuda.synchronize()
        y[:, t * block:(t + 1) * block] = out.cpu().numpy()
    want = _oracle_engine(oracle, block, x, ir)
    direct = np.stack([np.convolve(x[c].astype(np.float64), ir.astype(np.float64))[: x.shape[1]] for c in range(channels)])
    scale = np.abs(direct).max()
    for c in range(channels):
        assert ol.rel_l2(y[c], want[c]) <= 1e-6, "engine vs oracle-built engine"
        assert np.abs(y[c] - direct[c]).max() <= 5e-6 * scale, "engine vs float64 direct convolution"
    # reset returns the engine to silence
    eng.reset()
    eng.process(torch.zeros((channels, block), device=dev), out)
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0
    eng.close()


def test_engine_per_channel_filters(gpu):
    api, torch, dev = gpu
    block, channels, ir_len, nblocks = 128, 5, 300, 4
    rng = np.random.RandomState(3)
    ir = rng.uniform(-1, 1, (channels, ir_len)).astype(np.float32)
    x = rng.uniform(-1, 1, (channels, nblocks * block)).astype(np.float32)
    eng = api.ConvEngine(block, channels, torch.from_numpy(ir).to(dev), shared=False)  # device-resident impulse responses
    out = torch.empty((channels, block), device=dev)
    y = np.empty_like(x)
    for t in range(nblocks):
        eng.process(torch.from_numpy(np.ascontiguousarray(x[:, t * block:(t + 1) * block])).to(dev), out)
        torch.cuda.synchronize()
        y[:, t * block:(t + 1) * block] = out.cpu().numpy()
    for c in range(channels):
        d = np.convolve(x[c].astype(np.float64), ir[c].astype(np.float64))[: x.shape[1]]
        assert np.abs(y[c] - d).max() <= 5e-6 * np.abs(d).max()
    eng.close()


@pytest.mark.parametrize("block,channels", [(128, 5), (2048, 3), (512, 7)])
def test_engine_per_channel_single_partition_ragged_group(gpu, block, channels):
    """per-channel filters, ONE partition, channels not a multiple of the teams per workgroup: the idle teams of the last
    group must not read filter bins past the end of H (there is no next partition behind it)"""
    api, torch, dev = gpu
    ir_len, nblocks = block, 3
    rng = np.random.RandomState(block + channels)
    ir = rng.uniform(-1, 1, (channels, ir_len)).astype(np.float32)
    x = rng.uniform(-1, 1, (channels, nblocks * block)).astype(np.float32)
    eng = api.ConvEngine(block, channels, torch.from_numpy(ir).to(dev), shared=False)
    assert eng.partitions == 1
    out = torch.empty((channels, block), device=dev)
    y = np.empty_like(x)
    for t in range(nblocks):
        eng.process(torch.from_numpy(np.ascontiguousarray(x[:, t * block:(t + 1) * block])).to(dev), out)
        torch.cuda.synchronize()
        y[:, t * block:(t + 1) * block] = out.cpu().numpy()
    for c in range(channels):
        d = np.convolve(x[c].astype(np.float64), ir[c].astype(np.float64))[: x.shape[1]]
        assert np.abs(y[c] - d).max() <= 5e-6 * np.abs(d).max()
    eng.close()


def test_engine_rejects_bad_arguments(gpu):
    api, torch, dev = gpu
    for block in (0, 8, 24, 100):
        with pytest.raises(api.FFTError):
            api.ConvEngine(block, 1, np.ones(8, np.float32))
