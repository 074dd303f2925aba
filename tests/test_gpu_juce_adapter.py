"""Second half of SURVEY section 8 row f3: the JUCE FFT-engine layouts, batched on the device.

Expected values restate the reference adapter (chowdsp_fft_juce/chowdsp_fft_juce.cpp:32-86) line by line on top of
the oracle's fft_transform -- the adapter is glue around exactly that call -- and are cross-checked against numpy's
float64 FFT so that the restated glue itself is pinned.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu

ORDERS = [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17]  # N = 32 .. 131072: staged, register-direct, pipelined and split-plane plans (fused layouts, N <= 65536) and the two-launch form above


def ref_real_forward(oracle, so_real, buf, n, ignore_negative_freqs):
    """performRealOnlyForwardTransform, chowdsp_fft_juce.cpp:47-67, applied to each row of buf [batch, 2N]"""
    out = buf.copy()
    out[:, :n] = oracle.transform_batch(so_real, np.ascontiguousarray(buf[:, :n]), FORWARD)  # :49-55
    nyq = n // 2
    out[:, 2 * nyq] = out[:, 1]  # :58
    out[:, 2 * nyq + 1] = 0.0  # :59
    out[:, 1] = 0.0  # :60
    if not ignore_negative_freqs:  # :61-66
        c = out.view(np.complex64)
        for i in range(1, nyq):
            c[:, nyq + i] = np.conj(c[:, nyq - i])
    return out


def ref_real_inverse(oracle, so_real, buf, n):
    """performRealOnlyInverseTransform, chowdsp_fft_juce.cpp:69-86"""
    out = buf.copy()
    out[:, 1] = out[:, n]  # :72
    y = oracle.transform_batch(so_real, np.ascontiguousarray(out[:, :n]), BACKWARD)  # :75-83
    out[:, :n] = y * np.float32(1.0 / n)  # :85
    return out


@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("ignore_neg", [False, True])
def test_real_forward_layout(gpu, oracle, order, ignore_neg):
    api, torch, dev = gpu
    n, batch = 1 << order, 23
    eng = api.JuceFFT(order)
    so = oracle.new_setup(n, REAL)
    buf = np.full((batch, 2 * n), 9.5, np.float32)  # JUCE hands over 2N floats; the upper half is garbage on entry
    buf[:, :n] = ol.uniform(batch * n, 100 + order).reshape(batch, n)
    want = ref_real_forward(oracle, so, buf, n, ignore_neg)
    d = torch.from_numpy(buf).to(dev)
    eng.perform_real_only_forward_transform(d, ignore_neg)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    used = n + 2 if ignore_neg else 2 * n
    assert_frames_close(got[:, :used], want[:, :used], what=f"JUCE real forward order {order}")
    if ignore_neg:
        assert np.all(got[:, n + 2:] == 9.5), "buffer beyond bin N/2 must be left alone when negative freqs are ignored"
    assert np.all(got[:, 1] == 0.0) and np.all(got[:, n + 1] == 0.0)
    # pin the restated glue: interleaved bins == float64 rfft
    spec = np.fft.rfft(buf[:, :n].astype(np.float64), axis=1)
    c = got.view(np.complex64)
    assert np.abs(c[:, :n // 2 + 1] - spec).max() <= 2e-6 * np.abs(spec).max()
    if not ignore_neg:
        full = np.fft.fft(buf[:, :n].astype(np.float64), axis=1)
        assert np.abs(c[:, :n] - full).max() <= 2e-6 * np.abs(full).max()
    eng.close()
    oracle.destroy_setup(so)


@pytest.mark.parametrize("order", ORDERS)
def test_real_inverse_layout_and_round_trip(gpu, oracle, order):
    api, torch, dev = gpu
    n, batch = 1 << order, 17
    eng = api.JuceFFT(order)
    so = oracle.new_setup(n, REAL)
    x = ol.uniform(batch * n, 200 + order).reshape(batch, n)
    buf = np.zeros((batch, 2 * n), np.float32)
    buf[:, :n] = x
    spec = ref_real_forward(oracle, so, buf, n, True)
    want = ref_real_inverse(oracle, so, spec, n)
    d = torch.from_numpy(spec).to(dev)
    eng.perform_real_only_inverse_transform(d)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert_frames_close(got[:, :n], want[:, :n], what=f"JUCE real inverse order {order}")
    assert np.abs(got[:, :n] - x).max() <= 3e-6  # 1/N scaling included: forward then inverse is the identity
    # and device-side round trip
    d2 = torch.from_numpy(buf).to(dev)
    eng.perform_real_only_forward_transform(d2, False)
    eng.perform_real_only_inverse_transform(d2)
    torch.cuda.synchronize()
    assert np.abs(d2.cpu().numpy()[:, :n] - x).max() <= 3e-6
    eng.close()
    oracle.destroy_setup(so)


@pytest.mark.parametrize("order", [5, 8, 12, 14, 16])
def test_perform_complex(gpu, oracle, order):
    """perform, chowdsp_fft_juce.cpp:32-45: unscaled forward, inverse times 1/N"""
    api, torch, dev = gpu
    n, batch = 1 << order, 9
    eng = api.JuceFFT(order)
    so = oracle.new_setup(n, COMPLEX)
    x = ol.uniform(batch * 2 * n, 300 + order).reshape(batch, 2 * n)
    d, o = torch.from_numpy(x).to(dev), torch.empty((batch, 2 * n), device=dev)
    eng.perform(d, o, False)
    torch.cuda.synchronize()
    fwd = oracle.transform_batch(so, x, FORWARD)
    assert_frames_close(o.cpu().numpy(), fwd, what="JUCE perform fwd")
    eng.perform(o, o, True)  # in place like the adapter's inout_buffer
    torch.cuda.synchronize()
    want = oracle.transform_batch(so, fwd, BACKWARD) * np.float32(1.0 / n)  # :43-44
    assert_frames_close(o.cpu().numpy(), want, what="JUCE perform inverse")
    assert np.abs(o.cpu().numpy() - x).max() <= 3e-6
    eng.close()
    oracle.destroy_setup(so)


def test_order_below_5_rejected(gpu):
    api = gpu[0]
    with pytest.raises(ValueError):  # chowdsp_fft_juce.cpp:20-29
        api.JuceFFT(4)
