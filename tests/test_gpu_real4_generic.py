"""Long mixed-radix REAL frames by the real four-step with generic cores (csrc/real4_generic_kernels.hpp, round 5): N = R1 x R2,
two sweeps over the data instead of the complex two-pass transform + stand-alone pair stage (three).

What the reference does for these lengths: FFTPACK real passes of radix 2-5 + pffft_real_finalize
(/root/reference/simd/chowdsp_fft_impl_sse.cpp:552-786 radf2..radf5, :1081-1356 radb2..radb5, :849-1078; sizes of this kind are in
its "Other sizes" test, /root/reference/test/test.cpp:279-285).  Parity: forward and backward against the oracle (forward skipped
where N % 25 == 0: the reference's radf5_ps is wrong there, tests/test_gpu_other_sizes.py) AND against the float64 transform;
ragged batches, in place, strided frames; the packed (X0, X_{N/2}) slot and the self-mirrored rows bin by bin.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import MIXED_RADIX_MAXABS_TOL, assert_frames_close
from oracle_lib import BACKWARD, FORWARD, REAL

# real lengths with an admissible split N = R1 x R2 (R1 = 32 j <= 639, R2 even <= 639; tables.cpp split_real4_generic picks the one nearest to square)
SIZES = [41472, 46080, 61440, 76800, 92160, 96000, 122880, 153600, 245760, 73728, 110592]


def _exact(x, n, direction):
    x = np.asarray(x, dtype=np.float64)
    if direction == FORWARD:
        t = np.fft.rfft(x, axis=1)
        out = np.empty_like(x)
        out[:, 0], out[:, 1] = t[:, 0].real, t[:, n // 2].real
        out[:, 2::2], out[:, 3::2] = t[:, 1:n // 2].real, t[:, 1:n // 2].imag
        return out
    spec = np.zeros((x.shape[0], n // 2 + 1), dtype=np.complex128)
    spec[:, 0], spec[:, n // 2] = x[:, 0], x[:, 1]
    spec[:, 1:n // 2] = x[:, 2::2] + 1j * x[:, 3::2]
    return np.fft.irfft(spec, n=n, axis=1) * n


# lengths with a compile-time stage list (csrc/tables.hpp CFFT_STATIC_TILE_SIZES_A/B/C, at most 639 points)
STATIC = {48, 80, 96, 144, 160, 192, 240, 288, 320, 384, 400, 432, 480, 576, 150, 162, 180, 200, 216, 225, 243, 250, 256, 270, 300, 324, 360, 375,
          405, 450, 486, 500, 512, 540, 600, 625}


@pytest.fixture(params=[1, 2], ids=["static-lists", "every-split"])
def real4g(request):
    """option "real4_generic" (read when a plan is made): 1 = the default, splits whose two factors both have a compile-time stage
    list; 2 = every admissible split with run-time stage lists only (slower than the three sweeps: keeps those kernels tested)"""
    from chowdsp_fft_amd import api

    assert api.get_option("real4_generic") == 1, "the default"
    api.set_option("real4_generic", request.param)
    yield api
    api.set_option("real4_generic", 1)


def _splits(n, mode):
    return [(max(a, n // a) / min(a, n // a), a) for a in range(32, 640, 32)
            if n % a == 0 and (n // a) <= 639 and (n // a) % 2 == 0 and (n // a) >= 16 and _smooth(a)
            and (mode == 2 or (a in STATIC and (n // a) in STATIC))]


def test_split_rule_host_logic(real4g):
    """pure host logic (no GPU): which lengths take the form and with which factors"""
    api = real4g
    mode = api.get_option("real4_generic")
    for n in SIZES:
        d = api.plan_describe(n, api.FFT_REAL)
        cands = _splits(n, mode)
        if not cands:
            assert d["n1"] * d["n2"] == n // 2, (n, d)  # the complex two-pass plan of N / 2 points
            continue
        r1, r2 = d["n1"], d["n2"]
        assert d["passes"] == 2 and r1 * r2 == n and r1 % 32 == 0 and r2 % 2 == 0 and r1 <= 639 and r2 <= 639, (n, d)
        assert r1 == min(cands)[1], (n, d, min(cands))  # nearest to square among the admissible splits
    api.set_option("real4_generic", 0)
    d = api.plan_describe(41472, api.FFT_REAL)
    assert d["n1"] * d["n2"] == 41472 // 2, "option off: the complex two-pass plan of N / 2 points"
    api.set_option("real4_generic", mode)
    # not taken: short frames stay on chip (one pass); powers of two have their own real four-step (power-of-two factors)
    assert api.plan_describe(36864, api.FFT_REAL)["passes"] == 1
    d = api.plan_describe(1 << 17, api.FFT_REAL)
    assert d["passes"] == 2 and d["n1"] & (d["n1"] - 1) == 0 and d["n2"] & (d["n2"] - 1) == 0


def _smooth(v):
    for q in (2, 3, 5):
        while v % q == 0:
            v //= q
    return v == 1


def _gpu(gpu, s, x, direction):
    api, torch, dev = gpu
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    yd = torch.full_like(xd, -3.0)
    api.fft_transform_batch(s, xd, yd, direction)
    torch.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES)
def test_forward_and_backward_vs_oracle_and_float64(gpu, oracle, real4g, n):
    api = gpu[0]
    s, so = api.fft_new_setup(n, api.FFT_REAL), oracle.new_setup(n, REAL)
    assert s is not None
    batch = 5
    x = ol.uniform(batch * n, 300 + n % 1000).reshape(batch, n)
    got_f = _gpu(gpu, s, x, api.FFT_FORWARD)
    exact_f = _exact(x, n, FORWARD)
    assert_frames_close(got_f, exact_f, what=f"real N={n} fwd vs float64")
    if n % 25 != 0:
        assert_frames_close(got_f, oracle.transform_batch(so, x, FORWARD), what=f"real N={n} fwd vs oracle", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
    spec = exact_f.astype(np.float32)
    got_b = _gpu(gpu, s, spec, api.FFT_BACKWARD)
    assert_frames_close(got_b, oracle.transform_batch(so, spec, BACKWARD), what=f"real N={n} bwd vs oracle", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
    assert_frames_close(got_b, _exact(spec, n, BACKWARD), what=f"real N={n} bwd vs float64")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 37, 300])
def test_ragged_batches_in_place_and_round_trip(gpu, real4g, batch):
    """batches that fill the persistent grid partly / many times over (300 frames of 41472 samples run in several chunks on two
    lanes); in place; BACKWARD (FORWARD (x)) = N x"""
    api, torch, dev = gpu
    n = 41472
    s = api.fft_new_setup(n, api.FFT_REAL)
    g = torch.Generator(device=dev)
    g.manual_seed(batch)
    x = torch.rand((batch, n), device=dev, generator=g) * 2 - 1
    y = torch.empty_like(x)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    inplace = x.clone()
    api.fft_transform_batch(s, inplace, inplace, api.FFT_FORWARD)
    torch.cuda.synchronize()
    assert torch.equal(inplace, y), "in place == out of place"
    pick = sorted({0, batch // 2, batch - 1})
    assert_frames_close(y[pick].cpu().numpy(), _exact(x[pick].cpu().numpy(), n, FORWARD), what="forward vs float64")
    z = torch.empty_like(x)
    api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    err = float(((z / n - x).abs().amax(dim=1) / x.abs().amax(dim=1)).max())
    assert err <= 2e-6, f"round trip {err:.2e}"
    api.fft_destroy_setup(s)


@pytest.mark.gpu
def test_strided_frames(gpu, real4g):
    """frames embedded in larger rows (fft_transform_batch_strided): strides in floats, not multiples of the frame"""
    api, torch, dev = gpu
    n, batch, stride_in, stride_out = 96000, 7, 96000 + 64, 96000 + 160
    s = api.fft_new_setup(n, api.FFT_REAL)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    xin = torch.rand((batch, stride_in), device=dev, generator=g) * 2 - 1
    yout = torch.full((batch, stride_out), -4.0, device=dev)
    api.fft_transform_batch_strided(s, xin, stride_in, yout, stride_out, batch, api.FFT_FORWARD)
    torch.cuda.synchronize()
    got = yout[:, :n].cpu().numpy()
    assert_frames_close(got, _exact(xin[:, :n].cpu().numpy(), n, FORWARD), what="strided forward vs float64")
    assert bool((yout[:, n:] == -4.0).all()), "nothing written between the frames"
    back = torch.full((batch, stride_in), -5.0, device=dev)
    api.fft_transform_batch_strided(s, yout, stride_out, back, stride_in, batch, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    err = float(((back[:, :n] / n - xin[:, :n]).abs().amax(dim=1)).max())
    assert err <= 2e-6 and bool((back[:, n:] == -5.0).all())
    api.fft_destroy_setup(s)


@pytest.mark.gpu
def test_special_bins(gpu, real4g):
    """the packed slot (X0, X_{N/2}), the rows kb = 0 and R2/2 (their own mirrors) and the seam k = N/2 -/+ 1: single tones whose
    spectrum is known exactly"""
    api, torch, dev = gpu
    n = 41472
    s = api.fft_new_setup(n, api.FFT_REAL)
    r2 = api.plan_describe(n, api.FFT_REAL)["n2"]
    t = np.arange(n, dtype=np.float64)
    bins = [0, 1, r2 // 2, r2, r2 + 1, n // 4, n // 2 - 1, n // 2]
    x = np.stack([np.cos(2 * np.pi * k * t / n) for k in bins]).astype(np.float32)
    got = _gpu(gpu, s, x, api.FFT_FORWARD).astype(np.float64)
    for row, k in enumerate(bins):
        want = np.zeros(n)
        if k == 0:
            want[0] = n
        elif k == n // 2:
            want[1] = n
        else:
            want[2 * k] = n / 2
        assert np.abs(got[row] - want).max() <= 2e-6 * n, f"tone at bin {k}: max deviation {np.abs(got[row] - want).max():.3e}"
    api.fft_destroy_setup(s)


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch", [(41472, 5), (96000, 3), (122880, 2), (41472, 130)])
def test_fused_convolution_three_sweeps(gpu, oracle, real4g, n, batch):
    """fft_convolve_frames_batch of these frames: columns forward, ONE fused row pass (forward rows, product with the packed filter
    spectrum, backward rows, in place in the workspace), columns backward -- == the three-launch composition == the oracle's call
    sequence (reference test/test.cpp:212-218: real frames, DC / Nyquist as real products simd/chowdsp_fft_impl_sse.cpp:1640-1644);
    in place; a batch that runs in several chunks on two lanes."""
    api, torch, dev = gpu
    s, so = api.fft_new_setup(n, api.FFT_REAL), oracle.new_setup(n, REAL)
    x = ol.uniform(batch * n, 700 + batch).reshape(batch, n)
    h = ol.uniform(n, 79).reshape(1, n)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    one = torch.full_like(xd, -3.0)
    api.fft_convolve_frames_batch(s, xd, H, one, 1.0 / n)
    torch.cuda.synchronize()
    api.set_option("fused_conv_generic", 0)
    three = torch.full_like(xd, -5.0)
    api.fft_convolve_frames_batch(s, xd, H, three, 1.0 / n)
    torch.cuda.synchronize()
    api.set_option("fused_conv_generic", 1)
    inplace = xd.clone()
    api.fft_convolve_frames_batch(s, inplace, H, inplace, 1.0 / n)
    torch.cuda.synchronize()
    assert_frames_close(one.cpu().numpy(), three.cpu().numpy(), tol=3e-7, maxabs_tol=1.5e-6, what=f"real N={n}: three sweeps vs three launches")
    assert torch.equal(inplace, one), "in place == out of place"
    # float64: circular convolution of every frame with h
    pick = sorted({0, batch // 2, batch - 1})
    want = np.fft.irfft(np.fft.rfft(x[pick].astype(np.float64), axis=1) * np.fft.rfft(h.astype(np.float64), axis=1), n=n, axis=1)
    assert_frames_close(one.cpu().numpy()[pick], want, what=f"real N={n} fused convolution vs float64 circular convolution", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
    if n % 25 != 0:
        hb = oracle.transform(so, h[0], FORWARD, False)
        ref = np.stack([oracle.transform(so, oracle.convolve(so, oracle.transform(so, x[i], FORWARD, False), hb, np.zeros(n, np.float32), 1.0 / n), BACKWARD, False) for i in pick])
        assert_frames_close(one.cpu().numpy()[pick], ref, what=f"real N={n} fused convolution vs the oracle's call sequence", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch", [(20736, 5), (46080, 3), (57600, 4), (82944, 2), (38400, 70)])
def test_complex_fused_convolution_three_sweeps(gpu, oracle, n, batch):
    """The complex counterpart (csrc/conv4_generic_kernels.hpp): long mixed-radix COMPLEX frames whose general two-pass plan has both
    factors at most 639 points -- columns forward, one fused row pass in place in the workspace, columns backward -- == the
    three-launch composition == the oracle's call sequence (reference test/test.cpp:161-166); in place; several chunks."""
    from oracle_lib import COMPLEX

    api, torch, dev = gpu
    d = api.plan_describe(n, api.FFT_COMPLEX)
    assert d["passes"] == 2 and max(d["n1"], d["n2"]) <= 639 and d["n1"] % 16 == 0 and d["n2"] % 16 == 0, d
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    x = ol.uniform(batch * 2 * n, 900 + batch).reshape(batch, 2 * n)
    h = ol.uniform(2 * n, 81).reshape(1, 2 * n)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    one = torch.full_like(xd, -3.0)
    api.fft_convolve_frames_batch(s, xd, H, one, 1.0 / n)
    torch.cuda.synchronize()
    api.set_option("fused_conv_generic", 0)
    three = torch.full_like(xd, -5.0)
    api.fft_convolve_frames_batch(s, xd, H, three, 1.0 / n)
    torch.cuda.synchronize()
    api.set_option("fused_conv_generic", 1)
    inplace = xd.clone()
    api.fft_convolve_frames_batch(s, inplace, H, inplace, 1.0 / n)
    torch.cuda.synchronize()
    assert_frames_close(one.cpu().numpy(), three.cpu().numpy(), tol=3e-7, maxabs_tol=1.5e-6, what=f"complex N={n}: three sweeps vs three launches")
    assert torch.equal(inplace, one), "in place == out of place"
    pick = sorted({0, batch // 2, batch - 1})
    xc = x[pick][:, 0::2].astype(np.float64) + 1j * x[pick][:, 1::2]
    hc = h[:, 0::2].astype(np.float64) + 1j * h[:, 1::2]
    yc = np.fft.ifft(np.fft.fft(xc, axis=1) * np.fft.fft(hc, axis=1), axis=1)
    want = np.empty((len(pick), 2 * n))
    want[:, 0::2], want[:, 1::2] = yc.real, yc.imag
    assert_frames_close(one.cpu().numpy()[pick], want, what=f"complex N={n} fused convolution vs float64 circular convolution", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
    hb = oracle.transform(so, h[0], FORWARD, False)
    ref = np.stack([oracle.transform(so, oracle.convolve(so, oracle.transform(so, x[i], FORWARD, False), hb, np.zeros(2 * n, np.float32), 1.0 / n), BACKWARD, False) for i in pick])
    assert_frames_close(one.cpu().numpy()[pick], ref, what=f"complex N={n} fused convolution vs the oracle's call sequence", maxabs_tol=MIXED_RADIX_MAXABS_TOL)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
