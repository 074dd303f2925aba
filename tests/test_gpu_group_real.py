"""Real N = 131072 in ONE launch on the static XCD groups (csrc/group_real_kernels.hpp): the 65536-point complex core with the
pair stage done in registers by pass B (forward) / pass A (backward), rows and columns dealt to the members as 8 + 8 mirrored.
Parity with the oracle (reference test/test.cpp:288-295 runs real 2^17), agreement with the real four-step's two launches, in GROUP
mode and in the fall-back mode, for batches that fill the groups partly / exactly / many times over; in place; strided; packed DC /
Nyquist slot and the two self-paired rows (0 and 128) checked bin by bin; the hand-off failure path.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, FORWARD, REAL

pytestmark = pytest.mark.gpu

N = 131072


@pytest.fixture(params=["groups", "groups-solo"])
def real_groups(gpu, request):
    api = gpu[0]
    default_min = int(api.get_option("fused_min_batch"))
    assert api.get_option("fused_two_pass") == 2 and api.get_option("group_real") == 1
    api.set_option("group_force_solo", 1 if request.param.endswith("solo") else 0)
    api.set_option("fused_min_batch", 1)
    yield api
    api.set_option("fused_min_batch", default_min)
    api.set_option("group_force_solo", 0)
    api.set_option("group_real", 1)


def _run(gpu, s, x, direction, one_launch):
    api, torch, dev = gpu
    api.set_option("group_real", 1 if one_launch else 0)
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    yd = torch.full_like(xd, -9.0)
    api.fft_transform_batch(s, xd, yd, direction)
    api.stream_synchronize()
    api.set_option("group_real", 1)
    return yd.cpu().numpy()


@pytest.mark.parametrize("batch", [1, 2, 31, 32, 33, 100])
def test_vs_oracle_and_four_step(gpu, oracle, real_groups, batch):
    api = real_groups
    s, so = api.fft_new_setup(N, api.FFT_REAL), oracle.new_setup(N, REAL)
    x = ol.uniform(batch * N, 4100 + batch).reshape(batch, N)
    pick = sorted({0, batch // 2, batch - 1})
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        got = _run(gpu, s, x, direction, True)
        two = _run(gpu, s, x, direction, False)
        assert_frames_close(got, two, tol=3e-7, maxabs_tol=1.5e-6, what=f"one launch vs real four-step, direction {direction}")
        want = oracle.transform_batch(so, x[pick], odir)
        assert_frames_close(got[pick], want, what=f"real N={N} batch={batch} dir={direction} on the groups")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_every_bin_of_the_special_rows(gpu, real_groups):
    """bins 0 / N/2 (packed slot 0), M/2, rows kb = 0 and 128 (self-paired through LDS in member 15) and the rows next to them, against
    a float64 FFT, forward; and the backward transform of that spectrum returns N x the input"""
    api, torch, dev = gpu
    s = api.fft_new_setup(N, api.FFT_REAL)
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, (3, N)).astype(np.float32)
    got = _run(gpu, s, x, api.FFT_FORWARD, True)
    ref = np.fft.rfft(x.astype(np.float64), axis=1)
    scale = np.sqrt(np.mean(np.abs(ref) ** 2))
    assert np.max(np.abs(got[:, 0] - ref[:, 0].real)) / scale < 2e-6, "X[0] in packed slot 0"
    assert np.max(np.abs(got[:, 1] - ref[:, N // 2].real)) / scale < 2e-6, "X[N/2] in packed slot 0"
    z = got[:, 2:].reshape(3, -1, 2)
    z = z[..., 0] + 1j * z[..., 1]  # bins 1 .. M-1
    err = np.abs(z - ref[:, 1:N // 2]) / scale
    k = np.arange(1, N // 2)
    for kb in (0, 1, 127, 128, 129, 255):
        assert np.max(err[:, (k % 256) == kb]) < 2e-6, f"row kb = {kb}"
    assert np.max(err) < 2e-6
    back = _run(gpu, s, got, api.FFT_BACKWARD, True)
    assert np.max(np.abs(back / N - x)) < 3e-6
    api.fft_destroy_setup(s)


def test_in_place_and_strided(gpu, oracle, real_groups):
    api, torch, dev = gpu
    batch, pad = 37, 64
    s, so = api.fft_new_setup(N, api.FFT_REAL), oracle.new_setup(N, REAL)
    x = ol.uniform(batch * N, 78).reshape(batch, N)
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        want = oracle.transform_batch(so, x[[0, 17, 36]], odir)
        xd = torch.from_numpy(x).to(dev)
        api.fft_transform_batch(s, xd, xd, direction)  # in place: pass A of a frame has read all of it before pass B writes
        api.stream_synchronize()
        assert_frames_close(xd.cpu().numpy()[[0, 17, 36]], want, what="one-launch real transform in place")
        stride = N + pad
        buf = torch.full((batch, stride), 123.0, device=dev)
        buf[:, :N] = torch.from_numpy(x).to(dev)
        out = torch.full((batch, stride), -7.0, device=dev)
        api.fft_transform_batch_strided(s, buf, stride, out, stride, batch, direction)
        api.stream_synchronize()
        o = out.cpu().numpy()
        assert_frames_close(o[[0, 17, 36], :N], want, what="one-launch real transform, strided")
        assert np.all(o[:, N:] == -7.0), "padding between output frames untouched"
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_many_launches_and_default_threshold(gpu):
    """batch >= fused_min_batch takes the one-launch form by default; 20 round trips back to back on one stream"""
    api, torch, dev = gpu
    assert api.get_option("group_real") == 1
    s = api.fft_new_setup(N, api.FFT_REAL)
    batch = int(api.get_option("fused_min_batch"))
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    x = torch.rand((batch, N), device=dev, generator=g) * 2 - 1
    y, z = torch.empty_like(x), torch.empty_like(x)
    for _ in range(20):
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    api.stream_synchronize()
    assert float((z / N - x).abs().max()) < 3e-6
    # the same spectrum as the real four-step (whole batch, every bin)
    api.set_option("group_real", 0)
    try:
        y4 = torch.empty_like(x)
        api.fft_transform_batch(s, x, y4, api.FFT_FORWARD)
        api.stream_synchronize()
    finally:
        api.set_option("group_real", 1)
    scale = float(y4.abs().max())
    assert float((y - y4).abs().max()) / scale < 2e-6
    api.fft_destroy_setup(s)


def test_handoff_failure_is_reported(gpu):
    """spin limit 1: the bounded waits of the hand-off DO run out; the launch reports it at the next synchronisation and the next launch is fine"""
    api, torch, dev = gpu
    s = api.fft_new_setup(N, api.FFT_REAL)
    batch = 2 * int(api.get_option("fused_min_batch"))
    x = torch.rand((batch, N), device=dev) * 2 - 1
    y = torch.empty_like(x)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    api.stream_synchronize()
    good = y.clone()
    api.set_option("group_spin_limit", 1)
    try:
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        with pytest.raises(api.FFTError):
            api.stream_synchronize()
    finally:
        api.set_option("group_spin_limit", 0)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    api.stream_synchronize()
    assert torch.equal(y, good)
    api.fft_destroy_setup(s)


@pytest.mark.parametrize("batch", [1, 33, 70])
def test_fused_convolution_on_the_groups(gpu, real_groups, batch):
    """fft_convolve_frames_batch of real N = 131072: the product rides on the forward group kernel's stores (pair stage and product in
    registers, DC / Nyquist as two real products -- reference simd/chowdsp_fft_impl_sse.cpp:1640-1644), the backward transform follows in
    place: == the composition of the three public calls == the float64 circular convolution, in GROUP mode and in the fall-back mode"""
    api, torch, dev = gpu
    rng = np.random.RandomState(910 + batch)
    x = rng.uniform(-1, 1, (batch, N)).astype(np.float32)
    h = rng.uniform(-1, 1, (1, N)).astype(np.float32)
    want = np.fft.irfft(np.fft.rfft(x.astype(np.float64), axis=1) * np.fft.rfft(h.astype(np.float64), axis=1), n=N, axis=1)
    s = api.fft_new_setup(N, api.FFT_REAL)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    one = torch.full_like(xd, 3.0)
    api.fft_convolve_frames_batch(s, xd, H, one, 1.0 / N)
    api.stream_synchronize()
    api.set_option("fused_conv_group", 0)
    try:
        three = torch.full_like(xd, 5.0)
        api.fft_convolve_frames_batch(s, xd, H, three, 1.0 / N)
        api.stream_synchronize()
    finally:
        api.set_option("fused_conv_group", 1)
    inplace = xd.clone()
    api.fft_convolve_frames_batch(s, inplace, H, inplace, 1.0 / N)
    api.stream_synchronize()
    got = one.cpu().numpy().astype(np.float64)
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    assert err.max() <= 1.5e-6, f"relative L2 {err.max():.3e}"
    assert_frames_close(one.cpu().numpy(), three.cpu().numpy(), tol=4e-7, maxabs_tol=2e-6, what="product on the group kernel's stores vs composition")
    assert np.array_equal(inplace.cpu().numpy(), one.cpu().numpy()), "in place == out of place"
    api.fft_destroy_setup(s)
