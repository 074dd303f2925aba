"""The one-launch two-pass forms: both passes of N = 65536 complex in ONE kernel, the inter-pass hand-off inside an XCD's
L2 -- work handed out by per-XCD ticket queues (csrc/fused_twopass_kernels.hpp, option value 1) or done by static groups
of 16 workgroups per XCD (csrc/group_twopass_kernels.hpp, option value 2; with its fall-back mode in which every
workgroup transforms whole frames alone).  Forced on for every batch size here
(an option, off by default; when on, batches >= 2048 take it): parity with the oracle, and agreement with the two-launch form, for
batches that fill the queue's pipeline only partly, exactly, and many times over; both directions; in place; strided.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD

pytestmark = pytest.mark.gpu

N = 65536


FORM = {"tickets": 1, "groups": 2}


def _form_now(api):
    return int(api.get_option("fused_two_pass"))


@pytest.fixture(params=["tickets", "groups", "groups-solo"])
def fused_on(gpu, request):
    api = gpu[0]
    experiments = api.get_option("experiments") == 1
    if request.param == "tickets" and not experiments:
        pytest.skip("the ticket-queue form is a superseded form: only libraries built with EXPERIMENTS=1 contain it")
    default, default_min = _form_now(api), int(api.get_option("fused_min_batch"))
    api.set_option("fused_two_pass", FORM[request.param.split("-")[0]])
    api.set_option("group_force_solo", 1 if request.param.endswith("solo") else 0)
    api.set_option("fused_min_batch", 1)
    yield api
    api.set_option("fused_min_batch", default_min)
    if experiments:
        api.set_option("fused_lag", 4)
    api.set_option("group_force_solo", 0)
    api.set_option("fused_two_pass", default)


def _run(gpu, s, x, direction, form):
    api, torch, dev = gpu
    api.set_option("fused_two_pass", form)
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, direction)
    api.stream_synchronize()  # where the one-launch forms report a hand-off that did not complete
    return yd.cpu().numpy()




@pytest.mark.parametrize("batch", [1, 2, 7, 8, 9, 31, 64, 100])
@pytest.mark.parametrize("lag", [1, 3, 4])
def test_fused_vs_oracle_and_two_launch(gpu, oracle, fused_on, batch, lag):
    api = fused_on
    if _form_now(api) == 2 and lag != 4:
        pytest.skip("the group form has no lag")
    if _form_now(api) == 1:
        api.set_option("fused_lag", lag)
    s, so = api.fft_new_setup(N, api.FFT_COMPLEX), oracle.new_setup(N, COMPLEX)
    x = ol.uniform(batch * 2 * N, 900 + batch).reshape(batch, 2 * N)
    nref = min(batch, 3)
    form = _form_now(api)
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        got = _run(gpu, s, x, direction, form)
        two = _run(gpu, s, x, direction, 0)
        api.set_option("fused_two_pass", form)
        # same algorithm and tables; the compiler contracts the two kernels' multiply-adds differently
        assert_frames_close(got, two, tol=2e-7, maxabs_tol=6e-7, what="one-launch vs two-launch form")
        want = oracle.transform_batch(so, x[:nref], odir)
        assert_frames_close(got[:nref], want, what=f"fused N={N} batch={batch} lag={lag} dir={direction}")
        last = oracle.transform_batch(so, x[-1:], odir)
        assert_frames_close(got[-1:], last, what="last frame of the batch (queue drain)")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_fused_in_place_and_strided(gpu, oracle, fused_on):
    api, torch, dev = gpu
    batch, pad = 37, 64
    s, so = api.fft_new_setup(N, api.FFT_COMPLEX), oracle.new_setup(N, COMPLEX)
    x = ol.uniform(batch * 2 * N, 77).reshape(batch, 2 * N)
    want = oracle.transform_batch(so, x[[0, 17, 36]], FORWARD)
    xd = torch.from_numpy(x).to(dev)
    api.fft_transform_batch(s, xd, xd, api.FFT_FORWARD)  # in place: pass A of a frame has read all of it before pass B writes
    torch.cuda.synchronize()
    assert_frames_close(xd.cpu().numpy()[[0, 17, 36]], want, what="fused in place")
    stride = 2 * N + pad
    buf = torch.full((batch, stride), 123.0, device=dev)
    buf[:, :2 * N] = torch.from_numpy(x).to(dev)
    out = torch.full((batch, stride), -7.0, device=dev)
    api.fft_transform_batch_strided(s, buf, stride, out, stride, batch, api.FFT_FORWARD)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert_frames_close(o[[0, 17, 36], :2 * N], want, what="fused strided")
    assert np.all(o[:, 2 * N:] == -7.0), "padding between output frames untouched"
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_fused_many_launches_back_to_back(gpu, fused_on):
    """state and workspace are re-used by consecutive launches on one stream (tickets, claims and flags start from zero)"""
    api, torch, dev = gpu
    s = api.fft_new_setup(N, api.FFT_COMPLEX)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.rand((48, 2 * N), device=dev, generator=g) * 2 - 1
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    for _ in range(20):
        api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
    api.stream_synchronize()
    assert float((z / N - x).abs().max()) < 3e-6
    api.fft_destroy_setup(s)


def test_set_option_rejects_nonsense(gpu):
    api = gpu[0]
    with pytest.raises(api.FFTError):
        api.set_option("no_such_option", 1)
    with pytest.raises(api.FFTError):
        api.set_option("fused_lag", 99)
    if api.get_option("experiments") != 1:
        # the shipped library carries one form per size: the superseded forms' options are refused
        for name, value in (("fused_two_pass", 1), ("wave4096", 1), ("split8192", 1), ("conv_wave2048", 1), ("fused_lag", 4)):
            with pytest.raises(api.FFTError):
                api.set_option(name, value)
        assert api.get_option("fused_two_pass") == 2


@pytest.mark.parametrize("solo", [0, 1])
@pytest.mark.parametrize("batch", [1, 2, 31, 70])
def test_group_convolution_n65536(gpu, oracle, batch, solo):
    """fft_convolve_frames_batch of N = 65536 complex frames in ONE launch on the static groups (forward passes A and B, the
    product, the backward transform started in pass B's registers, its second pass), in GROUP mode and in the fall-back mode,
    == the three-launch composition == the oracle's call sequence (reference test/test.cpp:161-166)."""
    api, torch, dev = gpu
    default_min = int(api.get_option("fused_min_batch"))
    assert api.get_option("fused_two_pass") == 2 and api.get_option("fused_conv_group") == 1
    api.set_option("fused_min_batch", 1)
    api.set_option("group_force_solo", solo)
    try:
        s, so = api.fft_new_setup(N, api.FFT_COMPLEX), oracle.new_setup(N, COMPLEX)
        x = ol.uniform(batch * 2 * N, 1300 + batch).reshape(batch, 2 * N)
        h = ol.uniform(2 * N, 77).reshape(1, 2 * N)
        xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
        H = torch.empty_like(hd)
        api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
        one = torch.full_like(xd, -3.0)
        api.fft_convolve_frames_batch(s, xd, H, one, 1.0 / N)
        api.stream_synchronize()
        api.set_option("fused_conv_group", 0)
        three = torch.full_like(xd, -5.0)
        api.fft_convolve_frames_batch(s, xd, H, three, 1.0 / N)
        api.set_option("fused_conv_group", 1)
        inplace = xd.clone()
        api.fft_convolve_frames_batch(s, inplace, H, inplace, 1.0 / N)  # a member reads its columns of a frame before anybody writes it
        api.stream_synchronize()
        assert_frames_close(one.cpu().numpy(), three.cpu().numpy(), tol=3e-7, maxabs_tol=1e-6, what="one launch vs three launches")
        assert np.array_equal(inplace.cpu().numpy(), one.cpu().numpy()), "in place == out of place"
        hb = oracle.transform(so, h[0], FORWARD, False)
        pick = sorted({0, batch // 2, batch - 1})
        want = np.stack([oracle.transform(so, oracle.convolve(so, oracle.transform(so, x[i], FORWARD, False), hb, np.zeros(2 * N, np.float32), 1.0 / N), BACKWARD, False) for i in pick])
        assert_frames_close(one.cpu().numpy()[pick], want, what="fused convolution vs the oracle's call sequence")
        api.fft_destroy_setup(s)
        oracle.destroy_setup(so)
    finally:
        api.set_option("group_force_solo", 0)
        api.set_option("fused_min_batch", default_min)
        api.set_option("fused_conv_group", 1)


@pytest.mark.parametrize("polls", [0, 1, 3])
def test_group_mode_decision_race(gpu, polls):
    """group_reg_polls = 0 / 1 / 3: a workgroup gives up waiting for the grid to be complete almost at once, so the launch's mode
    (GROUP by the workgroup whose registration completes the grid, SOLO by whoever loses patience first -- one compare-and-swap)
    is decided in a RACE, differently from launch to launch.  Whatever it comes out as, every workgroup must follow the same
    decision and the result must be the transform: 30 launches (plain and fused convolution) against the two-launch form."""
    api, torch, dev = gpu
    default_min = int(api.get_option("fused_min_batch"))
    api.set_option("fused_min_batch", 1)
    api.set_option("group_reg_polls", polls)
    try:
        s = api.fft_new_setup(N, api.FFT_COMPLEX)
        g = torch.Generator(device=dev)
        g.manual_seed(11 + polls)
        x = torch.rand((40, 2 * N), device=dev, generator=g) * 2 - 1
        h = torch.rand((1, 2 * N), device=dev, generator=g) * 2 - 1
        H = torch.empty_like(h)
        api.fft_transform_unordered_batch(s, h, H, api.FFT_FORWARD)
        api.set_option("fused_two_pass", 0)
        api.set_option("fused_conv_group", 0)
        want, want_inv, want_conv = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        api.fft_transform_batch(s, x, want, api.FFT_FORWARD)
        api.fft_transform_batch(s, x, want_inv, api.FFT_BACKWARD)
        api.fft_convolve_frames_batch(s, x, H, want_conv, 1.0 / N)
        api.set_option("fused_two_pass", 2)
        api.set_option("fused_conv_group", 1)
        y = torch.empty_like(x)
        for rep in range(10):
            for direction, ref in ((api.FFT_FORWARD, want), (api.FFT_BACKWARD, want_inv)):
                y.fill_(-1.0)
                api.fft_transform_batch(s, x, y, direction)
                api.stream_synchronize()
                assert torch.equal(y, ref), f"launch {rep}, direction {direction}, polls {polls}"
            y.fill_(-1.0)
            api.fft_convolve_frames_batch(s, x, H, y, 1.0 / N)
            api.stream_synchronize()
            err = float(((y - want_conv).norm(dim=1) / want_conv.norm(dim=1)).max())
            assert err < 3e-7, f"fused convolution, launch {rep}: {err}"
        api.fft_destroy_setup(s)
    finally:
        api.set_option("group_reg_polls", 200)
        api.set_option("fused_min_batch", default_min)
        api.set_option("fused_two_pass", 2)
        api.set_option("fused_conv_group", 1)


def test_group_form_beside_another_streams_kernels(gpu):
    """The group form's grid is the whole chip; beside another stream's kernels it may not be completely resident.  Then its
    workgroups register late or not at all within `group_reg_polls`, the launch falls back to its SOLO mode by itself (or
    not, depending on timing) -- and must be correct either way.  Here: long-running kernels of a second stream (torch
    element-wise passes and our own N = 4096 transforms, which fill every CU) run while group-form transforms and fused
    convolutions are launched on the first, with short and with normal patience."""
    api, torch, dev = gpu
    default_min = int(api.get_option("fused_min_batch"))
    api.set_option("fused_min_batch", 1)
    try:
        s = api.fft_new_setup(N, api.FFT_COMPLEX)
        s4 = api.fft_new_setup(4096, api.FFT_COMPLEX)
        g = torch.Generator(device=dev)
        g.manual_seed(99)
        x = torch.rand((96, 2 * N), device=dev, generator=g) * 2 - 1
        api.set_option("fused_two_pass", 0)
        want = torch.empty_like(x)
        api.fft_transform_batch(s, x, want, api.FFT_FORWARD)
        torch.cuda.synchronize()
        api.set_option("fused_two_pass", 2)
        side = torch.cuda.Stream()
        main = torch.cuda.Stream()
        noise = torch.rand((1 << 28,), device=dev)            # 1 GiB
        frames = torch.rand((16384, 8192), device=dev)       # 512 MiB of N = 4096 frames
        scratch = torch.empty_like(frames)
        y = torch.empty_like(x)
        for polls in (2, 200):
            api.set_option("group_reg_polls", polls)
            for rep in range(6):
                with torch.cuda.stream(side):
                    for _ in range(4):
                        noise.mul_(1.0000001)
                        api.fft_transform_batch(s4, frames, scratch, api.FFT_FORWARD, stream=side)
                y.fill_(-1.0)
                main.wait_stream(torch.cuda.current_stream())
                api.fft_transform_batch(s, x, y, api.FFT_FORWARD, stream=main)
                api.stream_synchronize(main)
                side.synchronize()
                assert torch.equal(y, want), f"polls {polls}, launch {rep}"
        api.fft_destroy_setup(s)
        api.fft_destroy_setup(s4)
    finally:
        api.set_option("group_reg_polls", 200)
        api.set_option("fused_min_batch", default_min)
        api.set_option("fused_two_pass", 2)


@pytest.mark.parametrize("solo", [0, 1])
@pytest.mark.parametrize("n,batch", [(1 << 17, 1), (1 << 17, 37), (1 << 18, 3), (1 << 18, 21)])
def test_group_form_on_the_register_direct_tiles(gpu, oracle, n, batch, solo):
    """N = 2^17 (512 x 256: two groups of 16 workgroups per XCD) and 2^18 (512 x 512: one group of 32) in ONE launch
    (group_fast_kernels.hpp), GROUP and fall-back mode, both directions, in place and strided == the two-launch form == the oracle."""
    api, torch, dev = gpu
    default_min = int(api.get_option("fused_min_batch"))
    assert api.get_option("fused_two_pass") == 2 and api.get_option("group_fast") == 1
    api.set_option("fused_min_batch", 1)
    api.set_option("group_force_solo", solo)
    try:
        s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
        x = ol.uniform(batch * 2 * n, 4000 + batch).reshape(batch, 2 * n)
        xd = torch.from_numpy(x).to(dev)
        for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
            one = torch.full_like(xd, -3.0)
            api.fft_transform_batch(s, xd, one, direction)
            api.stream_synchronize()
            api.set_option("group_fast", 0)
            two = torch.full_like(xd, -5.0)
            api.fft_transform_batch(s, xd, two, direction)
            torch.cuda.synchronize()
            api.set_option("group_fast", 1)
            assert_frames_close(one.cpu().numpy(), two.cpu().numpy(), tol=2e-7, maxabs_tol=6e-7, what=f"N={n} one launch vs two launches")
            pick = sorted({0, batch - 1})
            want = oracle.transform_batch(so, x[pick], odir)
            assert_frames_close(one.cpu().numpy()[pick], want, what=f"N={n} batch={batch} dir={direction} solo={solo}")
        inplace = xd.clone()
        api.fft_transform_batch(s, inplace, inplace, api.FFT_FORWARD)
        stride = 2 * n + 64
        buf = torch.full((batch, stride), 123.0, device=dev)
        buf[:, :2 * n] = xd
        outb = torch.full((batch, stride), -7.0, device=dev)
        api.fft_transform_batch_strided(s, buf, stride, outb, stride, batch, api.FFT_FORWARD)
        api.stream_synchronize()
        ref = torch.empty_like(xd)
        api.set_option("group_fast", 0)
        api.fft_transform_batch(s, xd, ref, api.FFT_FORWARD)
        torch.cuda.synchronize()
        api.set_option("group_fast", 1)
        assert torch.equal(inplace, outb[:, :2 * n].contiguous()), "in place == strided"
        assert_frames_close(inplace.cpu().numpy(), ref.cpu().numpy(), tol=2e-7, maxabs_tol=6e-7, what="in place vs two launches")
        assert bool((outb[:, 2 * n:] == -7.0).all()), "padding between output frames untouched"
        api.fft_destroy_setup(s)
        oracle.destroy_setup(so)
    finally:
        api.set_option("group_force_solo", 0)
        api.set_option("fused_min_batch", default_min)
        api.set_option("group_fast", 1)
