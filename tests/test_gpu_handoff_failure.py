"""The one-launch two-pass forms (group_twopass_kernels.hpp, group_fast_kernels.hpp) hand a frame from pass A to pass B
inside an XCD's L2 behind BOUNDED waits; a wait that runs out is reported through a pinned host word at the next library
synchronisation.  Two things are pinned here:

  1. the failure path, end to end: with the developer option "group_spin_limit" = 1 the waits DO run out; EVERY synchronous
     entry point that can have run such a launch has to return the error for that launch (and name it), and the next call
     has to start clean;
  2. the hand-off itself under the conditions the CDNA4 guide prescribes for any cross-workgroup hand-off that is not in
     its table of validated forms: uneven load from a second stream, consumers L1-warm (a group's slot is re-used for
     every frame: a stale L1 line would be read if the `sc1` loads did not bypass it), EVERY output word compared, many
     launches.  (Form used: plain stores -> s_waitcnt vmcnt(0) -> barrier -> flag; consumer `sc1` loads -- sound because
     producer and consumers are established to sit on ONE XCD, whose L2 is the point of coherence for both; DESIGN.md
     section 4.)
Reference contract behind both: a transform either produces the reference's result or says so -- the reference's
entry points cannot fail (chowdsp_fft.h:138-160), so a failure here must never pass silently.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import COMPLEX, FORWARD

pytestmark = pytest.mark.gpu

N = 65536


@pytest.fixture
def short_fuse(gpu):
    """batches of any size take the one-launch form; restored afterwards whatever happens"""
    api = gpu[0]
    default_min = int(api.get_option("fused_min_batch"))
    assert api.get_option("fused_two_pass") == 2
    api.set_option("fused_min_batch", 1)
    yield api
    api.set_option("group_spin_limit", 0)
    api.set_option("fused_min_batch", default_min)
    api.set_option("group_force_solo", 0)
    api.set_option("group_reg_polls", 200)
    # no report may be left behind for later tests
    try:
        api.stream_synchronize()
    except api.FFTError:
        pass


def _last_error():
    from chowdsp_fft_amd import _lib

    return _lib.lib().fft_gpu_last_error().decode()


def _expect_handoff_error(api, call, what):
    with pytest.raises(api.FFTError) as e:
        call()
    assert "did not complete its hand-offs" in str(e.value) or "did not complete its hand-offs" in _last_error(), f"{what}: {e.value}"


@pytest.mark.parametrize("n", [N, 1 << 17])
def test_stream_synchronize_reports_and_clears(gpu, oracle, short_fuse, n):
    api, torch, dev = gpu
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    batch = 512 if n == N else 256
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.rand((batch, 2 * n), device=dev, generator=g) * 2 - 1
    y = torch.empty_like(x)
    api.set_option("group_spin_limit", 1)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    _expect_handoff_error(api, lambda: api.stream_synchronize(), "stream_synchronize")
    api.stream_synchronize()  # the report was taken: a second synchronisation is clean
    api.set_option("group_spin_limit", 0)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    api.stream_synchronize()
    pick = [0, batch // 2, batch - 1]
    assert_frames_close(y[pick].cpu().numpy(), oracle.transform_batch(so, x[pick].cpu().numpy(), FORWARD), what="the launch after a failed one")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_next_launch_reports_an_unsynchronised_failure(gpu, short_fuse):
    """nobody synchronised through the library: the NEXT one-launch call on the stream refuses to run and says why"""
    api, torch, dev = gpu
    s = api.fft_new_setup(N, api.FFT_COMPLEX)
    x = torch.rand((512, 2 * N), device=dev)
    y = torch.empty_like(x)
    api.set_option("group_spin_limit", 1)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    torch.cuda.synchronize()  # not a library synchronisation: sees nothing
    api.set_option("group_spin_limit", 0)
    _expect_handoff_error(api, lambda: api.fft_transform_batch(s, x, y, api.FFT_FORWARD), "next launch")
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    api.stream_synchronize()
    api.fft_destroy_setup(s)


def test_multi_gpu_entry_points_return_the_error(gpu, oracle, short_fuse):
    api, torch, dev = gpu
    total, shards = 512, 2
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    x = torch.rand((total, 2 * N), device=dev, generator=g) * 2 - 1
    setups = [api.fft_new_setup(N, api.FFT_COMPLEX) for _ in range(shards)]
    ranges = [api.shard_range(total, shards, r) for r in range(shards)]
    xs = [x[f:f + c] for f, c in ranges]
    ys = [torch.empty_like(a) for a in xs]
    counts = [c for _, c in ranges]
    torch.cuda.synchronize()
    api.set_option("group_spin_limit", 1)
    _expect_handoff_error(api, lambda: api.fft_transform_batch_multi_gpu(setups, xs, ys, counts, api.FFT_FORWARD), "fft_transform_batch_multi_gpu")
    api.set_option("group_spin_limit", 0)
    api.fft_transform_batch_multi_gpu(setups, xs, ys, counts, api.FFT_FORWARD)
    so = oracle.new_setup(N, COMPLEX)
    got = torch.cat(ys)
    pick = [0, counts[0] - 1, counts[0], total - 1]
    assert_frames_close(got[pick].cpu().numpy(), oracle.transform_batch(so, x[pick].cpu().numpy(), FORWARD), what="multi-GPU call after a failed one")
    # the fused convolution of N = 65536 runs on the same groups
    h = torch.rand((1, 2 * N), device=dev)
    H = torch.empty_like(h)
    api.fft_transform_unordered_batch(setups[0], h, H, api.FFT_FORWARD)
    torch.cuda.synchronize()
    api.set_option("group_spin_limit", 1)
    _expect_handoff_error(api, lambda: api.fft_convolve_frames_batch_multi_gpu(setups, xs, [H] * shards, ys, counts, 1.0 / N), "fft_convolve_frames_batch_multi_gpu")
    api.set_option("group_spin_limit", 0)
    api.fft_convolve_frames_batch_multi_gpu(setups, xs, [H] * shards, ys, counts, 1.0 / N)
    oracle.destroy_setup(so)
    for s in setups:
        api.fft_destroy_setup(s)


def test_host_batch_returns_the_error(gpu, oracle, short_fuse):
    api, torch, dev = gpu
    batch = 200  # 64-frame chunks on two streams
    x = ol.uniform(batch * 2 * N, 8).reshape(batch, 2 * N)
    y = np.empty_like(x)
    s, so = api.fft_new_setup(N, api.FFT_COMPLEX), oracle.new_setup(N, COMPLEX)
    api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)  # warm: pipe buffers, pinning
    api.set_option("group_spin_limit", 1)
    _expect_handoff_error(api, lambda: api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD), "fft_transform_host_batch")
    api.set_option("group_spin_limit", 0)
    y[:] = 0
    api.fft_transform_host_batch(s, x, y, api.FFT_FORWARD)
    pick = [0, 63, 64, batch - 1]
    assert_frames_close(y[pick], oracle.transform_batch(so, x[pick], FORWARD), what="host batch after a failed one")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_deferred_flush_and_legacy_call_report_the_error(gpu, oracle, short_fuse):
    api, torch, dev = gpu
    s, so = api.fft_new_setup(N, api.FFT_COMPLEX), oracle.new_setup(N, COMPLEX)
    frames = 48
    bufs = []
    for i in range(frames):
        a, ap = api.aligned_array(2 * N)
        b, bp = api.aligned_array(2 * N)
        a[:] = ol.uniform(2 * N, 100 + i)
        bufs.append((a, b, ap, bp))

    def record_all():
        api.defer_begin()
        for a, b, _, _ in bufs:
            api.fft_transform(s, a, b, None, api.FFT_FORWARD)

    api.set_option("group_spin_limit", 1)
    record_all()
    _expect_handoff_error(api, lambda: api.defer_flush(), "fft_gpu_defer_flush")
    api.set_option("group_spin_limit", 0)
    record_all()
    api.defer_flush()
    for i in (0, frames - 1):
        assert_frames_close(bufs[i][1], oracle.transform(so, bufs[i][0], FORWARD), what="deferred calls after a failed flush")
    # the legacy symbol itself, device pointers, batch 1 (void: the reason is in fft_gpu_last_error)
    xd = torch.from_numpy(bufs[0][0].copy()).to(dev)
    yd = torch.empty_like(xd)
    torch.cuda.synchronize()
    api.set_option("group_spin_limit", 1)
    api.fft_transform(s, xd, yd, None, api.FFT_FORWARD)
    assert "did not complete its hand-offs" in _last_error()
    api.set_option("group_spin_limit", 0)
    api.fft_transform(s, xd, yd, None, api.FFT_FORWARD)
    assert_frames_close(yd.cpu().numpy(), oracle.transform(so, bufs[0][0], FORWARD), what="legacy call after a failed one")
    for _, _, ap, bp in bufs:
        api.aligned_free(ap)
        api.aligned_free(bp)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def _side_load(api, torch, side, s4, frames, scratch, small, rounds):
    """uneven load: short grids that occupy a few CUs each (48 frames of N = 4096 = 48 persistent workgroups; a tiny
    element-wise pass), a few of them whole-chip"""
    with torch.cuda.stream(side):
        for r in range(rounds):
            api.fft_transform_batch(s4, frames[:48], scratch[:48], api.FFT_FORWARD, stream=side)
            small.mul_(1.0000001)
            if r % 5 == 4:
                api.fft_transform_batch(s4, frames, scratch, api.FFT_FORWARD, stream=side)


@pytest.mark.parametrize("handoff", [0, 1])
def test_group_handoff_stress_every_word_uneven_load(gpu, short_fuse, handoff):
    """N = 65536, 2048 frames (1 GiB), 50 launches beside a second stream's uneven load: every output word == the two-launch
    form's (bit for bit: the same arithmetic in the same order).  Both hand-off forms: the fast one (option group_handoff = 0)
    and the one the CDNA4 guide lists (1: sc1 stores, one workgroup per CU)."""
    api, torch, dev = gpu
    api.set_option("group_handoff", handoff)
    batch = 2048
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    x = torch.rand((batch, 2 * N), device=dev, generator=g) * 2 - 1
    s = api.fft_new_setup(N, api.FFT_COMPLEX)
    s4 = api.fft_new_setup(4096, api.FFT_COMPLEX)
    want = {}
    api.set_option("fused_two_pass", 0)
    for d in (api.FFT_FORWARD, api.FFT_BACKWARD):
        want[d] = torch.empty_like(x)
        api.fft_transform_batch(s, x, want[d], d)
    torch.cuda.synchronize()
    api.set_option("fused_two_pass", 2)
    api.set_option("group_reg_polls", 5000)  # patient: the grid completes behind the side stream's short kernels instead of going SOLO
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    frames = torch.rand((8192, 8192), device=dev)
    scratch = torch.empty_like(frames)
    small = torch.rand((1 << 16,), device=dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    try:
        for launch in range(50):
            d = api.FFT_FORWARD if launch % 2 == 0 else api.FFT_BACKWARD
            _side_load(api, torch, side, s4, frames, scratch, small, 10)
            y.fill_(-1.0)
            main.wait_stream(torch.cuda.current_stream())
            api.fft_transform_batch(s, x, y, d, stream=main)
            api.stream_synchronize(main)
            assert torch.equal(y, want[d]), f"launch {launch}: {int((y != want[d]).sum())} words differ"
            side.synchronize()
    finally:
        api.set_option("fused_two_pass", 2)
        api.set_option("group_handoff", 0)
        api.fft_destroy_setup(s)
        api.fft_destroy_setup(s4)


@pytest.mark.parametrize("n,batch", [(1 << 17, 1024), (1 << 18, 512)])
def test_groupfast_handoff_stress_uneven_load(gpu, short_fuse, n, batch):
    """the same for the register-direct group form (2^17 / 2^18; its arithmetic differs from the two-launch kernels' in the
    last bit, so the bound is 2e-7 relative per frame on every frame instead of bit equality)"""
    api, torch, dev = gpu
    g = torch.Generator(device=dev)
    g.manual_seed(4243)
    x = torch.rand((batch, 2 * n), device=dev, generator=g) * 2 - 1
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    s4 = api.fft_new_setup(4096, api.FFT_COMPLEX)
    api.set_option("group_fast", 0)
    want = torch.empty_like(x)
    api.fft_transform_batch(s, x, want, api.FFT_FORWARD)
    torch.cuda.synchronize()
    api.set_option("group_fast", 1)
    api.set_option("group_reg_polls", 5000)
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    frames = torch.rand((8192, 8192), device=dev)
    scratch = torch.empty_like(frames)
    small = torch.rand((1 << 16,), device=dev)
    y = torch.empty_like(x)
    first = None
    torch.cuda.synchronize()
    try:
        for launch in range(20):
            _side_load(api, torch, side, s4, frames, scratch, small, 10)
            y.fill_(-1.0)
            main.wait_stream(torch.cuda.current_stream())
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD, stream=main)
            api.stream_synchronize(main)
            rel = (y - want).norm(dim=1) / want.norm(dim=1)
            assert float(rel.max()) <= 2e-7, f"launch {launch}: frame {int(rel.argmax())} off by {float(rel.max()):.2e}"
            if first is None:
                first = y.clone()
            else:
                assert torch.equal(y, first), f"launch {launch} differs from launch 0 in {int((y != first).sum())} words"
            side.synchronize()
    finally:
        api.set_option("group_fast", 1)
        api.fft_destroy_setup(s)
        api.fft_destroy_setup(s4)


def test_real_group_handoff_stress_uneven_load(gpu, short_fuse):
    """the same for the REAL one-launch form (N = 131072 on the N = 65536 groups, group_real_kernels.hpp: pair stage in registers, the
    packed side in 64-byte runs that neighbours of a group complete in L2): 2048 frames (1 GiB), 30 launches alternating forward / backward
    beside a second stream's uneven load; every frame within 3e-7 of the real four-step's result and every launch bit-identical to the first
    of its direction (a stale or torn hand-off would differ from launch to launch)"""
    api, torch, dev = gpu
    n, batch = 1 << 17, 2048
    g = torch.Generator(device=dev)
    g.manual_seed(4244)
    x = torch.rand((batch, n), device=dev, generator=g) * 2 - 1
    s = api.fft_new_setup(n, api.FFT_REAL)
    s4 = api.fft_new_setup(4096, api.FFT_COMPLEX)
    want = {}
    api.set_option("group_real", 0)
    for d in (api.FFT_FORWARD, api.FFT_BACKWARD):
        want[d] = torch.empty_like(x)
        api.fft_transform_batch(s, x, want[d], d)
    torch.cuda.synchronize()
    api.set_option("group_real", 1)
    api.set_option("group_reg_polls", 5000)
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    frames = torch.rand((8192, 8192), device=dev)
    scratch = torch.empty_like(frames)
    small = torch.rand((1 << 16,), device=dev)
    y = torch.empty_like(x)
    first = {}
    torch.cuda.synchronize()
    try:
        for launch in range(30):
            d = api.FFT_FORWARD if launch % 2 == 0 else api.FFT_BACKWARD
            _side_load(api, torch, side, s4, frames, scratch, small, 10)
            y.fill_(-1.0)
            main.wait_stream(torch.cuda.current_stream())
            api.fft_transform_batch(s, x, y, d, stream=main)
            api.stream_synchronize(main)
            rel = (y - want[d]).norm(dim=1) / want[d].norm(dim=1)
            assert float(rel.max()) <= 3e-7, f"launch {launch}: frame {int(rel.argmax())} off by {float(rel.max()):.2e}"
            if d not in first:
                first[d] = y.clone()
            else:
                assert torch.equal(y, first[d]), f"launch {launch} differs from the first launch of its direction in {int((y != first[d]).sum())} words"
            side.synchronize()
    finally:
        api.set_option("group_real", 1)
        api.fft_destroy_setup(s)
        api.fft_destroy_setup(s4)
