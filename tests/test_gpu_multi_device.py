"""SURVEY section 8 row e at the C ABI: fft_gpu_shard_range / fft_transform_batch_multi_gpu / fft_gather_multi_gpu.
Runs on min(device count, 2) devices: on a one-GPU lease the two shards share device 0 (the entry points allow that),
which still exercises the sharding rule, the per-shard host threads and streams, and the peer-copy gather; with two
devices every plan, buffer and stream lives on its own device."""
import numpy as np
import pytest

import oracle_lib as ol
from chowdsp_fft_amd import sharding
from conftest import assert_frames_close
from oracle_lib import COMPLEX, FORWARD, REAL


def test_shard_range_matches_the_torch_distributed_rule():
    from chowdsp_fft_amd import api

    for total in (0, 1, 7, 8, 65536, 1000003):
        for shards in (1, 2, 3, 8):
            covered = 0
            for r in range(shards):
                assert api.shard_range(total, shards, r) == sharding.shard_range(total, shards, r)
                covered += api.shard_range(total, shards, r)[1]
            assert covered == total
    with pytest.raises(api.FFTError):
        api.shard_range(10, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,tr,total", [(4096, COMPLEX, 37), (1024, REAL, 101), (65536, COMPLEX, 5), (480, COMPLEX, 9)])
@pytest.mark.parametrize("shards", [1, 2, 3])
def test_sharded_transform_and_gather(gpu, oracle, n, tr, total, shards):
    api, torch, dev0 = gpu
    from chowdsp_fft_amd import _lib

    ndev = min(api.device_count(), 2)
    devices = [d % ndev for d in range(shards)]
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(total * nf, 31 + n).reshape(total, nf)
    so = oracle.new_setup(n, tr)
    want = oracle.transform_batch(so, x, FORWARD)
    setups, ins, outs, counts = [], [], [], []
    for r in range(shards):
        first, count = api.shard_range(total, shards, r)
        d = devices[r]
        assert _lib.lib().fft_gpu_set_device(d) == 0
        s = api.fft_new_setup(n, tr)
        assert s is not None and _lib.lib().fft_gpu_setup_device(s) == d
        setups.append(s)
        ins.append(torch.from_numpy(x[first:first + count].copy()).to(torch.device("cuda", d)))
        outs.append(torch.empty_like(ins[-1]))
        counts.append(count)
    assert _lib.lib().fft_gpu_set_device(0) == 0  # the caller's current device is none of the call's business
    for d in range(ndev):
        torch.cuda.synchronize(d)
    api.fft_transform_batch_multi_gpu(setups, ins, outs, counts, api.FFT_FORWARD)  # synchronous
    for r in range(shards):
        first, count = api.shard_range(total, shards, r)
        if count:
            assert_frames_close(outs[r].cpu().numpy(), want[first:first + count], what=f"shard {r} of {shards}")
    # all-gather: every device ends up with all frames, in batch order
    dsts = [torch.full((total, nf), -3.0, device=torch.device("cuda", d)) for d in range(ndev)]
    for d in range(ndev):
        torch.cuda.synchronize(d)
    api.fft_gather_multi_gpu(outs, devices, dsts, list(range(ndev)))
    for d in range(ndev):
        assert_frames_close(dsts[d].cpu().numpy(), want, what=f"gathered on device {d}")
    # gather to one root only
    root = torch.full((total, nf), -5.0, device=torch.device("cuda", ndev - 1))
    torch.cuda.synchronize(ndev - 1)  # the fill runs on torch's stream, the copies on the library's
    api.fft_gather_multi_gpu(outs, devices, [None] * (ndev - 1) + [root], list(range(ndev)))
    assert_frames_close(root.cpu().numpy(), want, what="gather to root")
    for s in setups:
        api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.gpu
def test_two_real_devices_peer_gather(gpu, oracle):
    """Needs TWO devices (skipped on a one-GPU lease, where the test above folds every shard onto device 0): shards produced
    on two different devices by fft_transform_batch on streams of the CALLER's (no synchronisation before the gather: the
    gather waits for the devices itself), collected by one-hop peer copies in both directions."""
    api, torch, dev0 = gpu
    from chowdsp_fft_amd import _lib

    if api.device_count() < 2:
        pytest.skip("one device: hipMemcpyPeerAsync between two real devices cannot run")
    n, total = 4096, 64
    x = ol.uniform(total * 2 * n, 77).reshape(total, 2 * n)
    so = oracle.new_setup(n, COMPLEX)
    want = oracle.transform_batch(so, x, FORWARD)
    setups, outs, streams = [], [], []
    for r in range(2):
        first, count = api.shard_range(total, 2, r)
        assert _lib.lib().fft_gpu_set_device(r) == 0
        s = api.fft_new_setup(n, api.FFT_COMPLEX)
        assert _lib.lib().fft_gpu_setup_device(s) == r
        d = torch.device("cuda", r)
        xin = torch.from_numpy(x[first:first + count].copy()).to(d)
        y = torch.empty_like(xin)
        st = torch.cuda.Stream(device=d)
        torch.cuda.synchronize(r)
        api.fft_transform_batch(s, xin, y, api.FFT_FORWARD, stream=st)  # NOT synchronised: the gather must wait for it
        setups.append(s)
        outs.append(y)
        streams.append(st)
    assert _lib.lib().fft_gpu_set_device(0) == 0
    dsts = [torch.full((total, 2 * n), -3.0, device=torch.device("cuda", d)) for d in range(2)]
    api.fft_gather_multi_gpu(outs, [0, 1], dsts, [0, 1])
    for d in range(2):
        assert_frames_close(dsts[d].cpu().numpy(), want, what=f"all-gather on real device {d}")
    for s in setups:
        api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.gpu
def test_release_stream_resources(gpu):
    api, torch, dev = gpu
    s = api.fft_new_setup(1 << 17, api.FFT_COMPLEX)  # two-pass: caches lanes + workspaces for the stream
    st = torch.cuda.Stream()
    x = torch.rand((3, 2 << 17), device=dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD, stream=st)
    st.synchronize()
    assert api.release_stream_resources(st) == 1
    assert api.release_stream_resources(st) == 0
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD, stream=st)  # builds a fresh set
    st.synchronize()
    ref = torch.view_as_real(torch.fft.fft(torch.view_as_complex(x.view(3, -1, 2).contiguous()), dim=1)).reshape(3, -1)
    assert float(((y - ref).norm(dim=1) / ref.norm(dim=1)).max()) < 1e-6
    assert api.release_stream_resources(st) == 1
    api.fft_destroy_setup(s)


def _oracle_conv_frames(oracle, so, xs, h, n):
    hb = oracle.transform(so, h, FORWARD, False)
    return np.stack([oracle.transform(so, oracle.convolve(so, oracle.transform(so, x, FORWARD, False), hb, np.zeros(len(h), np.float32), 1.0 / n),
                                      ol.BACKWARD, False) for x in xs])


@pytest.mark.gpu
@pytest.mark.parametrize("n,tr,total,shards", [(4096, REAL, 1001, 3), (4096, REAL, 64, 8), (480, COMPLEX, 77, 2), (65536, COMPLEX, 9, 2)])
def test_sharded_fused_convolution_small(gpu, oracle, n, tr, total, shards):
    """fft_convolve_frames_batch_multi_gpu == the oracle's forward -> convolve -> backward sequence (reference test/test.cpp:161-166),
    shard by shard, filter replicated per device, in place and out of place"""
    api, torch, dev0 = gpu
    from chowdsp_fft_amd import _lib

    ndev = min(api.device_count(), 2)
    devices = [d % ndev for d in range(shards)]
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(total * nf, 3 + n).reshape(total, nf)
    h = ol.uniform(nf, 4 + n)
    so = oracle.new_setup(n, tr)
    pick = np.unique(np.concatenate([[0, total - 1], [api.shard_range(total, shards, r)[0] for r in range(shards)]]))
    pick = pick[pick < total]
    want = _oracle_conv_frames(oracle, so, x[pick], h, n)
    setups, xs, ys, Hs, counts = [], [], [], [], []
    Hdev = {}
    for r in range(shards):
        first, count = api.shard_range(total, shards, r)
        d = devices[r]
        assert _lib.lib().fft_gpu_set_device(d) == 0
        s = api.fft_new_setup(n, tr)
        setups.append(s)
        td = torch.device("cuda", d)
        if d not in Hdev:  # the filter spectrum, replicated: computed once per device
            hd = torch.from_numpy(h.reshape(1, nf)).to(td)
            Hdev[d] = torch.empty_like(hd)
            api.fft_transform_unordered_batch(s, hd, Hdev[d], api.FFT_FORWARD)
        xs.append(torch.from_numpy(x[first:first + count].copy()).to(td))
        ys.append(torch.empty_like(xs[-1]))
        Hs.append(Hdev[d])
        counts.append(count)
    assert _lib.lib().fft_gpu_set_device(0) == 0
    for d in range(ndev):
        torch.cuda.synchronize(d)
    api.fft_convolve_frames_batch_multi_gpu(setups, xs, Hs, ys, counts, 1.0 / n)
    got = np.concatenate([y.cpu().numpy() for y in ys])
    assert_frames_close(got[pick], want, what="sharded fused convolution")
    api.fft_convolve_frames_batch_multi_gpu(setups, xs, Hs, xs, counts, 1.0 / n)  # in place
    got2 = np.concatenate([y.cpu().numpy() for y in xs])
    assert np.array_equal(got, got2)
    for s in setups:
        api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.gpu
def test_config4_full_one_million_frames_folded(gpu, oracle):
    """BASELINE configs[3] at its GLOBAL shape: 1 048 576 real N = 4096 overlap-save frames (16 GiB) as 8 shards
    (fft_gpu_shard_range), fused convolution per shard through fft_convolve_frames_batch_multi_gpu, then fft_gather_multi_gpu.
    On a one-GPU lease the 8 shards fold onto device 0 (the entry points allow it; a 288 GB MI355X holds the whole job);
    with more devices they spread.  Frames {0, every shard boundary - 1 / + 0, last} == the oracle's pipeline; EVERY frame ==
    the single-call fft_convolve_frames_batch of the whole batch (bit for bit: same kernel, frames are independent)."""
    api, torch, dev0 = gpu
    from chowdsp_fft_amd import _lib

    n, total, shards = 4096, 1 << 20, 8
    free, _ = torch.cuda.mem_get_info(0)
    ndev = min(api.device_count(), shards)
    if ndev == 1 and free < 40 * (1 << 30):
        pytest.skip("needs 40 GiB of free device memory (16 GiB frames + 16 GiB gathered + scratch)")
    if ndev > 1:
        pytest.skip("folded form is for one device; the spread form is test_sharded_fused_convolution_small")
    chunk = 65536  # frames per generator call: any frame's input can be regenerated from (seed, chunk)

    def gen_chunk(c, out=None):
        g = torch.Generator(device=dev0)
        g.manual_seed(5000 + c)
        r = torch.rand((chunk, n), device=dev0, generator=g)
        r = r * 2 - 1
        if out is not None:
            out.copy_(r)
        return r

    def fill(buf):
        for c in range(total // chunk):
            gen_chunk(c, buf[c * chunk:(c + 1) * chunk])

    x = torch.empty((total, n), device=dev0)
    fill(x)
    h = ol.uniform(n, 99)
    s = api.fft_new_setup(n, api.FFT_REAL)
    hd = torch.from_numpy(h.reshape(1, n)).to(dev0)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    setups = [api.fft_new_setup(n, api.FFT_REAL) for _ in range(shards)]  # replicated plans (one per shard)
    ranges = [api.shard_range(total, shards, r) for r in range(shards)]
    assert sum(c for _, c in ranges) == total and ranges[0][1] == 131072
    xs = [x[f:f + c] for f, c in ranges]
    torch.cuda.synchronize()
    api.fft_convolve_frames_batch_multi_gpu(setups, xs, [H] * shards, xs, [c for _, c in ranges], 1.0 / n)  # in place, synchronous
    dst = torch.empty((total, n), device=dev0)
    torch.cuda.synchronize()
    api.fft_gather_multi_gpu(xs, [0] * shards, [dst], [0])
    # oracle pipeline on the boundary frames
    pick = sorted({0, total - 1} | {f - 1 for f, _ in ranges[1:]} | {f for f, _ in ranges[1:]})
    so = oracle.new_setup(n, REAL)
    inputs = []
    for i in pick:
        inputs.append(gen_chunk(i // chunk)[i % chunk].cpu().numpy())
    want = _oracle_conv_frames(oracle, so, inputs, h, n)
    assert_frames_close(dst[pick].cpu().numpy(), want, what="1 M frames folded: shard boundaries")
    oracle.destroy_setup(so)
    # every frame against ONE call over the whole batch
    fill(x)
    api.fft_convolve_frames_batch(s, x, H, x, 1.0 / n)
    torch.cuda.synchronize()
    for c in range(total // chunk):
        sl = slice(c * chunk, (c + 1) * chunk)
        assert torch.equal(dst[sl], x[sl]), f"frames {sl} differ between the sharded and the single call"
    for p in setups + [s]:
        api.fft_destroy_setup(p)
    del x, dst
    torch.cuda.empty_cache()
