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
