"""Device tables are counted and given back (the reference frees everything a setup owns in fft_destroy_setup,
/root/reference/simd/chowdsp_fft_impl_sse.cpp:79-82; /root/reference/chowdsp_fft.h:119).

Each plan holds one reference to the tables of its (device, N, transform); tables without a plan are idle: cached up to
option "table_cache_mib" of device memory, freed by fft_gpu_trim_tables (and by fft_gpu_release_stream_resources).
A process that plans many sizes must not only grow: planning every accepted complex size up to 2^21 twice leaves the
device where the first sweep left it, and a trim brings it back to the start."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu

MIB = 1 << 20


def _free_bytes(torch):
    torch.cuda.synchronize()
    return int(torch.cuda.mem_get_info()[0])


@pytest.fixture()
def clean_cache(gpu):
    api = gpu[0]
    limit = api.get_option("table_cache_mib")
    api.trim_tables()
    yield api
    api.set_option("table_cache_mib", limit)
    api.trim_tables()


def test_counts_follow_the_plans(clean_cache):
    api = clean_cache
    base = api.table_cache_info()
    a = api.fft_new_setup(7200, api.FFT_COMPLEX)  # (a length no other test plans)
    b = api.fft_new_setup(7200, api.FFT_COMPLEX)
    one = api.table_cache_info()
    assert one["entries"] == base["entries"] + 1, "two plans of one size share one set of tables"
    assert one["idle_entries"] == base["idle_entries"]
    api.fft_destroy_setup(a)
    assert api.table_cache_info()["idle_entries"] == base["idle_entries"], "the second plan still holds them"
    api.fft_destroy_setup(b)
    idle = api.table_cache_info()
    assert idle["idle_entries"] == base["idle_entries"] + 1 and idle["idle_bytes"] > base["idle_bytes"]
    # a new plan of the same size takes the cached set instead of building one
    c = api.fft_new_setup(7200, api.FFT_COMPLEX)
    again = api.table_cache_info()
    assert again["entries"] == one["entries"] and again["idle_entries"] == base["idle_entries"]
    api.fft_destroy_setup(c)
    freed = api.trim_tables()
    assert freed >= idle["idle_bytes"] - base["idle_bytes"] > 0
    assert api.table_cache_info()["idle_entries"] == 0


def test_cache_limit_zero_frees_with_the_last_plan(clean_cache):
    api = clean_cache
    api.set_option("table_cache_mib", 0)
    base = api.table_cache_info()
    s = api.fft_new_setup(12800, api.FFT_COMPLEX)  # (a length no other test plans: plans other tests never destroyed stay live)
    assert api.table_cache_info()["entries"] == base["entries"] + 1
    api.fft_destroy_setup(s)
    assert api.table_cache_info() == base


def test_composite_plans_hold_their_parts(clean_cache, oracle):
    """real N = 131072 runs on the complex 65536-point plan's tables: they stay while the real plan lives and go with it"""
    api, n = clean_cache, 131072
    api.set_option("table_cache_mib", 0)
    base = api.table_cache_info()
    s = api.fft_new_setup(n, api.FFT_REAL)
    held = api.table_cache_info()
    # the real plan's own tables, and the complex 65536-point plan's unless a plan another test never destroyed already holds those
    assert held["entries"] >= base["entries"] + 1 and held["idle_entries"] == 0
    core = api.fft_new_setup(65536, api.FFT_COMPLEX)
    assert api.table_cache_info()["entries"] == held["entries"], "the part is the same cached set"
    api.fft_destroy_setup(core)
    assert api.table_cache_info()["entries"] == held["entries"], "... and the composite plan keeps it alive"
    # still correct after the part's own plan has gone
    import torch

    so = oracle.new_setup(n, REAL)
    x = ol.uniform(4 * n, 77).reshape(4, n)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD)
    torch.cuda.synchronize()
    assert_frames_close(yd.cpu().numpy(), oracle.transform_batch(so, x, FORWARD), what="real 131072 after its core's plan was destroyed")
    oracle.destroy_setup(so)
    api.fft_destroy_setup(s)
    assert api.table_cache_info() == base


def test_recorded_calls_keep_their_tables(clean_cache, oracle):
    """new_setup -> fft_transform -> destroy inside a deferred region with nothing cached: the recorded call's own reference
    keeps the tables until the flush has run it"""
    api, n = clean_cache, 2560  # (a length no other test plans)
    api.set_option("table_cache_mib", 0)
    base = api.table_cache_info()
    x, xp = api.aligned_array(2 * n)
    y, yp = api.aligned_array(2 * n)
    x[:] = ol.uniform(2 * n, 5)
    api.defer_begin()
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    api.fft_transform(s, x, y, None, api.FFT_FORWARD)
    api.fft_destroy_setup(s)
    assert api.table_cache_info()["entries"] == base["entries"] + 1, "held by the recorded call"
    api.defer_flush()
    assert api.table_cache_info() == base
    so = oracle.new_setup(n, COMPLEX)
    assert_frames_close(y, oracle.transform(so, np.array(x), FORWARD), what="recorded call after destroy")
    oracle.destroy_setup(so)
    api.aligned_free(xp)
    api.aligned_free(yp)


def test_planning_every_size_twice_does_not_grow(clean_cache):
    api = clean_cache
    import torch

    sizes = [n for n in range(16, (1 << 21) + 1, 16) if api.size_supported(n, api.FFT_COMPLEX)]
    assert len(sizes) == 332
    limit = api.get_option("table_cache_mib") * MIB

    def sweep():
        for n in sizes:
            s = api.fft_new_setup(n, api.FFT_COMPLEX)
            assert s is not None, n
            api.fft_destroy_setup(s)

    start = _free_bytes(torch)
    live = api.table_cache_info()["entries"]  # (plans other tests still hold; none idle after the fixture's trim)
    sweep()
    after1, info1 = _free_bytes(torch), api.table_cache_info()
    assert info1["idle_bytes"] <= limit and info1["entries"] - info1["idle_entries"] == live
    assert start - after1 <= limit + 8 * MIB, f"one sweep keeps {(start - after1) / MIB:.1f} MiB"
    sweep()
    after2, info2 = _free_bytes(torch), api.table_cache_info()
    assert info2 == info1
    assert abs(after2 - after1) <= MIB, f"second sweep moved free memory by {(after2 - after1) / MIB:.2f} MiB"
    api.trim_tables()
    assert api.table_cache_info()["entries"] == live
    assert abs(_free_bytes(torch) - start) <= MIB, "trim gives everything back"
