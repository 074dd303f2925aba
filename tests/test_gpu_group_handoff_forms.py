"""Option "group_handoff" (include/chowdsp_fft_gpu.h): the inter-workgroup hand-off of the one-launch two-pass forms either in
the fast form (0: plain payload and flag stores, two workgroups per CU) or in the form the CDNA4 programming guide lists as
measured-valid (1: sc1 payload and flag stores, one workgroup per CU).  Results must not depend on it -- nor on where a
workgroup runs (/root/reference/chowdsp_fft.h:128-136: a transform is a pure function of its input): both forms against the
oracle and against each other, bit for bit, in GROUP and in fall-back mode; the forms that do not offer the listed hand-off
(complex 2^17 groups, the N = 65536 fused convolution) must then run their multi-launch composition and still be right.
The every-word stress under uneven load runs on both forms in tests/test_gpu_handoff_failure.py.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu


@pytest.fixture()
def listed(gpu):
    api = gpu[0]
    default_min = int(api.get_option("fused_min_batch"))
    assert api.get_option("group_handoff") == 0, "the fast form is the default"
    api.set_option("fused_min_batch", 1)
    yield api
    api.set_option("group_handoff", 0)
    api.set_option("group_force_solo", 0)
    api.set_option("fused_min_batch", default_min)
    api.stream_synchronize()


def _both_forms(gpu, s, x, direction):
    api, torch, dev = gpu
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = []
    for form in (0, 1):
        api.set_option("group_handoff", form)
        yd = torch.full_like(xd, -7.0)
        api.fft_transform_batch(s, xd, yd, direction)
        api.stream_synchronize()
        out.append(yd.cpu().numpy())
    api.set_option("group_handoff", 0)
    return out


@pytest.mark.parametrize("solo", [0, 1])
@pytest.mark.parametrize("n,tr,batch", [(65536, COMPLEX, 70), (131072, REAL, 40)])
def test_listed_form_matches_fast_form_and_oracle(gpu, oracle, listed, n, tr, batch, solo):
    api = listed
    api.set_option("group_force_solo", solo)
    s = api.fft_new_setup(n, api.FFT_COMPLEX if tr == COMPLEX else api.FFT_REAL)
    so = oracle.new_setup(n, tr)
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(batch * nf, 9100 + n % 97).reshape(batch, nf)
    pick = sorted({0, batch // 2, batch - 1})
    for d, od in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        fast, lst = _both_forms(gpu, s, x, d)
        assert np.array_equal(fast, lst), f"N={n} dir={d}: the two hand-off forms differ in {int((fast != lst).sum())} words"
        assert_frames_close(lst[pick], oracle.transform_batch(so, x[pick], od), what=f"N={n} listed hand-off, direction {d}, solo {solo}")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_forms_without_a_listed_variant_fall_back_to_several_launches(gpu, oracle, listed):
    """complex 2^17 (register-direct groups) and the N = 65536 fused convolution have no listed variant: with the option on they
    must take their multi-launch composition -- no in-kernel hand-off -- and give the oracle's result"""
    api, torch, dev = gpu
    api.set_option("group_handoff", 1)
    n = 1 << 17
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    x = ol.uniform(6 * 2 * n, 77).reshape(6, 2 * n)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD)
    api.stream_synchronize()
    assert_frames_close(yd.cpu().numpy(), oracle.transform_batch(so, x, FORWARD), what="c131072 with group_handoff = 1")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
    # fused convolution N = 65536: == the three-call sequence on the oracle
    n = 65536
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    x = ol.uniform(5 * 2 * n, 78).reshape(5, 2 * n)
    h = ol.uniform(2 * n, 79).reshape(1, 2 * n)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    yd = torch.empty_like(xd)
    api.fft_convolve_frames_batch(s, xd, H, yd, 1.0 / n)
    api.stream_synchronize()
    hb = oracle.transform(so, h[0], FORWARD, False)
    want = np.stack([oracle.transform(so, oracle.convolve(so, oracle.transform(so, x[i], FORWARD, False), hb, np.zeros(2 * n, np.float32), 1.0 / n), BACKWARD, False)
                     for i in range(5)])
    assert_frames_close(yd.cpu().numpy(), want, what="fused convolution N=65536 with group_handoff = 1 vs the oracle's call sequence")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_option_is_validated(gpu):
    api = gpu[0]
    with pytest.raises(api.FFTError):
        api.set_option("group_handoff", 2)
    assert api.get_option("group_handoff") == 0
