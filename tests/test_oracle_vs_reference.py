"""Pins the oracle restatement against the REAL reference compiled in place (oracle/_ref, built by
`make -C oracle ref` from /root/reference; skipped where that .so does not exist).

The restatement keeps the SSE path's operation order and is compiled with -ffp-contract=off, so the
assertion is BIT identity with the reference's SSE path (use_avx_if_available = false), for every
power-of-two size the reference's own tests cover (test/test.cpp:270-277: 2^5 .. 2^19; here also 2^4).
The AVX path (different radix-8 finalize, FMA) is only required to agree within the parity bound.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import PARITY_TOL
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

SIZES_FAST = [1 << k for k in range(4, 15)]
SIZES_SLOW = [1 << k for k in range(15, 20)]
# the reference's "Other sizes" (test/test.cpp:279-285) and more 2^a 3^b 5^c sizes: radix-3 / radix-5 passes
SIZES_OTHER = [96, 192, 384, 480, 640, 768, 9216, 48, 80, 160, 240, 320, 720, 1440, 2000, 3072, 5120, 15360, 24000]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n", SIZES_FAST + SIZES_SLOW + SIZES_OTHER)
@pytest.mark.parametrize("tr", [COMPLEX, REAL])
def test_bit_identical_to_reference_sse(oracle, reflib, n, tr):
    so, sr = oracle.new_setup(n, tr), reflib.new_setup(n, tr, False)
    assert (so is None) == (sr is None)
    if so is None:
        return
    assert reflib.simd_width(sr) == 16
    nf = 2 * n if tr == COMPLEX else n
    x = ol.uniform(nf, 4321 + n)
    for ordered in (True, False):
        fr = reflib.transform(sr, x, FORWARD, ordered)
        fo = oracle.transform(so, x, FORWARD, ordered)
        assert np.array_equal(_bits(fo), _bits(fr)), f"forward ordered={ordered}"
        br = reflib.transform(sr, fr, BACKWARD, ordered)
        bo = oracle.transform(so, fr, BACKWARD, ordered)
        assert np.array_equal(_bits(bo), _bits(br)), f"backward ordered={ordered}"
    a = reflib.transform(sr, x, FORWARD, False)
    b = reflib.transform(sr, ol.uniform(nf, 99), FORWARD, False)
    ab0 = ol.uniform(nf, 5)
    assert np.array_equal(_bits(oracle.convolve(so, a, b, ab0, 1.0 / n)), _bits(reflib.convolve(sr, a, b, ab0, 1.0 / n)))
    assert np.array_equal(_bits(oracle.accumulate(a, b)), _bits(reflib.accumulate(sr, a, b)))
    oracle.destroy_setup(so)
    reflib.destroy_setup(sr)


@pytest.mark.parametrize("n", [64, 1024, 4096, 65536])
@pytest.mark.parametrize("tr", [COMPLEX, REAL])
def test_reference_tone_inputs(oracle, reflib, n, tr):
    """the reference tests' own input (test/test.cpp:23-27, :82-85)"""
    so, sr = oracle.new_setup(n, tr), reflib.new_setup(n, tr, False)
    x = ol.tone(n, tr == COMPLEX)
    assert np.array_equal(_bits(oracle.transform(so, x, FORWARD)), _bits(reflib.transform(sr, x, FORWARD)))
    oracle.destroy_setup(so)
    reflib.destroy_setup(sr)


@pytest.mark.parametrize("n", [128, 1024, 4096, 65536])
@pytest.mark.parametrize("tr", [COMPLEX, REAL])
def test_reference_avx_path_within_parity_bound(oracle, reflib, n, tr):
    """AVX (V = 8) vs the SSE-path oracle: SURVEY.md section 6 measured 1.1e-7 .. 2.5e-7; bound 1e-6"""
    sa = reflib.new_setup(n, tr, True)
    if sa is None or reflib.simd_width(sa) != 32:
        pytest.skip("host CPU without AVX2+FMA (or N below the AVX minimum)")
    so = oracle.new_setup(n, tr)
    x = ol.uniform(2 * n if tr == COMPLEX else n, 77)
    assert ol.rel_l2(reflib.transform(sa, x, FORWARD), oracle.transform(so, x, FORWARD)) <= PARITY_TOL
    oracle.destroy_setup(so)
    reflib.destroy_setup(sa)


def test_reference_setup_quirks(reflib):
    """sizes the reference rejects (common.hpp:168-177) come back as a TAGGED null (chowdsp_fft.cpp:271-273)"""
    assert reflib.new_setup(16, REAL, False) is None
    assert reflib.new_setup(8, COMPLEX, False) is None
    # fft_bytes_required = 2*Ncvec*V*4 + sizeof(FFT_Setup) (sse.cpp:67-72): 8N+96 complex, 4N+96 real
    assert reflib.bytes_required(4096, COMPLEX, False) == 8 * 4096 + 96
    assert reflib.bytes_required(1024, REAL, False) == 4 * 1024 + 96
