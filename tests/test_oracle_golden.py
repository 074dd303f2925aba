"""The oracle (plain-C restatement, oracle/pffft_restate.c) against the committed golden vectors
(outputs of the reference's own SSE path, tests/golden/make_golden.py) and against float64 numpy.

Bit-identity with the reference is asserted in test_oracle_vs_reference.py (needs oracle/_ref).  Here the
bound is 1e-7 relative L2 per frame -- below fp32 round-off of a length-4096 transform (~1.7e-7 vs
float64), so an algorithmic deviation cannot hide, while a last-bit libm cosf/sinf difference between
hosts (twiddle tables) does not fail the suite on another machine.
"""
import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

GOLDEN_TOL = 1.0e-7
SEED = 1234


def _input(golden, tag, n, name):
    cplx = tag == "c"
    if name == "tone":
        return golden[f"{tag}{n}_tone_in"]
    if name == "uniform":
        return ol.uniform(2 * n if cplx else n, SEED)
    return ol.impulse(n, cplx, 1)


@pytest.mark.parametrize("n", [16, 32, 64, 1024, 4096, 96, 480])
@pytest.mark.parametrize("tag", ["c", "r"])
@pytest.mark.parametrize("name", ["tone", "uniform", "impulse"])
def test_oracle_matches_golden_ordered(oracle, golden, n, tag, name):
    tr = COMPLEX if tag == "c" else REAL
    s = oracle.new_setup(n, tr)
    if f"{tag}{n}_{name}_fwd" not in golden:
        assert s is None  # real N=16 is rejected by the reference too
        return
    assert s is not None
    x = _input(golden, tag, n, name)
    f = oracle.transform(s, x, FORWARD, True)
    assert ol.rel_l2(f, golden[f"{tag}{n}_{name}_fwd"]) <= GOLDEN_TOL
    b = oracle.transform(s, golden[f"{tag}{n}_{name}_fwd"], BACKWARD, True)
    assert ol.rel_l2(b, golden[f"{tag}{n}_{name}_bwd"]) <= GOLDEN_TOL
    oracle.destroy_setup(s)


@pytest.mark.parametrize("tag,n", [("c", 64), ("r", 64)])
def test_oracle_unordered_layout_matches_golden(oracle, golden, tag, n):
    tr = COMPLEX if tag == "c" else REAL
    s = oracle.new_setup(n, tr)
    x = ol.uniform(2 * n if tag == "c" else n, SEED)
    u = oracle.transform(s, x, FORWARD, False)
    assert ol.rel_l2(u, golden[f"{tag}{n}_uniform_fwd_unordered_sse"]) <= GOLDEN_TOL
    # the unordered layout is a permutation of the ordered one (complex case: same multiset of values)
    if tag == "c":
        o = oracle.transform(s, x, FORWARD, True)
        assert np.allclose(np.sort(u), np.sort(o), rtol=0, atol=0)
    oracle.destroy_setup(s)


@pytest.mark.parametrize("n", [64, 4096])
@pytest.mark.parametrize("tag", ["c", "r"])
def test_oracle_convolution_pipeline_matches_golden(oracle, golden, n, tag):
    """reference test/test.cpp:161-166, :212-215: unordered fwd x2 -> convolve(1/N) -> unordered bwd"""
    tr = COMPLEX if tag == "c" else REAL
    s = oracle.new_setup(n, tr)
    for suffix, (xa, xb) in {
        "conv": (golden[f"{tag}{n}_conv_in_a"], golden[f"{tag}{n}_conv_in_b"]),
        "conv_uniform": (ol.uniform(golden[f"{tag}{n}_conv"].size, SEED), ol.uniform(golden[f"{tag}{n}_conv"].size, SEED + 1)),
    }.items():
        a = oracle.transform(s, xa, FORWARD, False)
        b = oracle.transform(s, xb, FORWARD, False)
        ab = oracle.convolve(s, a, b, np.zeros_like(a), 1.0 / n)
        y = oracle.transform(s, ab, BACKWARD, False)
        assert ol.rel_l2(y, golden[f"{tag}{n}_{suffix}"]) <= GOLDEN_TOL
    oracle.destroy_setup(s)


def test_oracle_65536_every_64th_bin(oracle, golden):
    s = oracle.new_setup(65536, COMPLEX)
    f = oracle.transform(s, ol.uniform(2 * 65536, SEED), FORWARD, True)
    assert ol.rel_l2(f.reshape(-1, 2)[::64].ravel(), golden["c65536_uniform_fwd_every64"]) <= GOLDEN_TOL
    oracle.destroy_setup(s)


@pytest.mark.parametrize("n", [16, 64, 256, 2048, 8192])
def test_oracle_complex_vs_float64_dft(oracle, n):
    """independent check: float64 numpy DFT; bound 5e-7 (fp32 round-off of the 4-lane radix-4 algorithm)"""
    s = oracle.new_setup(n, COMPLEX)
    x = ol.uniform(2 * n, n)
    f = oracle.transform(s, x, FORWARD, True).astype(np.float64)
    ref = np.fft.fft(x[0::2].astype(np.float64) + 1j * x[1::2])
    assert ol.rel_l2(f[0::2] + 1j * f[1::2], ref) <= 5e-7
    b = oracle.transform(s, f.astype(np.float32), BACKWARD, True)
    assert np.abs(b / n - x).max() <= 2e-6  # BACKWARD(FORWARD(x)) = N x   (reference chowdsp_fft.h:128-129)
    oracle.destroy_setup(s)


@pytest.mark.parametrize("n", [32, 64, 256, 2048, 8192])
def test_oracle_real_packed_format_vs_float64(oracle, n):
    """packed layout [X0, X(N/2), Re X1, Im X1, ...] (reference chowdsp_fft_juce.cpp:58-61,74)"""
    s = oracle.new_setup(n, REAL)
    x = ol.uniform(n, n + 7)
    f = oracle.transform(s, x, FORWARD, True).astype(np.float64)
    spec = np.fft.rfft(x.astype(np.float64))
    packed = np.empty(n)
    packed[0], packed[1] = spec[0].real, spec[n // 2].real
    packed[2::2], packed[3::2] = spec[1:n // 2].real, spec[1:n // 2].imag
    assert ol.rel_l2(f, packed) <= 5e-7
    b = oracle.transform(s, f.astype(np.float32), BACKWARD, True)
    assert np.abs(b / n - x).max() <= 2e-6
    oracle.destroy_setup(s)


def test_oracle_size_rules(oracle):
    """reference simd/chowdsp_fft_impl_common.hpp:168-177, 216-225: complex N % 16, real N % 32, N/4 = 2^a 3^b 5^c"""
    assert oracle.new_setup(8, COMPLEX) is None
    assert oracle.new_setup(16, REAL) is None
    assert oracle.new_setup(0, COMPLEX) is None
    assert oracle.new_setup(112, COMPLEX) is None  # 7 is not an accepted factor
    assert oracle.new_setup(48, REAL) is None
    for n, tr in ((16, COMPLEX), (32, REAL), (96, COMPLEX), (96, REAL), (480, COMPLEX), (9216, REAL)):
        s = oracle.new_setup(n, tr)
        assert s is not None
        oracle.destroy_setup(s)


def test_oracle_plan_matches_reference_pass_table(oracle):
    """SURVEY.md section 8 pass table (measured on the reference): radix-4 passes + one leading radix-2 if needed"""
    expect = {(1024, COMPLEX): [4, 4, 4, 4], (4096, COMPLEX): [4] * 5, (65536, COMPLEX): [4] * 7,
              (1024, REAL): [4, 4, 4, 4], (4096, REAL): [4] * 5, (2048, COMPLEX): [2, 4, 4, 4, 4]}
    for (n, tr), passes in expect.items():
        s = oracle.new_setup(n, tr)
        assert oracle.passes(s) == passes
        oracle.destroy_setup(s)


def test_oracle_accumulate_and_aliasing(oracle):
    a, b = ol.uniform(256, 1), ol.uniform(256, 2)
    assert np.array_equal(oracle.accumulate(a, b), a + b)
    # in-place transform (input and output may alias, reference chowdsp_fft.h:136)
    import ctypes as C

    s = oracle.new_setup(256, COMPLEX)
    x = ol.aligned_copy(ol.uniform(512, 3))
    want = oracle.transform(s, x, FORWARD, True)
    oracle.L.oracle_transform(s, x.ctypes.data, x.ctypes.data, FORWARD, 1)
    assert np.array_equal(x, want)
    oracle.destroy_setup(s)
    del C
