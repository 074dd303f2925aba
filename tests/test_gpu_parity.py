"""Parity of the HIP path against the oracle (reference SSE algorithm), through the C ABI.

Bound (conftest.PARITY_TOL): 1e-6 relative per frame -- relative L2 and max-abs / frame max-abs -- the
tolerance the north star states for fp32 ("results match the reference SSE path within 1e-6 relative").
Sizes the oracle finishes in seconds are compared frame by frame; BASELINE.json's full-size
configurations are checked through size-independent properties in test_gpu_fullsize.py.

The call sequences follow the reference's own tests (test/test.cpp): ordered forward vs the comparator,
backward scaled by 1/N, preallocated setups, in-place buffers, and the unordered-forward x2 -> convolve
-> unordered-backward (+ accumulate) convolution pipeline.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import PARITY_TOL, TONE_MAXABS_TOL, assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD, REAL

pytestmark = pytest.mark.gpu

SEED = 1234
COMPLEX_SIZES = [16, 32, 64, 128, 256, 512, 1024, 2048, 4096]
REAL_SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
TWO_PASS_SIZES = [8192, 16384, 32768, 65536, 131072, 524288]  # reference tests go to 2^19 (test/test.cpp:270-277)


def _frame(n, tr):
    return 2 * n if tr == COMPLEX else n


def _gpu_transform(gpu, s, x, direction):
    api, torch, dev = gpu
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, direction)
    torch.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("n", COMPLEX_SIZES)
@pytest.mark.parametrize("batch", [1, 3, 37, 130])  # 1, odd, > one workgroup's frames, ragged tail
def test_complex_batch_vs_oracle(gpu, oracle, n, batch):
    api = gpu[0]
    x = ol.uniform(batch * 2 * n, SEED + n + batch).reshape(batch, 2 * n)
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    assert s is not None
    want_f = oracle.transform_batch(so, x, FORWARD)
    got_f = _gpu_transform(gpu, s, x, api.FFT_FORWARD)
    assert_frames_close(got_f, want_f, what=f"c2c fwd N={n}")
    want_b = oracle.transform_batch(so, want_f, BACKWARD)
    got_b = _gpu_transform(gpu, s, want_f, api.FFT_BACKWARD)
    assert_frames_close(got_b, want_b, what=f"c2c bwd N={n}")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("n", REAL_SIZES)
@pytest.mark.parametrize("batch", [1, 5, 67])
def test_real_batch_vs_oracle(gpu, oracle, n, batch):
    api = gpu[0]
    x = ol.uniform(batch * n, SEED + 3 * n + batch).reshape(batch, n)
    s, so = api.fft_new_setup(n, api.FFT_REAL), oracle.new_setup(n, REAL)
    assert s is not None
    want_f = oracle.transform_batch(so, x, FORWARD)
    got_f = _gpu_transform(gpu, s, x, api.FFT_FORWARD)
    assert_frames_close(got_f, want_f, what=f"r2c N={n}")
    # packed layout details: bins 0 and 1 are the purely real DC and Nyquist terms
    assert np.allclose(got_f[:, 0], x.astype(np.float64).sum(axis=1), atol=1e-3)
    want_b = oracle.transform_batch(so, want_f, BACKWARD)
    got_b = _gpu_transform(gpu, s, want_f, api.FFT_BACKWARD)
    assert_frames_close(got_b, want_b, what=f"c2r N={n}")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("n", TWO_PASS_SIZES)
@pytest.mark.parametrize("tr", [COMPLEX, REAL])
def test_two_pass_sizes_vs_oracle(gpu, oracle, n, tr):
    api = gpu[0]
    batch = 3 if n <= 65536 else 2
    nf = _frame(n, tr)
    x = ol.uniform(batch * nf, SEED + n).reshape(batch, nf)
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    assert s is not None
    want_f = oracle.transform_batch(so, x, FORWARD)
    assert_frames_close(_gpu_transform(gpu, s, x, api.FFT_FORWARD), want_f, what=f"two-pass fwd N={n}")
    want_b = oracle.transform_batch(so, want_f, BACKWARD)
    assert_frames_close(_gpu_transform(gpu, s, want_f, api.FFT_BACKWARD), want_b, what=f"two-pass bwd N={n}")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_one_million_point_frames(gpu, oracle):
    """2^20 (the largest size of the power-of-two tile kernels): forward and backward, complex and real, two frames"""
    api = gpu[0]
    for tr in (COMPLEX, REAL):
        n = 1 << 20
        x = ol.uniform(2 * _frame(n, tr), 5).reshape(2, -1)
        s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
        want_f = oracle.transform_batch(so, x, FORWARD)
        assert_frames_close(_gpu_transform(gpu, s, x, api.FFT_FORWARD), want_f, what="2^20 fwd")
        assert_frames_close(_gpu_transform(gpu, s, want_f, api.FFT_BACKWARD), oracle.transform_batch(so, want_f, BACKWARD), what="2^20 bwd")
        api.fft_destroy_setup(s)
        oracle.destroy_setup(so)


@pytest.mark.parametrize("key_n", [16, 32, 64, 1024, 4096, 96, 480])
@pytest.mark.parametrize("tag", ["c", "r"])
@pytest.mark.parametrize("name", ["tone", "uniform", "impulse"])
def test_golden_vectors(gpu, golden, key_n, tag, name):
    """committed outputs of the reference's SSE path (tests/golden/make_golden.py)"""
    api = gpu[0]
    tr = api.FFT_COMPLEX if tag == "c" else api.FFT_REAL
    if f"{tag}{key_n}_{name}_fwd" not in golden:
        assert api.fft_new_setup(key_n, tr) is None  # real N=16: rejected, like the reference
        return
    cplx = tag == "c"
    if name == "tone":
        x = golden[f"{tag}{key_n}_tone_in"]
    elif name == "uniform":
        x = ol.uniform(2 * key_n if cplx else key_n, SEED)
    else:
        x = ol.impulse(key_n, cplx, 1)
    s = api.fft_new_setup(key_n, tr)
    f = _gpu_transform(gpu, s, x[None, :], api.FFT_FORWARD)[0]
    mt = TONE_MAXABS_TOL if name == "tone" else None  # single-tone input: the named element-wise exception (conftest.py)
    assert_frames_close(f, golden[f"{tag}{key_n}_{name}_fwd"], what=f"golden fwd {tag}{key_n} {name}", maxabs_tol=mt)
    b = _gpu_transform(gpu, s, golden[f"{tag}{key_n}_{name}_fwd"][None, :], api.FFT_BACKWARD)[0]
    assert_frames_close(b, golden[f"{tag}{key_n}_{name}_bwd"], what=f"golden bwd {tag}{key_n} {name}", maxabs_tol=mt)
    api.fft_destroy_setup(s)


def test_golden_65536_every_64th_bin(gpu, golden):
    api = gpu[0]
    s = api.fft_new_setup(65536, api.FFT_COMPLEX)
    f = _gpu_transform(gpu, s, ol.uniform(2 * 65536, SEED)[None, :], api.FFT_FORWARD)[0]
    got = f.reshape(-1, 2)[::64].ravel()
    want = golden["c65536_uniform_fwd_every64"]
    # sub-sampled bins: compare against the sub-sample's own norm
    assert ol.rel_l2(got, want) <= PARITY_TOL
    api.fft_destroy_setup(s)


# ---- legacy single-frame interface, host pointers: the reference's own test sequences -------------------
@pytest.mark.parametrize("n", [32, 64, 256, 1024, 4096, 16384, 65536])
@pytest.mark.parametrize("tr", [COMPLEX, REAL])
@pytest.mark.parametrize("preallocate", [False, True])
def test_reference_test_sequence_host_pointers(gpu, oracle, n, tr, preallocate):
    """reference test/test.cpp:16-73 (complex) / :75-128 (real): in-place ordered forward, backward * 1/N,
    plain and preallocated setups, fft_simd_width_bytes == 16; comparator = the SSE-path oracle; the
    reference's own tolerance (2e-7 * number of floats, absolute, :9-14) is checked besides the parity bound"""
    api = gpu[0]
    nf = _frame(n, tr)
    data, dp = api.aligned_array(nf)
    work, wp = api.aligned_array(nf)
    data[:] = ol.tone(n, tr == COMPLEX)
    x0 = data.copy()
    prealloc = None
    if preallocate:
        nbytes = api.fft_bytes_required(n, tr, False)
        prealloc = api.aligned_malloc(nbytes)
        s = api.fft_new_setup_preallocated(n, tr, prealloc, False)
    else:
        s = api.fft_new_setup(n, tr, False)
    assert s is not None
    assert api.fft_simd_width_bytes(s) == 16
    so = oracle.new_setup(n, tr)

    api.fft_transform(s, data, data, work, api.FFT_FORWARD)
    want_f = oracle.transform(so, x0, FORWARD)
    assert np.abs(data - want_f).max() <= 2.0e-7 * nf
    assert_frames_close(data, want_f, what="legacy fwd", maxabs_tol=TONE_MAXABS_TOL)  # tone input
    api.fft_transform(s, data, data, work, api.FFT_BACKWARD)
    want_b = oracle.transform(so, want_f, BACKWARD)
    assert np.abs(data / n - want_b / n).max() <= 2.0e-7 * nf
    assert_frames_close(data, want_b, what="legacy bwd", maxabs_tol=TONE_MAXABS_TOL)
    assert np.abs(data / n - x0).max() <= 2e-6  # BACKWARD(FORWARD(x)) = N x

    if preallocate:
        api.aligned_free(prealloc)  # caller-owned; fft_destroy_setup must not be called (reference chowdsp_fft.h:110-112)
    else:
        api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
    api.aligned_free(dp)
    api.aligned_free(wp)


@pytest.mark.parametrize("n", [32, 64, 1024, 4096, 32768])
@pytest.mark.parametrize("tr", [COMPLEX, REAL])
def test_reference_convolution_sequence_host_pointers(gpu, oracle, n, tr):
    """reference test/test.cpp:131-180 / :182-233: unordered fwd x2 -> fft_convolve_unordered(1/N) -> unordered
    bwd (-> fft_accumulate with aliasing for REAL), against the same pipeline run by the oracle"""
    api = gpu[0]
    nf = _frame(n, tr)
    cplx = tr == COMPLEX
    s1, p1 = api.aligned_array(nf)
    s2, p2 = api.aligned_array(nf)
    out, po = api.aligned_array(nf)
    work, pw = api.aligned_array(nf)
    s1[:] = ol.tone(n, cplx, 100.0)
    s2[:] = ol.tone(n, cplx, 200.0)
    x1, x2 = s1.copy(), s2.copy()
    out[:] = 0.0
    norm = 1.0 / n
    s, so = api.fft_new_setup(n, tr, False), oracle.new_setup(n, tr)

    api.fft_transform_unordered(s, s1, s1, work, api.FFT_FORWARD)
    api.fft_transform_unordered(s, s2, s2, work, api.FFT_FORWARD)
    api.fft_convolve_unordered(s, s1, s2, out, norm)
    api.fft_transform_unordered(s, out, out, work, api.FFT_BACKWARD)

    a = oracle.transform(so, x1, FORWARD, False)
    b = oracle.transform(so, x2, FORWARD, False)
    ab = oracle.convolve(so, a, b, np.zeros(nf, np.float32), norm)
    want = oracle.transform(so, ab, BACKWARD, False)
    if not cplx:
        add = ol.uniform(nf, 9)
        addbuf, pa = api.aligned_array(nf)
        addbuf[:] = add
        api.fft_accumulate(s, out, addbuf, out, n)  # ab aliases a, as in the reference test (:218)
        want = oracle.accumulate(want, add)
        api.aligned_free(pa)
    assert np.abs(out - want).max() <= 2.0e-7 * nf  # the reference's tolerance
    assert_frames_close(out, want, what="convolution pipeline", maxabs_tol=TONE_MAXABS_TOL)  # tone inputs
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
    for p in (p1, p2, po, pw):
        api.aligned_free(p)


@pytest.mark.parametrize("n", [64, 4096])
@pytest.mark.parametrize("tag", ["c", "r"])
def test_convolution_golden(gpu, golden, n, tag):
    """same pipeline against the committed output of the reference itself"""
    api, torch, dev = gpu
    tr = api.FFT_COMPLEX if tag == "c" else api.FFT_REAL
    s = api.fft_new_setup(n, tr)
    for suffix, (xa, xb) in {
        "conv": (golden[f"{tag}{n}_conv_in_a"], golden[f"{tag}{n}_conv_in_b"]),
        "conv_uniform": (ol.uniform(golden[f"{tag}{n}_conv"].size, SEED), ol.uniform(golden[f"{tag}{n}_conv"].size, SEED + 1)),
    }.items():
        a, b = torch.from_numpy(xa.copy()).to(dev), torch.from_numpy(xb.copy()).to(dev)
        A, B = torch.empty_like(a), torch.empty_like(b)
        api.fft_transform_unordered_batch(s, a, A, api.FFT_FORWARD)
        api.fft_transform_unordered_batch(s, b, B, api.FFT_FORWARD)
        acc = torch.zeros_like(a)
        api.fft_convolve_unordered_batch(s, A, B, acc, 1.0 / n)
        y = torch.empty_like(a)
        api.fft_transform_unordered_batch(s, acc, y, api.FFT_BACKWARD)
        fused = torch.empty_like(a)
        api.fft_convolve_frames_batch(s, a, B, fused, 1.0 / n)
        torch.cuda.synchronize()
        want = golden[f"{tag}{n}_{suffix}"]
        assert_frames_close(y.cpu().numpy(), want, what=f"unfused {suffix}")
        assert_frames_close(fused.cpu().numpy(), want, what=f"fused {suffix}")
    api.fft_destroy_setup(s)


# sizes with a fused convolution kernel (incl. the 128-thread-workgroup 2048-point plan and the pipelined variants), and
# sizes where fft_convolve_frames_batch runs the three steps as separate launches (8192-point plan, two-pass, mixed radix)
@pytest.mark.parametrize("n,tr", [(16, COMPLEX), (1024, COMPLEX), (2048, COMPLEX), (4096, COMPLEX), (32, REAL), (1024, REAL),
                                  (2048, REAL), (4096, REAL), (8192, REAL),
                                  (8192, COMPLEX), (65536, COMPLEX), (16384, REAL), (32768, REAL), (480, COMPLEX), (960, REAL),
                                  # fused on the split-plane kernels (frame in registers): one launch
                                  (16384, COMPLEX), (32768, COMPLEX), (65536, REAL),
                                  # two-pass register-direct tiles: the product rides on the forward transform's pass-B stores
                                  (131072, COMPLEX), (262144, COMPLEX), (131072, REAL), (524288, REAL)])
def test_batched_convolution_ops_vs_oracle(gpu, oracle, n, tr):
    """fft_convolve_unordered_batch (per-frame and shared filter), accumulate into a non-zero ab, fused kernel"""
    api, torch, dev = gpu
    nf, batch = _frame(n, tr), 9
    x = ol.uniform(batch * nf, 21 + n).reshape(batch, nf)
    h = ol.uniform(batch * nf, 22 + n).reshape(batch, nf)
    acc0 = ol.uniform(batch * nf, 23 + n).reshape(batch, nf)
    s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    X, H = torch.empty_like(xd), torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, xd, X, api.FFT_FORWARD)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    # ab += a*b*scaling with a NON-zero ab and a per-frame b.  The unordered layouts of the two
    # implementations differ, so the initial ab is the (unordered) spectrum of a time-domain frame acc0 in
    # each implementation's own layout; the time-domain results must agree.
    y = torch.empty_like(xd)
    want = np.empty_like(x)
    for f in range(batch):
        a = oracle.transform(so, x[f], FORWARD, False)
        b = oracle.transform(so, h[f], FORWARD, False)
        ab = oracle.transform(so, acc0[f], FORWARD, False)
        want[f] = oracle.transform(so, oracle.convolve(so, a, b, ab, 0.5), BACKWARD, False)
    acc_gpu = torch.from_numpy(acc0).to(dev)
    A0 = torch.empty_like(acc_gpu)
    api.fft_transform_unordered_batch(s, acc_gpu, A0, api.FFT_FORWARD)
    api.fft_convolve_unordered_batch(s, X, H, A0, 0.5)
    api.fft_transform_unordered_batch(s, A0, y, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    assert_frames_close(y.cpu().numpy(), want, what="per-frame multiply-accumulate")

    # shared filter + fused kernel agree with each other and with the oracle
    acc = torch.zeros_like(xd)
    api.fft_convolve_unordered_batch(s, X, H[:1].contiguous(), acc, 1.0 / n, b_is_shared=True)
    api.fft_transform_unordered_batch(s, acc, y, api.FFT_BACKWARD)
    fused = torch.empty_like(xd)
    api.fft_convolve_frames_batch(s, xd, H[:1].contiguous(), fused, 1.0 / n)
    torch.cuda.synchronize()
    b0 = oracle.transform(so, h[0], FORWARD, False)
    for f in range(batch):
        a = oracle.transform(so, x[f], FORWARD, False)
        want[f] = oracle.transform(so, oracle.convolve(so, a, b0, np.zeros(nf, np.float32), 1.0 / n), BACKWARD, False)
    assert_frames_close(y.cpu().numpy(), want, what="shared-filter convolution")
    assert_frames_close(fused.cpu().numpy(), want, what="fused convolution")
    inplace = xd.clone()
    api.fft_convolve_frames_batch(s, inplace, H[:1].contiguous(), inplace, 1.0 / n)  # x == y
    torch.cuda.synchronize()
    assert_frames_close(inplace.cpu().numpy(), want, what="fused convolution in place")
    # accumulate
    out = torch.empty_like(xd)
    api.fft_accumulate_batch(s, xd, hd, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), x + h)  # one fp32 add per element: bit-exact
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_in_place_and_device_pointer_legacy_calls(gpu, oracle):
    """input and output may alias (reference chowdsp_fft.h:136); legacy symbols also accept device pointers"""
    api, torch, dev = gpu
    for n, tr in ((4096, COMPLEX), (1024, REAL), (65536, COMPLEX)):
        nf = _frame(n, tr)
        x = ol.uniform(4 * nf, 31).reshape(4, nf)
        s, so = api.fft_new_setup(n, tr), oracle.new_setup(n, tr)
        want = oracle.transform_batch(so, x, FORWARD)
        xd = torch.from_numpy(x).to(dev)
        api.fft_transform_batch(s, xd, xd, api.FFT_FORWARD)  # in place, batched
        torch.cuda.synchronize()
        assert_frames_close(xd.cpu().numpy(), want, what="in-place batch")
        one = torch.from_numpy(x[0].copy()).to(dev)
        api.fft_transform(s, one, one, None, api.FFT_FORWARD)  # legacy symbol, device pointers, work = NULL
        assert_frames_close(one.cpu().numpy(), want[0], what="legacy device pointer")
        api.fft_destroy_setup(s)
        oracle.destroy_setup(so)


def test_edge_cases(gpu):
    api, torch, dev = gpu
    # sizes the reference rejects, or that are too long for this backend -> NULL, no crash
    for n, tr in ((0, COMPLEX), (-64, COMPLEX), (8, COMPLEX), (16, REAL), (24, COMPLEX), (112, COMPLEX), (48, REAL),
                  (1 << 28, COMPLEX), (3 << 27, COMPLEX), (1 << 26, REAL), ((1 << 22) + 16, COMPLEX)):
        assert api.fft_new_setup(n, tr) is None
    s = api.fft_new_setup(256, api.FFT_COMPLEX)
    x = torch.zeros(512, device=dev)
    api.fft_transform_batch(s, x, x, api.FFT_FORWARD, batch=0)  # empty batch is a no-op
    with pytest.raises(ValueError):
        api.fft_transform_batch(s, torch.zeros(100, device=dev), x, api.FFT_FORWARD)  # ragged buffer
    # linearity / zero input / impulse -> all ones
    api.fft_transform_batch(s, x, x, api.FFT_FORWARD)
    torch.cuda.synchronize()
    assert float(x.abs().max()) == 0.0
    imp = torch.zeros(512, device=dev)
    imp[0] = 1.0
    y = torch.empty_like(imp)
    api.fft_transform_batch(s, imp, y, api.FFT_FORWARD)
    torch.cuda.synchronize()
    yc = y.cpu().numpy()
    assert np.array_equal(yc[0::2], np.ones(256, np.float32)) and np.array_equal(yc[1::2], np.zeros(256, np.float32))
    api.fft_destroy_setup(s)


def test_two_streams_share_one_setup(gpu, oracle):
    """the setup is immutable and shareable (reference chowdsp_fft.h:87-91): two streams, one plan"""
    api, torch, dev = gpu
    n = 16384  # two-pass size: exercises the per-stream workspace
    x = ol.uniform(8 * 2 * n, 41).reshape(8, 2 * n)
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    want = oracle.transform_batch(so, x, FORWARD)
    xd = torch.from_numpy(x).to(dev)
    y1, y2 = torch.empty_like(xd), torch.empty_like(xd)
    st1, st2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        api.fft_transform_batch(s, xd, y1, api.FFT_FORWARD, stream=st1)
        api.fft_transform_batch(s, xd, y2, api.FFT_FORWARD, stream=st2)
    torch.cuda.synchronize()
    assert_frames_close(y1.cpu().numpy(), want, what="stream 1")
    assert_frames_close(y2.cpu().numpy(), want, what="stream 2")
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_wave_per_frame_form_of_n4096(gpu, oracle):
    """option "wave4096": one wavefront per 4096-point frame (csrc/wave_kernels.hpp) -- same results as the workgroup plan"""
    if gpu[0].get_option("experiments") != 1:
        pytest.skip("a superseded form: only libraries built with EXPERIMENTS=1 contain it")
    api, torch, dev = gpu
    n, batch = 4096, 37
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX), oracle.new_setup(n, COMPLEX)
    x = ol.uniform(batch * 2 * n, 4242).reshape(batch, 2 * n)
    want_f = oracle.transform_batch(so, x, FORWARD)
    want_b = oracle.transform_batch(so, want_f, BACKWARD)
    api.set_option("wave4096", 1)
    try:
        assert_frames_close(_gpu_transform(gpu, s, x, api.FFT_FORWARD), want_f, what="wave-per-frame fwd")
        assert_frames_close(_gpu_transform(gpu, s, want_f, api.FFT_BACKWARD), want_b, what="wave-per-frame bwd")
        # strided frames take the same kernel
        stride = 2 * n + 32
        buf = torch.zeros((batch, stride), device=dev)
        buf[:, :2 * n] = torch.from_numpy(x).to(dev)
        out = torch.full((batch, stride), 9.0, device=dev)
        api.fft_transform_batch_strided(s, buf, stride, out, stride, batch, api.FFT_FORWARD)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert_frames_close(o[:, :2 * n], want_f, what="wave-per-frame strided")
        assert np.all(o[:, 2 * n:] == 9.0)
    finally:
        api.set_option("wave4096", 0)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("tr", [COMPLEX, REAL])
def test_split_plane_form_of_n8192(gpu, oracle, tr):
    """option "split8192": the 8192-point core (N = 8192 complex, real N = 16384) through the 256-work-item split-plane
    kernels (csrc/split_plane_kernels.hpp, 32 x 32 x 8) -- same results as the interleaved-image plan; the fused
    convolution of these sizes always runs on them"""
    if gpu[0].get_option("experiments") != 1:
        pytest.skip("a superseded form: only libraries built with EXPERIMENTS=1 contain it")
    api, torch, dev = gpu
    n = 8192 if tr == COMPLEX else 16384
    nf, batch = _frame(n, tr), 11
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX if tr == COMPLEX else api.FFT_REAL), oracle.new_setup(n, tr)
    x = ol.uniform(batch * nf, 777 + n).reshape(batch, nf)
    want_f = oracle.transform_batch(so, x, FORWARD)
    want_b = oracle.transform_batch(so, want_f, BACKWARD)
    api.set_option("split8192", 1)
    try:
        assert_frames_close(_gpu_transform(gpu, s, x, api.FFT_FORWARD), want_f, what="split-plane 8192 fwd")
        assert_frames_close(_gpu_transform(gpu, s, want_f, api.FFT_BACKWARD), want_b, what="split-plane 8192 bwd")
        stride = nf + 64
        buf = torch.zeros((batch, stride), device=dev)
        buf[:, :nf] = torch.from_numpy(x).to(dev)
        out = torch.full((batch, stride), 9.0, device=dev)
        api.fft_transform_batch_strided(s, buf, stride, out, stride, batch, api.FFT_FORWARD)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert_frames_close(o[:, :nf], want_f, what="split-plane 8192 strided")
        assert np.all(o[:, nf:] == 9.0)
    finally:
        api.set_option("split8192", 0)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


@pytest.mark.parametrize("n,tr", [(4096, REAL), (2048, COMPLEX)])
def test_wave_per_frame_fused_convolution(gpu, oracle, n, tr):
    """option "conv_wave2048": fft_convolve_frames_batch of 2048-point cores by one wavefront per frame
    (csrc/split_plane_kernels.hpp, SplitGeom<64>) -- same results as the workgroup plan and the oracle's pipeline"""
    if gpu[0].get_option("experiments") != 1:
        pytest.skip("a superseded form: only libraries built with EXPERIMENTS=1 contain it")
    api, torch, dev = gpu
    nf, batch = _frame(n, tr), 37
    x = ol.uniform(batch * nf, 91 + n).reshape(batch, nf)
    h = ol.uniform(nf, 92 + n).reshape(1, nf)
    s, so = api.fft_new_setup(n, api.FFT_COMPLEX if tr == COMPLEX else api.FFT_REAL), oracle.new_setup(n, tr)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    H = torch.empty_like(hd)
    api.fft_transform_unordered_batch(s, hd, H, api.FFT_FORWARD)
    b0 = oracle.transform(so, h[0], FORWARD, False)
    want = np.empty_like(x)
    for f in range(batch):
        a = oracle.transform(so, x[f], FORWARD, False)
        want[f] = oracle.transform(so, oracle.convolve(so, a, b0, np.zeros(nf, np.float32), 1.0 / n), BACKWARD, False)
    api.set_option("conv_wave2048", 1)
    try:
        y = torch.empty_like(xd)
        api.fft_convolve_frames_batch(s, xd, H, y, 1.0 / n)
        inplace = xd.clone()
        api.fft_convolve_frames_batch(s, inplace, H, inplace, 1.0 / n)
        torch.cuda.synchronize()
        assert_frames_close(y.cpu().numpy(), want, what="wave-per-frame fused convolution")
        assert_frames_close(inplace.cpu().numpy(), want, what="wave-per-frame fused convolution in place")
    finally:
        api.set_option("conv_wave2048", 0)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)


def test_misaligned_device_buffers_are_refused(gpu):
    """16-byte alignment is the reference's contract for its SIMD buffers (sse.cpp:1527-1528); a misaligned device address
    would be a memory fault inside kernels that move frames with 16-byte accesses, so the batched calls return EINVAL"""
    api, torch, dev = gpu
    n, batch = 96, 8
    s = api.fft_new_setup(n, api.FFT_COMPLEX)
    buf = torch.zeros(batch * 2 * n + 8, device=dev)
    out = torch.empty(batch * 2 * n + 8, device=dev)
    x = buf[2:2 + batch * 2 * n]  # 8 bytes off
    with pytest.raises(api.FFTError):
        api.fft_transform_batch(s, x, out[:batch * 2 * n], api.FFT_FORWARD)
    with pytest.raises(api.FFTError):
        api.fft_transform_batch(s, buf[:batch * 2 * n], out[2:2 + batch * 2 * n], api.FFT_FORWARD)
    api.fft_transform_batch(s, buf[4:4 + batch * 2 * n], out[:batch * 2 * n], api.FFT_FORWARD)  # 16 bytes off: fine
    torch.cuda.synchronize()
    api.fft_destroy_setup(s)
