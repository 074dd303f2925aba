"""pytest configuration: the `gpu` marker, the CPU checkers (oracle / compiled reference) and golden data.

  -m "not gpu" : oracle vs golden vectors / compiled reference, host logic, C-ABI surface (no GPU needed)
  -m gpu       : parity of the HIP path (through the C ABI) against the oracle -- needs one MI355X
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

# the developer options of the library ("group_spin_limit": hand-off waits that DO run out, for the failure-path tests) are locked
# unless this is set before the library reads it (capi.cpp developer_options)
os.environ.setdefault("CHOWDSP_FFT_DEVELOPER", "1")

import oracle_lib  # noqa: E402

# parity bound of the north star: results match the reference SSE path within 1e-6 relative fp32,
# normalised per frame: relative L2 error <= 1e-6.
PARITY_TOL = 1.0e-6
# Secondary, element-wise check: max-abs error over the frame's max-abs value, same bound (1e-6) by default.
# Named exception, passed explicitly (maxabs_tol=TONE_MAXABS_TOL) by the tests it applies to: on SINGLE-TONE inputs
# the reference algorithm is itself up to 8e-7 away from the exact (float64) result in this metric (its twiddle
# angles are formed in float: oracle vs float64, N=262144 tone, BACKWARD: 8.2e-7; N=4096: 5.6e-7), so two correct
# fp32 implementations can differ by more than 1e-6 element-wise while each is within 1e-6 of the truth.  Those tests
# use 2e-6 (1e-6 for each side); their relative-L2 bound stays 1e-6.
# Second named exception (maxabs_tol=MIXED_RADIX_MAXABS_TOL, tests/test_gpu_other_sizes.py): sizes with radix-3 / radix-5
# passes.  The reference evaluates those passes' twiddle angles in float too, and its element-wise distance from the
# exact transform grows with the length (N = 9216 complex: 1.4e-6).  The tests that use this bound also require the HIP
# result to be within 1e-6 (relative L2 AND element-wise) of the float64 transform of the same input.
MAXABS_TOL = PARITY_TOL
TONE_MAXABS_TOL = 2.0e-6
MIXED_RADIX_MAXABS_TOL = 2.0e-6


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A source-only checkout (no built shared objects travelled with it): build the product library and the
    # checker once, exactly as __graft_entry__.build() does.  This is compiling, not a fallback compute path:
    # if hipcc is missing the build fails loudly and so do the tests that need the library.
    lib = os.path.join(ROOT, "chowdsp_fft_amd", "libchowdsp_fft_hip.so")
    ora = os.path.join(ROOT, "oracle", "liboracle_pffft.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def reflib():
    if not oracle_lib.ref_available():
        pytest.skip("oracle/_ref/libchowdsp_fft_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    return oracle_lib.RefLib()


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(HERE, "golden", "reference_sse.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def gpu():
    """(api module, torch, device) -- fails loudly if the HIP library or the GPU is missing."""
    import torch

    assert torch.cuda.is_available(), "gpu tests need a visible MI355X"
    from chowdsp_fft_amd import api

    assert api.device_count() >= 1, "libchowdsp_fft_hip.so sees no HIP device"
    return api, torch, torch.device("cuda:0")


def assert_frames_close(got, want, tol=PARITY_TOL, what="", maxabs_tol=None):
    """per-frame relative L2 <= tol and max-abs/frame-max <= maxabs_tol (default: tol; see TONE_MAXABS_TOL)"""
    if maxabs_tol is None:
        maxabs_tol = tol
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    g2 = got.reshape(-1, got.shape[-1]) if got.ndim > 1 else got[None, :]
    w2 = want.reshape(-1, want.shape[-1]) if want.ndim > 1 else want[None, :]
    num = np.linalg.norm(g2 - w2, axis=1)
    den = np.maximum(np.linalg.norm(w2, axis=1), 1e-300)
    l2 = float((num / den).max())
    mx = float((np.abs(g2 - w2).max(axis=1) / np.maximum(np.abs(w2).max(axis=1), 1e-300)).max())
    assert l2 <= tol, f"{what}: relative L2 {l2:.3e} > {tol:.1e}"
    assert mx <= maxabs_tol, f"{what}: max-abs/frame-max {mx:.3e} > {maxabs_tol:.1e}"
    return l2, mx
