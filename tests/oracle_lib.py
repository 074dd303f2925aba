"""ctypes access to the two CPU checkers (test infrastructure only):

  * `Oracle`  -- oracle/liboracle_pffft.so, the plain-C restatement of the reference SSE path
  * `RefLib`  -- oracle/_ref/libchowdsp_fft_ref.so, the real reference compiled in place from
                 /root/reference by oracle/Makefile (SSE path with use_avx=False, AVX path with True)

Both operate on 64-byte aligned numpy buffers.  Nothing under chowdsp_fft_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle_pffft.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libchowdsp_fft_ref.so")

FORWARD, BACKWARD = 0, 1
REAL, COMPLEX = 0, 1


def aligned_empty(n_floats: int, align: int = 64) -> np.ndarray:
    raw = np.empty(n_floats * 4 + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n_floats * 4].view(np.float32)


def aligned_copy(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    out = aligned_empty(a.size)
    out[:] = a
    return out


def ensure_oracle_built() -> None:
    if not os.path.exists(ORACLE_SO):
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle_pffft.so"], check=True, capture_output=True)


class Oracle:
    def __init__(self):
        ensure_oracle_built()
        L = C.CDLL(ORACLE_SO)
        L.oracle_new_setup.restype = C.c_void_p
        L.oracle_new_setup.argtypes = [C.c_int, C.c_int]
        L.oracle_destroy_setup.argtypes = [C.c_void_p]
        L.oracle_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.oracle_transform_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_convolve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.oracle_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_num_passes.argtypes = [C.c_void_p]
        L.oracle_pass_radix.argtypes = [C.c_void_p, C.c_int]
        self.L = L

    def new_setup(self, n: int, transform: int):
        return self.L.oracle_new_setup(n, transform) or None

    def destroy_setup(self, s) -> None:
        self.L.oracle_destroy_setup(s)

    def transform(self, s, x: np.ndarray, direction: int, ordered: bool = True) -> np.ndarray:
        xin = aligned_copy(x)
        out = aligned_empty(xin.size)
        self.L.oracle_transform(s, xin.ctypes.data, out.ctypes.data, direction, 1 if ordered else 0)
        return out

    def transform_batch(self, s, x: np.ndarray, direction: int) -> np.ndarray:
        """x: [batch, frame_floats] -> same shape, ordered transform of every frame"""
        x2 = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x2)
        self.L.oracle_transform_batch(s, x2.ctypes.data, out.ctypes.data, x2.shape[0], direction)
        return out

    def convolve(self, s, a, b, ab, scaling: float) -> np.ndarray:
        a2, b2, ab2 = aligned_copy(a), aligned_copy(b), aligned_copy(ab)
        self.L.oracle_convolve(s, a2.ctypes.data, b2.ctypes.data, ab2.ctypes.data, scaling)
        return ab2

    def accumulate(self, a, b) -> np.ndarray:
        a2, b2 = aligned_copy(a), aligned_copy(b)
        out = aligned_empty(a2.size)
        self.L.oracle_accumulate(a2.ctypes.data, b2.ctypes.data, out.ctypes.data, a2.size)
        return out

    def passes(self, s):
        return [self.L.oracle_pass_radix(s, i) for i in range(self.L.oracle_num_passes(s))]


def ref_available() -> bool:
    return os.path.exists(REF_SO)


class RefLib:
    """The real reference (compiled from /root/reference in place; see oracle/Makefile)."""

    def __init__(self):
        if not ref_available():
            raise FileNotFoundError(REF_SO)
        L = C.CDLL(REF_SO)  # RTLD_LOCAL: its fft_* symbols stay private to this handle
        L.fft_new_setup.restype = C.c_void_p
        L.fft_new_setup.argtypes = [C.c_int, C.c_int, C.c_bool]
        L.fft_destroy_setup.argtypes = [C.c_void_p]
        L.fft_simd_width_bytes.argtypes = [C.c_void_p]
        L.fft_bytes_required.restype = C.c_size_t
        L.fft_bytes_required.argtypes = [C.c_int, C.c_int, C.c_bool]
        for name in ("fft_transform", "fft_transform_unordered"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.fft_convolve_unordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.fft_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L = L

    def new_setup(self, n: int, transform: int, use_avx: bool = False):
        h = self.L.fft_new_setup(n, transform, use_avx)
        # the reference tags SSE handles in the low pointer bits (chowdsp_fft.cpp:213-229) and tags a
        # NULL from an unsupported size too (:271-273), so "no setup" comes back as 0x1, not 0
        if h is None or (h & ~3) == 0:
            return None
        return h

    def destroy_setup(self, s) -> None:
        self.L.fft_destroy_setup(s)

    def simd_width(self, s) -> int:
        return self.L.fft_simd_width_bytes(s)

    def bytes_required(self, n, transform, use_avx=False) -> int:
        return self.L.fft_bytes_required(n, transform, use_avx)

    def transform(self, s, x, direction: int, ordered: bool = True) -> np.ndarray:
        xin = aligned_copy(x)
        out = aligned_empty(xin.size)
        work = aligned_empty(xin.size)
        fn = self.L.fft_transform if ordered else self.L.fft_transform_unordered
        fn(s, xin.ctypes.data, out.ctypes.data, work.ctypes.data, direction)
        return out

    def convolve(self, s, a, b, ab, scaling: float) -> np.ndarray:
        a2, b2, ab2 = aligned_copy(a), aligned_copy(b), aligned_copy(ab)
        self.L.fft_convolve_unordered(s, a2.ctypes.data, b2.ctypes.data, ab2.ctypes.data, scaling)
        return ab2

    def accumulate(self, s, a, b) -> np.ndarray:
        a2, b2 = aligned_copy(a), aligned_copy(b)
        out = aligned_empty(a2.size)
        self.L.fft_accumulate(s, a2.ctypes.data, b2.ctypes.data, out.ctypes.data, a2.size)
        return out


# ---- input generators shared by the tests and by tests/golden/make_golden.py -------------------------
def tone(n: int, complex_: bool, hz: float = 100.0) -> np.ndarray:
    """the reference tests' input (test/test.cpp:23-27, :82-85): sin/cos(3.14f * (hz/48000) * i) in float"""
    i = np.arange(n, dtype=np.float32)
    arg = (np.float32(3.14) * (np.float32(hz) / np.float32(48000.0))) * i
    if complex_:
        out = np.empty(2 * n, dtype=np.float32)
        out[0::2] = np.sin(arg.astype(np.float32)).astype(np.float32)
        out[1::2] = np.cos(arg.astype(np.float32)).astype(np.float32)
        return out
    return np.sin(arg).astype(np.float32)


def uniform(n_floats: int, seed: int = 1234) -> np.ndarray:
    """uniform(-1, 1) from a Mersenne-Twister stream (numpy RandomState = mt19937)"""
    return np.random.RandomState(seed).uniform(-1.0, 1.0, n_floats).astype(np.float32)


def impulse(n: int, complex_: bool, pos: int = 1) -> np.ndarray:
    out = np.zeros(2 * n if complex_ else n, dtype=np.float32)
    out[2 * pos if complex_ else pos] = 1.0
    return out


def rel_l2(a, b) -> float:
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a = np.asarray(a, dtype=dt)
    b = np.asarray(b, dtype=dt)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def max_over_frame_max(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
