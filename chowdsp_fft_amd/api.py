"""Python mirror of the chowdsp_fft operator interface, calling the HIP backend through its C ABI.

Function names, argument order and semantics follow include/chowdsp_fft.h (reference chowdsp_fft.h:81-163)
so that the parity tests read like the reference's own tests (reference test/test.cpp):

    setup = fft_new_setup(N, FFT_COMPLEX)
    fft_transform(setup, x, y, None, FFT_FORWARD)        # numpy (host pointers) or torch (device pointers)
    fft_transform_batch(setup, X, Y, FFT_FORWARD)        # torch tensors [batch, frame_floats] on the GPU

Arrays are passed by pointer, never copied here: numpy arrays go down as host pointers (the library
stages them), torch CUDA tensors as device pointers.  PyTorch is used only as a device-memory / stream
provider.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FFT_BACKWARD, FFT_COMPLEX, FFT_FORWARD, FFT_REAL  # noqa: F401  (re-exported)


class FFTError(RuntimeError):
    pass


def _ptr(a) -> int:
    """Raw address of a numpy array or torch tensor (must be contiguous fp32)."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
            raise TypeError("expected a C-contiguous float32 numpy array")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):  # torch tensor
        import torch

        if a.dtype != torch.float32 or not a.is_contiguous():
            raise TypeError("expected a contiguous float32 tensor")
        return a.data_ptr()
    raise TypeError(f"unsupported buffer type {type(a)!r}")


def _stream_handle(stream) -> int:
    """hipStream_t as an integer.  None -> torch's current stream if torch has initialised HIP, else NULL."""
    if stream is None:
        try:
            import torch

            if torch.cuda.is_available() and torch.cuda.is_initialized():
                return int(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return 0
    if isinstance(stream, int):
        return stream
    return int(stream.cuda_stream)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = _lib.lib().fft_gpu_last_error().decode(errors="replace")
        raise FFTError(f"{what} failed (code {rc}): {msg}")


# ---- legacy single-frame interface (names as in the reference header) -----------------------------
def fft_bytes_required(N: int, transform: int, use_avx_if_available: bool = True) -> int:
    return int(_lib.lib().fft_bytes_required(N, transform, use_avx_if_available))


def fft_new_setup(N: int, transform: int, use_avx_if_available: bool = True):
    """Returns an opaque handle (int) or None for an unsupported size / no device."""
    h = _lib.lib().fft_new_setup(N, transform, use_avx_if_available)
    return h or None


def fft_new_setup_preallocated(N: int, transform: int, data, use_avx_if_available: bool = True):
    h = _lib.lib().fft_new_setup_preallocated(N, transform, data, use_avx_if_available)
    return h or None


def fft_destroy_setup(setup) -> None:
    _lib.lib().fft_destroy_setup(setup)


def fft_simd_width_bytes(setup) -> int:
    return int(_lib.lib().fft_simd_width_bytes(setup))


def fft_transform(setup, input, output, work, direction: int) -> None:
    _lib.lib().fft_transform(setup, _ptr(input), _ptr(output), _ptr(work), direction)


def fft_transform_unordered(setup, input, output, work, direction: int) -> None:
    _lib.lib().fft_transform_unordered(setup, _ptr(input), _ptr(output), _ptr(work), direction)


def fft_convolve_unordered(setup, a, b, ab, scaling: float) -> None:
    _lib.lib().fft_convolve_unordered(setup, _ptr(a), _ptr(b), _ptr(ab), scaling)


def fft_accumulate(setup, a, b, ab, N: int) -> None:
    _lib.lib().fft_accumulate(setup, _ptr(a), _ptr(b), _ptr(ab), N)


def aligned_malloc(nb_bytes: int) -> int:
    return int(_lib.lib().aligned_malloc(nb_bytes) or 0)


def aligned_free(p: int) -> None:
    _lib.lib().aligned_free(p)


def aligned_array(n_floats: int):
    """float32 numpy view over a 64-byte aligned host block from aligned_malloc; returns (array, pointer)."""
    p = aligned_malloc(4 * n_floats)
    if not p:
        raise MemoryError
    arr = np.ctypeslib.as_array((C.c_float * n_floats).from_address(p))
    return arr, p


# ---- batched device-pointer extension (include/chowdsp_fft_gpu.h) -----------------------------------
def _batch_of(setup, t) -> int:
    n = _lib.lib().fft_gpu_setup_size(setup)
    if n < 0:
        raise FFTError("invalid setup handle")
    frame = n if _lib.lib().fft_gpu_setup_is_real(setup) else 2 * n
    total = t.numel() if hasattr(t, "numel") else t.size
    if total % frame:
        raise ValueError(f"buffer of {total} floats is not a whole number of {frame}-float frames")
    return total // frame


def fft_transform_batch(setup, input, output, direction: int, batch: int | None = None, stream=None) -> None:
    b = _batch_of(setup, input) if batch is None else batch
    _check(_lib.lib().fft_transform_batch(setup, _ptr(input), _ptr(output), b, direction, _stream_handle(stream)),
           "fft_transform_batch")


def fft_transform_batch_strided(setup, input, in_stride: int, output, out_stride: int, batch: int, direction: int,
                                stream=None) -> None:
    """frames embedded in larger buffers: frame b at input[b*in_stride:], output[b*out_stride:] (strides in floats)"""
    _check(_lib.lib().fft_transform_batch_strided(setup, _ptr(input), in_stride, _ptr(output), out_stride, batch,
                                                  direction, _stream_handle(stream)),
           "fft_transform_batch_strided")


def fft_transform_2d_batch(setup, rows: int, input, output, batch: int, direction: int, stream=None) -> None:
    _check(_lib.lib().fft_transform_2d_batch(setup, rows, _ptr(input), _ptr(output), batch, direction, _stream_handle(stream)),
           "fft_transform_2d_batch")


def fft_transform_unordered_batch(setup, input, output, direction: int, batch: int | None = None, stream=None) -> None:
    b = _batch_of(setup, input) if batch is None else batch
    _check(_lib.lib().fft_transform_unordered_batch(setup, _ptr(input), _ptr(output), b, direction, _stream_handle(stream)),
           "fft_transform_unordered_batch")


def fft_convolve_unordered_batch(setup, a, b, ab, scaling: float, b_is_shared: bool = False,
                                 batch: int | None = None, stream=None) -> None:
    nb = _batch_of(setup, a) if batch is None else batch
    _check(_lib.lib().fft_convolve_unordered_batch(setup, _ptr(a), _ptr(b), _ptr(ab), scaling, nb,
                                                   1 if b_is_shared else 0, _stream_handle(stream)),
           "fft_convolve_unordered_batch")


def fft_accumulate_batch(setup, a, b, ab, count: int | None = None, stream=None) -> None:
    n = (a.numel() if hasattr(a, "numel") else a.size) if count is None else count
    _check(_lib.lib().fft_accumulate_batch(setup, _ptr(a), _ptr(b), _ptr(ab), n, _stream_handle(stream)),
           "fft_accumulate_batch")


def fft_convolve_frames_batch(setup, x, dft_filter, y, scaling: float, batch: int | None = None, stream=None) -> None:
    nb = _batch_of(setup, x) if batch is None else batch
    _check(_lib.lib().fft_convolve_frames_batch(setup, _ptr(x), _ptr(dft_filter), _ptr(y), scaling, nb,
                                                _stream_handle(stream)),
           "fft_convolve_frames_batch")


def fft_transform_host_batch(setup, input, output, direction: int, batch: int | None = None) -> None:
    """numpy (host) buffers of `batch` frames; pinned in place and pipelined over PCIe (synchronous)."""
    b = _batch_of(setup, input) if batch is None else batch
    _check(_lib.lib().fft_transform_host_batch(setup, _ptr(input), _ptr(output), b, direction), "fft_transform_host_batch")


class JuceFFT:
    """Batched mirror of the reference's JUCE engine adapter `juce::dsp::ChowDSP_FFT`
    (chowdsp_fft_juce/chowdsp_fft_juce.cpp:13-126): one real and one complex plan of size 2**order, the three
    transform methods with JUCE's buffer shapes, applied to `batch` device buffers per call."""

    def __init__(self, order: int):
        if order < 5:  # :20-29, pffft needs N >= 32 for real transforms
            raise ValueError("order < 5 is not supported")
        self.order, self.size = order, 1 << order
        self.setup_real = fft_new_setup(self.size, FFT_REAL)
        self.setup_complex = fft_new_setup(self.size, FFT_COMPLEX)
        if self.setup_real is None or self.setup_complex is None:
            raise FFTError("no plan for this size / no device")

    def perform(self, input, output, inverse: bool, stream=None) -> None:
        """[batch, 2N] interleaved complex in -> out; the inverse is scaled by 1/N"""
        b = _batch_of(self.setup_complex, input)
        _check(_lib.lib().fft_juce_perform_batch(self.setup_complex, _ptr(input), _ptr(output), b, 1 if inverse else 0,
                                                 _stream_handle(stream)), "fft_juce_perform_batch")

    def perform_real_only_forward_transform(self, inout, ignore_negative_freqs: bool = False, stream=None) -> None:
        """[batch, 2N] buffers, N samples at the start of each -> N/2+1 (or N) interleaved complex bins"""
        b = (inout.numel() if hasattr(inout, "numel") else inout.size) // (2 * self.size)
        _check(_lib.lib().fft_juce_real_forward_batch(self.setup_real, _ptr(inout), b, 1 if ignore_negative_freqs else 0,
                                                      _stream_handle(stream)), "fft_juce_real_forward_batch")

    def perform_real_only_inverse_transform(self, inout, stream=None) -> None:
        b = (inout.numel() if hasattr(inout, "numel") else inout.size) // (2 * self.size)
        _check(_lib.lib().fft_juce_real_inverse_batch(self.setup_real, _ptr(inout), b, _stream_handle(stream)),
               "fft_juce_real_inverse_batch")

    def close(self) -> None:
        for name in ("setup_real", "setup_complex"):
            h = getattr(self, name, None)
            if h:
                fft_destroy_setup(h)
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConvEngine:
    """Streaming partitioned overlap-save convolver (include/chowdsp_fft_gpu.h, fft_conv_engine_*)."""

    def __init__(self, block: int, channels: int, ir, shared: bool = True, stream=None):
        ir_len = ir.shape[-1]
        self.block, self.channels = block, channels
        self._h = _lib.lib().fft_conv_engine_create(block, channels, _ptr(ir), ir_len, 1 if shared else 0, _stream_handle(stream))
        if not self._h:
            raise FFTError("fft_conv_engine_create failed: " + _lib.lib().fft_gpu_last_error().decode(errors="replace"))

    @property
    def partitions(self) -> int:
        return int(_lib.lib().fft_conv_engine_partitions(self._h))

    def process(self, x, y, stream=None) -> None:
        _check(_lib.lib().fft_conv_engine_process(self._h, _ptr(x), _ptr(y), _stream_handle(stream)), "fft_conv_engine_process")

    def reset(self, stream=None) -> None:
        _check(_lib.lib().fft_conv_engine_reset(self._h, _stream_handle(stream)), "fft_conv_engine_reset")

    def close(self) -> None:
        if self._h:
            _lib.lib().fft_conv_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(total_frames: int, shards: int, rank: int):
    """fft_gpu_shard_range: (first frame, frame count) of block `rank` (same rule as sharding.shard_range)"""
    first, count = C.c_size_t(0), C.c_size_t(0)
    _check(_lib.lib().fft_gpu_shard_range(total_frames, shards, rank, C.byref(first), C.byref(count)), "fft_gpu_shard_range")
    return first.value, count.value


def fft_transform_batch_multi_gpu(setups, inputs, outputs, counts, direction: int) -> None:
    """fft_transform_batch_multi_gpu: shard d = counts[d] frames of inputs[d] -> outputs[d] on the device of setups[d]"""
    n = len(setups)
    vp = C.c_void_p * n
    _check(_lib.lib().fft_transform_batch_multi_gpu(vp(*setups), n, vp(*[_ptr(a) for a in inputs]), vp(*[_ptr(a) for a in outputs]),
                                                    (C.c_size_t * n)(*counts), direction), "fft_transform_batch_multi_gpu")


def fft_convolve_frames_batch_multi_gpu(setups, xs, dft_filters, ys, counts, scaling: float) -> None:
    """fft_convolve_frames_batch_multi_gpu: shard d = counts[d] frames, ys[d] = BACKWARD(FORWARD(xs[d]) * dft_filters[d] * scaling)
    on the device of setups[d]; dft_filters[d] = the filter spectrum replicated on that device"""
    n = len(setups)
    vp = C.c_void_p * n
    _check(_lib.lib().fft_convolve_frames_batch_multi_gpu(vp(*setups), n, vp(*[_ptr(a) for a in xs]), vp(*[_ptr(a) for a in dft_filters]),
                                                          vp(*[_ptr(a) for a in ys]), (C.c_size_t * n)(*counts), scaling),
           "fft_convolve_frames_batch_multi_gpu")


def fft_gather_multi_gpu(shards, shard_devices, dsts, dst_devices) -> None:
    """fft_gather_multi_gpu: every shard tensor is copied, in order, into every non-None destination tensor"""
    n, m = len(shards), len(dsts)
    vp = C.c_void_p
    _check(_lib.lib().fft_gather_multi_gpu((vp * n)(*[_ptr(a) for a in shards]), (C.c_int * n)(*shard_devices),
                                           (C.c_size_t * n)(*[a.numel() for a in shards]), n,
                                           (vp * m)(*[_ptr(a) for a in dsts]), (C.c_int * m)(*dst_devices), m), "fft_gather_multi_gpu")


def defer_begin() -> None:
    _check(_lib.lib().fft_gpu_defer_begin(), "fft_gpu_defer_begin")


def defer_flush() -> None:
    _check(_lib.lib().fft_gpu_defer_flush(), "fft_gpu_defer_flush")


def release_stream_resources(stream=None) -> int:
    return int(_lib.lib().fft_gpu_release_stream_resources(_stream_handle(stream)))


def trim_tables(device: int = -1) -> int:
    """fft_gpu_trim_tables: free the device tables no plan references any more; returns the bytes given back"""
    return int(_lib.lib().fft_gpu_trim_tables(int(device)))


def table_cache_info(device: int = -1) -> dict:
    """fft_gpu_table_cache_info: {entries, bytes, idle_entries, idle_bytes} of the device-table cache"""
    import ctypes as C

    info = (C.c_size_t * 4)()
    _check(_lib.lib().fft_gpu_table_cache_info(int(device), info), "fft_gpu_table_cache_info")
    return {"entries": int(info[0]), "bytes": int(info[1]), "idle_entries": int(info[2]), "idle_bytes": int(info[3])}


def set_option(name: str, value: int) -> None:
    """fft_gpu_set_option: run-time knobs of the library (chowdsp_fft_gpu.h)"""
    _check(_lib.lib().fft_gpu_set_option(name.encode(), int(value)), f"fft_gpu_set_option({name})")


def workspace_bytes(setup, batch: int) -> int:
    """fft_gpu_workspace_bytes: device memory a two-pass plan caches per stream for a call of `batch` frames (0: none)"""
    return int(_lib.lib().fft_gpu_workspace_bytes(setup, batch))


def get_option(name: str) -> int:
    """fft_gpu_get_option: current value of a run-time knob (-1: not readable)"""
    return int(_lib.lib().fft_gpu_get_option(name.encode()))


def stream_synchronize(stream=None) -> None:
    """fft_gpu_stream_synchronize: waits for `stream` and reports a one-launch two-pass transform on it whose hand-offs
    did not complete (its output is not valid) -- the point where those forms' results are validated."""
    _check(_lib.lib().fft_gpu_stream_synchronize(_stream_handle(stream)), "fft_gpu_stream_synchronize")


def size_supported(N: int, transform: int) -> bool:
    return bool(_lib.lib().fft_gpu_size_supported(N, transform))


def plan_describe(N: int, transform: int):
    """dict(complex_len, passes, n1, n2, radices) or None for an unsupported size (host logic, no GPU)"""
    buf = (C.c_int * 8)()
    if _lib.lib().fft_gpu_plan_describe(N, transform, buf) != 0:
        return None
    return {"complex_len": buf[0], "passes": buf[1], "n1": buf[2], "n2": buf[3],
            "radices": [r for r in buf[4:8] if r]}


def device_count() -> int:
    return int(_lib.lib().fft_gpu_device_count())
