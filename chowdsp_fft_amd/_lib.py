"""ctypes binding of libchowdsp_fft_hip.so -- the C-ABI drop-in library (include/chowdsp_fft.h,
include/chowdsp_fft_gpu.h).

The library is the product; this module only declares its prototypes.  There is deliberately no
fallback: if the shared object is missing or fails to load, importing raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHOWDSP_FFT_LIB: developer override used by tools/time_variants.py to A/B differently built libraries
LIB_PATH = os.environ.get("CHOWDSP_FFT_LIB") or os.path.join(_HERE, "libchowdsp_fft_hip.so")

# enums of include/chowdsp_fft.h (reference chowdsp_fft.h:64-75)
FFT_FORWARD, FFT_BACKWARD = 0, 1
FFT_REAL, FFT_COMPLEX = 0, 1

# every symbol include/*.h declares: name -> (restype, argtypes)
_vp, _fp, _sz, _i, _b, _f = C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_bool, C.c_float
PROTOTYPES = {
    # --- legacy (include/chowdsp_fft.h)
    "fft_bytes_required": (_sz, [_i, _i, _b]),
    "fft_new_setup": (_vp, [_i, _i, _b]),
    "fft_new_setup_preallocated": (_vp, [_i, _i, _vp, _b]),
    "fft_destroy_setup": (None, [_vp]),
    "fft_simd_width_bytes": (_i, [_vp]),
    "fft_transform": (None, [_vp, _fp, _fp, _fp, _i]),
    "fft_transform_unordered": (None, [_vp, _fp, _fp, _fp, _i]),
    "fft_convolve_unordered": (None, [_vp, _fp, _fp, _fp, _f]),
    "fft_accumulate": (None, [_vp, _fp, _fp, _fp, _i]),
    "aligned_malloc": (_vp, [_sz]),
    "aligned_free": (None, [_vp]),
    # --- extension (include/chowdsp_fft_gpu.h)
    "fft_gpu_device_count": (_i, []),
    "fft_gpu_set_device": (_i, [_i]),
    "fft_gpu_malloc": (_vp, [_sz]),
    "fft_gpu_free": (None, [_vp]),
    "fft_gpu_host_malloc": (_vp, [_sz]),
    "fft_gpu_host_free": (None, [_vp]),
    "fft_gpu_memcpy": (_i, [_vp, _vp, _sz, _i, _vp]),
    "fft_gpu_stream_synchronize": (_i, [_vp]),
    "fft_gpu_last_error": (C.c_char_p, []),
    "fft_gpu_set_option": (_i, [C.c_char_p, C.c_long]),
    "fft_gpu_get_option": (C.c_long, [C.c_char_p]),
    "fft_gpu_release_stream_resources": (_i, [_vp]),
    "fft_gpu_trim_tables": (_sz, [_i]),
    "fft_gpu_table_cache_info": (_i, [_i, C.POINTER(C.c_size_t)]),
    "fft_gpu_defer_begin": (_i, []),
    "fft_gpu_defer_flush": (_i, []),
    "fft_gpu_size_supported": (_i, [_i, _i]),
    "fft_gpu_plan_describe": (_i, [_i, _i, C.POINTER(C.c_int)]),
    "fft_gpu_setup_size": (_i, [_vp]),
    "fft_gpu_setup_is_real": (_i, [_vp]),
    "fft_gpu_setup_device": (_i, [_vp]),
    "fft_gpu_workspace_bytes": (_sz, [_vp, _sz]),
    "fft_juce_perform_batch": (_i, [_vp, _fp, _fp, _sz, _i, _vp]),
    "fft_juce_real_forward_batch": (_i, [_vp, _fp, _sz, _i, _vp]),
    "fft_juce_real_inverse_batch": (_i, [_vp, _fp, _sz, _vp]),
    "fft_transform_batch": (_i, [_vp, _fp, _fp, _sz, _i, _vp]),
    "fft_transform_batch_strided": (_i, [_vp, _fp, _sz, _fp, _sz, _sz, _i, _vp]),
    "fft_transform_unordered_batch": (_i, [_vp, _fp, _fp, _sz, _i, _vp]),
    "fft_transform_2d_batch": (_i, [_vp, _i, _fp, _fp, _sz, _i, _vp]),
    "fft_convolve_unordered_batch": (_i, [_vp, _fp, _fp, _fp, _f, _sz, _i, _vp]),
    "fft_accumulate_batch": (_i, [_vp, _fp, _fp, _fp, _sz, _vp]),
    "fft_convolve_frames_batch": (_i, [_vp, _fp, _fp, _fp, _f, _sz, _vp]),
    "fft_transform_host_batch": (_i, [_vp, _fp, _fp, _sz, _i]),
    "fft_gpu_shard_range": (_i, [_sz, _i, _i, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "fft_transform_batch_multi_gpu": (_i, [C.POINTER(C.c_void_p), _i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _i]),
    "fft_convolve_frames_batch_multi_gpu": (_i, [C.POINTER(C.c_void_p), _i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _f]),
    "fft_gather_multi_gpu": (_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_size_t), _i, C.POINTER(C.c_void_p), C.POINTER(C.c_int), _i]),
    "fft_conv_engine_create": (_vp, [_i, _i, _fp, _i, _i, _vp]),
    "fft_conv_engine_process": (_i, [_vp, _fp, _fp, _vp]),
    "fft_conv_engine_reset": (_i, [_vp, _vp]),
    "fft_conv_engine_partitions": (_i, [_vp]),
    "fft_conv_engine_destroy": (None, [_vp]),
}


def load(path: str = LIB_PATH, strict: bool = True) -> C.CDLL:
    """dlopen the library (RTLD_LOCAL) and attach prototypes.  Raises OSError if it is not built.
    strict=False (tools/time_variants.py only): tolerate older builds that lack newer entry points."""
    if not os.path.exists(path):
        raise OSError(
            f"{path} not found: build it with `make -C chowdsp_fft_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path."
        )
    # PyTorch-ROCm ships its own copy of the HIP runtime.  A process that uses both (tests, bench.py) must end up with ONE
    # runtime: load torch's first, so that this library's libamdhip64 dependency resolves to the copy already in the
    # process (loading ours first and torch's second leaves the second one without devices).  C / C++ hosts have no torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in PROTOTYPES.items():
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB
