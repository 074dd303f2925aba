"""chowdsp_fft_amd -- MI355X (gfx950) backend of the chowdsp_fft hot path.

`chowdsp_fft_amd.api` mirrors the reference's C interface (fft_new_setup / fft_transform /
fft_convolve_unordered ...) on top of libchowdsp_fft_hip.so; `chowdsp_fft_amd.csrc` holds the HIP
kernels and the C-ABI layer.  Importing this package does not touch the GPU.
"""
from . import api  # noqa: F401
from ._lib import FFT_BACKWARD, FFT_COMPLEX, FFT_FORWARD, FFT_REAL, LIB_PATH  # noqa: F401
