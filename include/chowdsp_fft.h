/*
  chowdsp_fft.h -- drop-in C interface of the MI355X (gfx950) backend.

  This header declares, with identical names, argument order, enum values and C linkage, the eleven
  entry points of the reference library's public header (reference: chowdsp_fft.h:64-75 for the two
  enums, :81-163 for the functions).  A program written against the reference header compiles and links
  against libchowdsp_fft_hip.so unchanged; C++ callers keep spelling the functions
  chowdsp::fft::fft_transform (...) and keep the `= true` default for use_avx_if_available, C callers get
  plain C declarations.

  What differs behind the interface:
    * every transform runs as a HIP kernel on the current HIP device.  Pointers passed to
      fft_transform / fft_transform_unordered / fft_convolve_unordered / fft_accumulate may be host
      pointers (they are staged through device memory, one frame per call) or device pointers
      (hipMalloc / torch tensors; used in place).  There is no CPU fallback: if no HIP device is
      usable, setup creation returns NULL.
    * `work` is accepted and ignored (scratch lives in LDS / device memory).
    * `use_avx_if_available` is accepted for source compatibility and ignored;
      fft_simd_width_bytes() always reports 16, the value the reference reports on its SSE/NEON path.
    * the "unordered" frequency-domain layout is implementation defined in the reference (its SSE and
      AVX paths already differ); here it is the canonical ordered layout.
    * accepted sizes: the reference's rule -- N % 16 == 0 (complex) / N % 32 == 0 (real), N = 2^a 3^b 5^c
      (simd/chowdsp_fft_impl_common.hpp:168-177, 216-225), for every such N up to a complex length of 2^21
      (complex N <= 2097152, real N <= 4194304), and beyond that -- the reference has no upper bound -- for every N
      that is the product of two accepted complex lengths of at most 2^21 each, up to 2^27 complex points
      (real: N <= 2^25); those frames run through the plans of the two factors.  fft_gpu_size_supported()
      (chowdsp_fft_gpu.h) is the GPU-free test; sizes it rejects return NULL from setup creation, exactly like an
      unsupported size does in the reference (which returns the tagged pointer 0x1 there,
      chowdsp_fft.cpp:271-273 -- a defect this backend does not imitate).
    * one deliberate numerical deviation: for REAL sizes divisible by 25 the reference's FORWARD transform is
      wrong (radf5_ps with ido > 1, simd/chowdsp_fft_impl_sse.cpp:747-782; SSE and AVX alike: N = 800 is
      O(1) away from the DFT and does not round-trip).  This backend returns the exact DFT for those sizes;
      everywhere else results match the reference's SSE path within 1e-6 relative per frame.

  Throughput work should use the batched device-pointer extension in chowdsp_fft_gpu.h.
*/

#pragma once

#ifdef __cplusplus
#include <cstddef>

extern "C"
{
namespace chowdsp::fft
{
#else
#include <stdbool.h>
#include <stddef.h>
#endif

/* reference chowdsp_fft.h:64-68 -- FFT_FORWARD = 0, FFT_BACKWARD = 1 */
typedef enum
{
    FFT_FORWARD,
    FFT_BACKWARD
} fft_direction_t;

/* reference chowdsp_fft.h:71-75 -- FFT_REAL = 0, FFT_COMPLEX = 1 */
typedef enum
{
    FFT_REAL,
    FFT_COMPLEX
} fft_transform_t;

#ifdef __cplusplus
#define CHOWDSP_FFT_DEFAULT_TRUE = true
#else
#define CHOWDSP_FFT_DEFAULT_TRUE
#endif

/* Size in bytes of the caller-provided block fft_new_setup_preallocated() needs
   (reference chowdsp_fft.h:81).  Never smaller than the reference's own figure for the same (N, transform),
   so buffers sized for the reference remain valid. */
size_t fft_bytes_required (int N, fft_transform_t transform, bool use_avx_if_available CHOWDSP_FFT_DEFAULT_TRUE);

/* Create a plan for length-N transforms (reference chowdsp_fft.h:92).  The plan is immutable and may
   be shared by concurrent threads.  NULL for an unsupported N or when no HIP device is usable. */
void* fft_new_setup (int N, fft_transform_t transform, bool use_avx_if_available CHOWDSP_FFT_DEFAULT_TRUE);

/* Same, with the host-side plan header placed in caller memory `data` of at least
   fft_bytes_required() bytes (reference chowdsp_fft.h:114).  The caller keeps ownership of `data` and must
   not call fft_destroy_setup(); device-side tables come from a process-wide cache. */
void* fft_new_setup_preallocated (int N, fft_transform_t transform, void* data, bool use_avx_if_available CHOWDSP_FFT_DEFAULT_TRUE);

/* Release a plan made by fft_new_setup() (reference chowdsp_fft.h:119). */
void fft_destroy_setup (void* setup);

/* Width in bytes of the SIMD registers of the plan (reference chowdsp_fft.h:122); always 16 here. */
int fft_simd_width_bytes (void* setup);

/* One unscaled transform of one frame (reference chowdsp_fft.h:138): BACKWARD(FORWARD(x)) == N * x.
   COMPLEX: N interleaved (re, im) pairs in and out.
   REAL:    N floats in time domain; frequency domain is the packed layout
            [X0.re, X(N/2).re, X1.re, X1.im, ..., X(N/2-1).re, X(N/2-1).im].
   input and output may alias; `work` is ignored. */
void fft_transform (void* setup, const float* input, float* output, float* work, fft_direction_t direction);

/* Same transform, frequency-domain order left to the implementation (reference chowdsp_fft.h:145). */
void fft_transform_unordered (void* setup, const float* input, float* output, float* work, fft_direction_t direction);

/* dft_ab += (dft_a * dft_b) * scaling on frequency-domain frames produced by
   fft_transform_unordered (reference chowdsp_fft.h:154).  For REAL plans the two packed purely real
   bins (DC, Nyquist) are multiplied as reals.  Pointers may alias. */
void fft_convolve_unordered (void* setup, const float* dft_a, const float* dft_b, float* dft_ab, float scaling);

/* ab[i] = a[i] + b[i], i < N (reference chowdsp_fft.h:160). */
void fft_accumulate (void* setup, const float* a, const float* b, float* ab, int N);

/* 64-byte aligned host allocation / release (reference chowdsp_fft.h:162-163). */
void* aligned_malloc (size_t nb_bytes);
void aligned_free (void* p);

#undef CHOWDSP_FFT_DEFAULT_TRUE

#ifdef __cplusplus
} // namespace chowdsp::fft
} // extern "C"
#endif
