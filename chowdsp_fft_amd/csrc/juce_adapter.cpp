// Batched, device-resident counterpart of the reference's JUCE FFT-engine adapter
// (chowdsp_fft_juce/chowdsp_fft_juce.cpp:32-86) -- second half of SURVEY section 8 row f3: the packed-real <->
// interleaved conversions performed on the device.  Each function does for `batch` buffers what one call of the
// adapter's method does for one: same buffer shapes, same scaling, same layout.
#include "../../include/chowdsp_fft_gpu.h"

#include "host_common.hpp"
#include "launch.hpp"
#include "plans.hpp"
#include "tables.hpp"

#include <cstdlib>

using namespace cfft::host;
namespace cf = chowdsp::fft;

namespace
{
int fail (const char* what, hipError_t e)
{
    set_last_error (what, e);
    return cf::FFT_GPU_ELAUNCH;
}

// One launch: the layout conversion is the pair stage's own load / store pattern (real_kernels.hpp / split_plane_kernels.hpp:
// LAYOUT) for every power-of-two N that runs in one pass (N <= 65536; round 3 added 32768 and 65536 on the split-plane kernels).  Returns FFT_GPU_OK, an error, or +1 = "no fused form for this setup" (longer / mixed-radix frames: the
// caller converts in a second launch).
int fused_real (void* setup, float* inout, size_t batch, bool forward, bool mirror, hipStream_t stream, const char* what)
{
    static const bool off = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_JUCE_FUSED");
        return e != nullptr && std::atoi (e) == 0;
    }();
    Setup* s = as_setup (setup);
    if (off || s == nullptr || s->tables == nullptr)
        return 1;
    const DeviceTables* t = s->tables;
    const int m = s->N / 2;
    if (t->gen_ns > 0 || t->real4_r1 > 0 || t->mixed_n2 > 0 || t->huge_r > 0 || t->core_n != m || (m & (m - 1)) != 0)
        return 1;
    const DeviceGuard guard (s->device);
    hipError_t e = hipErrorNotSupported;
    if (m == 32768)
        e = launch_real_split_juce (m, forward, (float2*) inout, t->stage_twiddles, t->real_twiddles, batch, stream, mirror, 1.0f / (float) s->N);
    else if (m == 16384 && cf::fft_gpu_get_option ("split16384") == 1)
        // (the split-plane table of this length follows the LDS-image plan's in the same allocation, as in capi.cpp run_transform)
        e = launch_real_split_juce (m, forward, (float2*) inout, t->stage_twiddles + cfft::CorePlan<16384>::type::tw_total, t->real_twiddles, batch, stream, mirror, 1.0f / (float) s->N);
    else if (m <= 8192)
        e = launch_real_juce (m, forward, (float2*) inout, t->stage_twiddles, t->real_twiddles, batch, stream, mirror, 1.0f / (float) s->N);
    if (e == hipErrorNotSupported)
    {
        (void) hipGetLastError();
        return 1;
    }
    return e == hipSuccess ? cf::FFT_GPU_OK : fail (what, e);
}
} // namespace

#pragma GCC visibility push(default)
extern "C"
{
namespace chowdsp::fft
{
// FFT::Instance::perform (:32-45): complex transform, the inverse scaled by 1/N
int fft_juce_perform_batch (void* setup_complex, const float* input, float* output, size_t batch, int inverse, void* stream)
{
    const int n = fft_gpu_setup_size (setup_complex);
    if (n < 0 || fft_gpu_setup_is_real (setup_complex) != 0 || input == nullptr || output == nullptr)
        return FFT_GPU_EINVAL;
    const int rc = fft_transform_batch (setup_complex, input, output, batch, inverse ? FFT_BACKWARD : FFT_FORWARD, stream);
    if (rc != FFT_GPU_OK || ! inverse)
        return rc;
    const DeviceGuard guard (fft_gpu_setup_device (setup_complex));
    const hipError_t e = launch_scale (output, batch * 2 * (size_t) n, 1.0f / (float) n, (hipStream_t) stream);
    return e == hipSuccess ? FFT_GPU_OK : fail ("fft_juce_perform_batch: scale kernel launch", e);
}

// performRealOnlyForwardTransform (:47-67): `batch` buffers of 2N floats, the N real samples at the start of each.
// Out: N/2+1 interleaved complex bins (Im of DC and Nyquist zero); ignore_negative_freqs == 0 also fills bins
// N/2+1 .. N-1 with the conjugates.
int fft_juce_real_forward_batch (void* setup_real, float* inout, size_t batch, int ignore_negative_freqs, void* stream)
{
    const int n = fft_gpu_setup_size (setup_real);
    if (n < 0 || fft_gpu_setup_is_real (setup_real) != 1 || inout == nullptr)
        return FFT_GPU_EINVAL;
    if (batch == 0)
        return FFT_GPU_OK;
    const int fused = fused_real (setup_real, inout, batch, true, ignore_negative_freqs == 0, (hipStream_t) stream, "fft_juce_real_forward_batch: fused launch");
    if (fused <= 0)
        return fused;
    const int rc = fft_transform_batch_strided (setup_real, inout, 2 * (size_t) n, inout, 2 * (size_t) n, batch, FFT_FORWARD, stream);
    if (rc != FFT_GPU_OK)
        return rc;
    const DeviceGuard guard (fft_gpu_setup_device (setup_real));
    const hipError_t e = launch_juce_unpack ((float2*) inout, (unsigned) n / 2, batch, ignore_negative_freqs == 0, (hipStream_t) stream);
    return e == hipSuccess ? FFT_GPU_OK : fail ("fft_juce_real_forward_batch: unpack kernel launch", e);
}

// performRealOnlyInverseTransform (:69-86): in: N/2+1 interleaved bins at the start of each 2N-float buffer; out: the N
// real samples, scaled by 1/N, at the start of each buffer (the rest of the buffer is left as the reference leaves
// it: unspecified).
int fft_juce_real_inverse_batch (void* setup_real, float* inout, size_t batch, void* stream)
{
    const int n = fft_gpu_setup_size (setup_real);
    if (n < 0 || fft_gpu_setup_is_real (setup_real) != 1 || inout == nullptr)
        return FFT_GPU_EINVAL;
    if (batch == 0)
        return FFT_GPU_OK;
    const int fused = fused_real (setup_real, inout, batch, false, false, (hipStream_t) stream, "fft_juce_real_inverse_batch: fused launch");
    if (fused <= 0)
        return fused;
    hipError_t e;
    {
        const DeviceGuard guard (fft_gpu_setup_device (setup_real));
        e = launch_juce_pack_scale ((float2*) inout, (unsigned) n / 2, batch, 1.0f / (float) n, (hipStream_t) stream);
    }
    if (e != hipSuccess)
        return fail ("fft_juce_real_inverse_batch: pack kernel launch", e);
    return fft_transform_batch_strided (setup_real, inout, 2 * (size_t) n, inout, 2 * (size_t) n, batch, FFT_BACKWARD, stream);
}
} // namespace chowdsp::fft
}
#pragma GCC visibility pop
