// Instantiation + launch of the batched complex-to-complex kernels (gfx950).
#include "c2c_kernels.hpp"
#include "wave_kernels.hpp"

#include <atomic>
#include <cstdlib>
#include "launch_util.hpp"

namespace cfft::host
{
namespace
{
template <int N, bool INV>
hipError_t launch_c2c_n (const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream, unsigned in_stride, unsigned out_stride)
{
    using P = typename CorePlan<N>::type;
    using G = Geometry<P>;
    // CFFT_C2C_STREAM_MIN_N (experiment): the streamed-twiddle kernel at 4 waves per SIMD for shorter plans measured
    // slower than resident twiddles + prefetch at 2 (N=4096 5.26 vs 5.74 TB/s, N=8192 4.83 vs 5.65)
#ifndef CFFT_C2C_STREAM_MIN_N
#define CFFT_C2C_STREAM_MIN_N kStreamedCoreN
#endif
    if constexpr (N >= CFFT_C2C_STREAM_MIN_N)
    {
        // 1024 work-items, 128 VGPRs: twiddles streamed from L2 (c2c_kernel_streamed), one frame per workgroup
        auto kernel = c2c_kernel_streamed<P, INV>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, batch);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, out, tw, (unsigned) batch, in_stride ? in_stride : (unsigned) N, out_stride ? out_stride : (unsigned) N);
        return hipGetLastError();
    }
    else
    {
    // CFFT_C2C_PREFETCH: 0 = never, 1 = every size, default (2) = only one-frame-per-workgroup plans, where the
    // software-pipelined kernel keeps a compile-time store count and measured +6.8 % (N=4096: 5.35 -> 5.71 TB/s);
    // multi-frame workgroups (N <= 2048) measured -2 % .. +1 % and keep the simple kernel.
#ifndef CFFT_C2C_PREFETCH
#define CFFT_C2C_PREFETCH 2
#endif
    constexpr bool use_pf = (CFFT_C2C_PREFETCH == 1) || (CFFT_C2C_PREFETCH == 2 && G::FPB == 1);
    const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
    const bool strided = (in_stride != 0 && in_stride != (unsigned) N) || (out_stride != 0 && out_stride != (unsigned) N);
    auto kernel = [] { // only the variant in use is instantiated
        if constexpr (use_pf)
            return &c2c_kernel_pf<P, G::FPB, INV, false>;
        else
            return &c2c_kernel<P, G::FPB, INV, false>;
    }();
    if (strided) // frames embedded in larger buffers: run-time strides, plain kernel
        kernel = &c2c_kernel<P, G::FPB, INV, true>;
    const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, ngroups);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, out, tw, (unsigned) batch, in_stride ? in_stride : (unsigned) N, out_stride ? out_stride : (unsigned) N);
    return hipGetLastError();
    }
}
} // namespace

#if CFFT_EXPERIMENTS
std::atomic<int> g_wave4096 { -1 };
bool wave4096_enabled()
{
    int v = g_wave4096.load();
    if (v < 0)
    {
        const char* e = std::getenv ("CHOWDSP_FFT_WAVE4096");
        v = (e != nullptr && e[0] == '1') ? 1 : 0;
        g_wave4096 = v;
    }
    return v != 0;
}

// N = 4096 by one wavefront per frame (wave_kernels.hpp); its twiddle rows follow the workgroup plan's table
template <bool INV>
hipError_t launch_c2c_wave4096 (const float2* in, float2* out, const float2* stage_tw, size_t batch, hipStream_t stream, unsigned in_stride, unsigned out_stride)
{
    auto kernel = c2c_wave4096_kernel<INV>;
    constexpr size_t lds = (size_t) 4 * kWavePlaneFloats * sizeof (float);
    const unsigned grid = persistent_grid (kernel, 256, lds, (batch + 3) / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    const float2* tw = stage_tw + CorePlan<4096>::type::tw_total;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (256), lds, stream, in, out, tw, (unsigned) batch, in_stride ? in_stride : 4096u, out_stride ? out_stride : 4096u);
    return hipGetLastError();
}

void set_wave4096 (bool on) { g_wave4096 = on ? 1 : 0; }
#else
void set_wave4096 (bool) {}
#endif

hipError_t launch_c2c (int n, bool inverse, const float2* in, float2* out, const float2* stage_tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || in_stride > kMaxFrameStride || out_stride > kMaxFrameStride)
        return hipErrorInvalidValue;
    // one wavefront per 4096-point frame instead of the 256-work-item workgroup plan (option "wave4096" /
    // CHOWDSP_FFT_WAVE4096=1): measured at the same rate (5.76 vs 5.80 TB/s, profiles/r02_wave4096.txt), kept as the
    // alternative form; packed and strided batches alike
#if CFFT_EXPERIMENTS
    if (n == 4096 && wave4096_enabled())
        return inverse ? launch_c2c_wave4096<true> (in, out, stage_tw, batch, stream, (unsigned) in_stride, (unsigned) out_stride)
                       : launch_c2c_wave4096<false> (in, out, stage_tw, batch, stream, (unsigned) in_stride, (unsigned) out_stride);
#endif
    if (inverse)
    {
        CFFT_DISPATCH_CORE_N (n, return (launch_c2c_n<N_, true> (in, out, stage_tw, batch, stream, (unsigned) in_stride, (unsigned) out_stride)));
    }
    else
    {
        CFFT_DISPATCH_CORE_N (n, return (launch_c2c_n<N_, false> (in, out, stage_tw, batch, stream, (unsigned) in_stride, (unsigned) out_stride)));
    }
    return hipErrorInvalidValue;
}
} // namespace cfft::host
