// Instantiation + launch of the batched real (r2c / c2r) kernels (gfx950).
#include "launch_util.hpp"
#include "real_kernels.hpp"

namespace cfft::host
{
namespace
{
// the JUCE engine's layouts fused into the pair stage (real_kernels.hpp: LAYOUT): frames of 2 M complex slots, in place
template <int M, bool FWD>
hipError_t launch_real_juce_m (float2* inout, const float2* tw, const float2* rtw, size_t batch, hipStream_t stream, bool mirror, float scale)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    if constexpr (M >= kStreamedCoreN)
        return hipErrorNotSupported; // the caller keeps the two-launch form for these lengths
    else
    {
        const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
        if constexpr (FWD)
        {
            auto kernel = mirror ? &r2c_kernel<P, G::FPB, true, 2> : &r2c_kernel<P, G::FPB, true, 1>;
            const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes_real, ngroups);
            if (grid == 0)
                return hipErrorNoDevice;
            hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes_real, stream, (const float2*) inout, inout, tw, rtw, (unsigned) batch, (unsigned) (2 * M), (unsigned) (2 * M));
        }
        else
        {
            auto kernel = &c2r_kernel<P, G::FPB, true, 1>;
            const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes_real, ngroups);
            if (grid == 0)
                return hipErrorNoDevice;
            hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes_real, stream, (const float2*) inout, inout, tw, rtw, (unsigned) batch, (unsigned) (2 * M), (unsigned) (2 * M), scale);
        }
        return hipGetLastError();
    }
}

template <int M, bool FWD>
hipError_t launch_real_m (const float2* in, float2* out, const float2* tw, const float2* rtw, size_t batch, hipStream_t stream, unsigned in_stride, unsigned out_stride)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    if constexpr (M >= kStreamedCoreN)
    {
        // 1024 work-items, 128 VGPRs: twiddles streamed from L2, one frame per workgroup
        auto kernel = FWD ? r2c_kernel_streamed<P> : c2r_kernel_streamed<P>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, batch);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, out, tw, rtw, (unsigned) batch, in_stride ? in_stride : (unsigned) M, out_stride ? out_stride : (unsigned) M);
        return hipGetLastError();
    }
    else
    {
    const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
    const bool strided = (in_stride != 0 && in_stride != (unsigned) M) || (out_stride != 0 && out_stride != (unsigned) M);
    if constexpr (FWD)
    {
        // CFFT_REAL_PREFETCH: 0 never, 1 always, default 2 = only one-frame-per-workgroup plans (measured on MI355X:
        // N=8192 +2 %, N=1024 / N=4096 -2 % -- same picture as the complex kernels)
#ifndef CFFT_REAL_PREFETCH
#define CFFT_REAL_PREFETCH 2
#endif
        constexpr bool use_pf = (CFFT_REAL_PREFETCH == 1) || (CFFT_REAL_PREFETCH == 2 && G::FPB == 1);
        auto kernel = [] { // only the variant in use is instantiated
            if constexpr (use_pf)
                return &r2c_kernel_pf<P, G::FPB, false>;
            else
                return &r2c_kernel<P, G::FPB, false>;
        }();
        if (strided) // frames embedded in larger buffers: run-time strides, plain kernel
            kernel = &r2c_kernel<P, G::FPB, true>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes_real, ngroups);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes_real, stream, in, out, tw, rtw, (unsigned) batch, in_stride ? in_stride : (unsigned) M, out_stride ? out_stride : (unsigned) M);
    }
    else
    {
        // CFFT_C2R_PREFETCH_MIN: smallest M that takes the software-pipelined variant (packed batches).  Measured on
        // MI355X: M = 8192 (real N = 16384) 4.33 -> 4.98 TB/s, M = 4096 -1.4 %.
#ifndef CFFT_C2R_PREFETCH_MIN
#define CFFT_C2R_PREFETCH_MIN 8192
#endif
        constexpr bool c2r_pf = (G::FPB == 1 && M >= CFFT_C2R_PREFETCH_MIN);
        if constexpr (c2r_pf)
        {
            if (! strided)
            {
                auto kpf = c2r_kernel_pf<P>;
                const unsigned gpf = persistent_grid (kpf, G::threads, G::lds_bytes_real, ngroups);
                if (gpf == 0)
                    return hipErrorNoDevice;
                hipLaunchKernelGGL (kpf, dim3 (gpf), dim3 (G::threads), G::lds_bytes_real, stream, in, out, tw, rtw, (unsigned) batch);
                return hipGetLastError();
            }
        }
        auto kernel = &c2r_kernel<P, G::FPB, true>;
        if constexpr (! c2r_pf) // (otherwise packed batches never get here)
        {
            if (! strided)
                kernel = &c2r_kernel<P, G::FPB, false>;
        }
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes_real, ngroups);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes_real, stream, in, out, tw, rtw, (unsigned) batch, in_stride ? in_stride : (unsigned) M, out_stride ? out_stride : (unsigned) M, 1.0f);
    }
    return hipGetLastError();
    }
}
} // namespace

hipError_t launch_r2c (int m, const float2* in, float2* out, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || in_stride > kMaxFrameStride || out_stride > kMaxFrameStride)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_CORE_N (m, return (launch_real_m<N_, true> (in, out, stage_tw, real_tw, batch, stream, (unsigned) in_stride, (unsigned) out_stride)));
    return hipErrorInvalidValue;
}

hipError_t launch_real_juce (int m, bool forward, float2* inout, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream, bool mirror, float scale)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || (size_t) 2 * (size_t) m > kMaxFrameStride)
        return hipErrorInvalidValue;
    if (forward)
    {
        CFFT_DISPATCH_CORE_N (m, return (launch_real_juce_m<N_, true> (inout, stage_tw, real_tw, batch, stream, mirror, scale)));
    }
    else
    {
        CFFT_DISPATCH_CORE_N (m, return (launch_real_juce_m<N_, false> (inout, stage_tw, real_tw, batch, stream, mirror, scale)));
    }
    return hipErrorNotSupported;
}

hipError_t launch_c2r (int m, const float2* in, float2* out, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || in_stride > kMaxFrameStride || out_stride > kMaxFrameStride)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_CORE_N (m, return (launch_real_m<N_, false> (in, out, stage_tw, real_tw, batch, stream, (unsigned) in_stride, (unsigned) out_stride)));
    return hipErrorInvalidValue;
}
} // namespace cfft::host
