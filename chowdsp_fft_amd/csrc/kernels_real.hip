// Instantiation + launch of the batched real (r2c / c2r) kernels (gfx950).
#include "launch_util.hpp"
#include "real_kernels.hpp"

namespace cfft::host
{
namespace
{
template <int M, bool FWD>
hipError_t launch_real_m (const float2* in, float2* out, const float2* tw, const float2* rtw, size_t batch, hipStream_t stream)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
    if constexpr (FWD)
    {
        // CFFT_REAL_PREFETCH: 0 never, 1 always, default 2 = only one-frame-per-workgroup plans (measured on MI355X:
        // N=8192 +2 %, N=1024 / N=4096 -2 % -- same picture as the complex kernels)
#ifndef CFFT_REAL_PREFETCH
#define CFFT_REAL_PREFETCH 2
#endif
        constexpr bool use_pf = (CFFT_REAL_PREFETCH == 1) || (CFFT_REAL_PREFETCH == 2 && G::FPB == 1);
        auto kernel = use_pf ? r2c_kernel_pf<P, G::FPB> : r2c_kernel<P, G::FPB>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, ngroups);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, out, tw, rtw, (unsigned) batch);
    }
    else
    {
        auto kernel = c2r_kernel<P, G::FPB>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, ngroups);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, out, tw, rtw, (unsigned) batch);
    }
    return hipGetLastError();
}
} // namespace

hipError_t launch_r2c (int m, const float2* in, float2* out, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_N (m, return (launch_real_m<N_, true> (in, out, stage_tw, real_tw, batch, stream)));
    return hipErrorInvalidValue;
}

hipError_t launch_c2r (int m, const float2* in, float2* out, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_N (m, return (launch_real_m<N_, false> (in, out, stage_tw, real_tw, batch, stream)));
    return hipErrorInvalidValue;
}
} // namespace cfft::host
