// Instantiation + launch of the frequency-domain multiply-accumulate / accumulate / fused convolution kernels.
#include "conv_kernels.hpp"
#include "launch_util.hpp"

namespace cfft::host
{
namespace
{
unsigned stream_grid (size_t items)
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
    const size_t want = (items + 255) / 256;
    return (unsigned) std::max<size_t> (1, std::min<size_t> (want, (size_t) dev.num_cus * 8));
}

template <int N>
hipError_t launch_cf_c2c_n (const float2* in, const float2* H, float2* out, const float2* tw, float scaling, size_t batch, hipStream_t stream)
{
    using P = typename CorePlan<N>::type;
    using G = Geometry<P>;
    auto kernel = conv_frames_c2c_kernel<P, G::FPB>;
    const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
    const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, ngroups);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, H, out, tw, scaling, (unsigned) batch);
    return hipGetLastError();
}

template <int M>
hipError_t launch_cf_real_m (const float2* in, const float2* H, float2* out, const float2* tw, const float2* rtw, float scaling, size_t batch, hipStream_t stream)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    auto kernel = conv_frames_real_kernel<P, G::FPB>;
    const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
    const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, ngroups);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, H, out, tw, rtw, scaling, (unsigned) batch);
    return hipGetLastError();
}
} // namespace

hipError_t launch_conv_mac (bool real_packed, const float* a, const float* b, float* ab, float scaling, size_t cplx_per_frame, size_t batch, bool b_is_shared, hipStream_t stream)
{
    const size_t total_pairs = batch * cplx_per_frame / 2;
    if (total_pairs == 0)
        return hipSuccess;
    const unsigned grid = stream_grid (total_pairs);
    if (grid == 0)
        return hipErrorNoDevice;
    if (real_packed)
        hipLaunchKernelGGL (conv_mac_kernel<true>, dim3 (grid), dim3 (256), 0, stream, (const float4*) a, (const float4*) b, (float4*) ab, scaling, (unsigned) (cplx_per_frame / 2), total_pairs, b_is_shared ? 1 : 0);
    else
        hipLaunchKernelGGL (conv_mac_kernel<false>, dim3 (grid), dim3 (256), 0, stream, (const float4*) a, (const float4*) b, (float4*) ab, scaling, (unsigned) (cplx_per_frame / 2), total_pairs, b_is_shared ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_accumulate (const float* a, const float* b, float* ab, size_t count, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    const unsigned grid = stream_grid ((count + 3) / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (accumulate_kernel, dim3 (grid), dim3 (256), 0, stream, a, b, ab, count);
    return hipGetLastError();
}

hipError_t launch_fdl_mac (const float* fdl, const float* H, float* Y, float scale, size_t frame_floats, unsigned channels, unsigned P, unsigned head, bool h_shared, hipStream_t stream)
{
    const unsigned ppf = (unsigned) (frame_floats / 4);
    const unsigned grid = stream_grid ((size_t) channels * ppf);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (fdl_mac_kernel, dim3 (grid), dim3 (256), 0, stream, (const float4*) fdl, (const float4*) H, (float4*) Y, scale, ppf, channels, P, head, h_shared ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_ols_assemble (const float* in, float* prev, float* frame, size_t block, unsigned channels, hipStream_t stream)
{
    const unsigned grid = stream_grid ((size_t) channels * block / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (ols_assemble_kernel, dim3 (grid), dim3 (256), 0, stream, (const float4*) in, (float4*) prev, (float4*) frame, (unsigned) (block / 4), channels);
    return hipGetLastError();
}

hipError_t launch_ols_extract (const float* y, float* out, size_t block, unsigned channels, hipStream_t stream)
{
    const unsigned grid = stream_grid ((size_t) channels * block / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (ols_extract_kernel, dim3 (grid), dim3 (256), 0, stream, (const float4*) y, (float4*) out, (unsigned) (block / 4), channels);
    return hipGetLastError();
}

hipError_t launch_conv_frames_c2c (int n, const float2* in, const float2* H, float2* out, const float2* stage_tw, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_N (n, return (launch_cf_c2c_n<N_> (in, H, out, stage_tw, scaling, batch, stream)));
    return hipErrorInvalidValue;
}

hipError_t launch_conv_frames_real (int m, const float2* in, const float2* H, float2* out, const float2* stage_tw, const float2* real_tw, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_N (m, return (launch_cf_real_m<N_> (in, H, out, stage_tw, real_tw, scaling, batch, stream)));
    return hipErrorInvalidValue;
}
} // namespace cfft::host
