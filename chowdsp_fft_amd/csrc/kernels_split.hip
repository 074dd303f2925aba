// Instantiation + launch of the split-plane single-pass kernels (N = 32768 complex, real N = 65536).
#include "launch_util.hpp"
#include "split_plane_kernels.hpp"

namespace cfft::host
{
namespace
{
template <class K>
hipError_t launch_split (K kernel, const float2* in, float2* out, const float2* tw, const float2* rtw, bool real, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || in_stride > 0xFFFFFFFFull || out_stride > 0xFFFFFFFFull)
        return hipErrorInvalidValue;
    // one 1024-thread workgroup per CU (132 KiB of LDS): persistent over the batch
    const unsigned grid = persistent_grid (kernel, kSplitT, kSplitLdsBytes, batch);
    if (grid == 0)
        return hipErrorNoDevice;
    const unsigned is = in_stride ? (unsigned) in_stride : (unsigned) kSplitN, os = out_stride ? (unsigned) out_stride : (unsigned) kSplitN;
    (void) real;
    (void) rtw;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (kSplitT), kSplitLdsBytes, stream, in, out, tw, (unsigned) batch, is, os);
    return hipGetLastError();
}
} // namespace

hipError_t launch_c2c_split (bool inverse, const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    return inverse ? launch_split (c2c_split_kernel<true>, in, out, tw, nullptr, false, batch, stream, in_stride, out_stride)
                   : launch_split (c2c_split_kernel<false>, in, out, tw, nullptr, false, batch, stream, in_stride, out_stride);
}

hipError_t launch_real_split (bool forward, const float2* in, float2* out, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || in_stride > 0xFFFFFFFFull || out_stride > 0xFFFFFFFFull)
        return hipErrorInvalidValue;
    auto kernel = forward ? r2c_split_kernel<0> : c2r_split_kernel<0>;
    const unsigned grid = persistent_grid (kernel, kSplitT, kSplitLdsBytes, batch);
    if (grid == 0)
        return hipErrorNoDevice;
    const unsigned is = in_stride ? (unsigned) in_stride : (unsigned) kSplitN, os = out_stride ? (unsigned) out_stride : (unsigned) kSplitN;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (kSplitT), kSplitLdsBytes, stream, in, out, tw, real_tw, (unsigned) batch, is, os);
    return hipGetLastError();
}
} // namespace cfft::host
