// Layout conversions of the JUCE-engine adapter (juce_adapter.cpp), as streaming kernels.
//
// The reference adapter (chowdsp_fft_juce/chowdsp_fft_juce.cpp:47-86) converts between this library's packed real
// spectrum [X0.re, X(N/2).re, X1.re, X1.im, ...] and JUCE's interleaved layout (N/2+1 complex bins, optionally
// followed by the redundant negative-frequency bins) with scalar loops on the host, one frame per call.  Here the
// same conversions run on a whole batch of 2N-float device buffers.
#include "butterflies.hpp"
#include "launch.hpp"

#include <algorithm>

namespace cfft
{
// frame = N complex slots c[0..N-1] (2N floats); M = N/2.  After the in-place packed r2c: c[0] = (X0, X(N/2)).
// -> c[0] = (X0, 0), c[M] = (X(N/2), 0)   (chowdsp_fft_juce.cpp:57-60)
// -> mirror != 0: c[M+i] = conj(c[M-i]), i = 1..M-1   (:61-66)
__global__ __launch_bounds__ (256) void juce_unpack_kernel (float2* __restrict__ data, unsigned M, unsigned batch, int mirror)
{
    const unsigned per = mirror ? M : 1u; // work-items per frame: without the mirror only c[0] and c[M] change
    const size_t total = (size_t) batch * per;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride)
    {
        const size_t f = idx / per;
        const unsigned i = (unsigned) (idx % per);
        float2* __restrict__ c = data + f * 2 * M;
        if (i == 0)
        {
            const float2 p = c[0];
            c[0] = make_float2 (p.x, 0.f);
            c[M] = make_float2 (p.y, 0.f);
        }
        else if (mirror)
        {
            const float2 x = c[M - i];
            c[M + i] = make_float2 (x.x, -x.y);
        }
    }
}

// inverse direction: c[0] = (c[0].re, c[M].re) (chowdsp_fft_juce.cpp:71-73), and the 1/N of :85 applied to the
// spectrum instead of the signal (the transform is linear; for the power-of-two sizes JUCE uses, the scaling is exact
// either way)
__global__ __launch_bounds__ (256) void juce_pack_scale_kernel (float2* __restrict__ data, unsigned M, unsigned batch, float scale)
{
    const size_t total = (size_t) batch * M;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride)
    {
        const size_t f = idx / M;
        const unsigned i = (unsigned) (idx % M);
        float2* __restrict__ c = data + f * 2 * M;
        float2 x = c[i];
        if (i == 0)
            x.y = c[M].x;
        c[i] = make_float2 (x.x * scale, x.y * scale);
    }
}

__global__ __launch_bounds__ (256) void scale_kernel (float4* __restrict__ data, size_t quads, float scale)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride)
    {
        const float4 x = gload4 (&data[i]);
        gstore4 (&data[i], make_float4 (x.x * scale, x.y * scale, x.z * scale, x.w * scale));
    }
}
} // namespace cfft

namespace cfft::host
{
namespace
{
unsigned stream_grid (size_t items)
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
    return (unsigned) std::max<size_t> (1, std::min<size_t> ((items + 255) / 256, (size_t) dev.num_cus * 8));
}
} // namespace

hipError_t launch_juce_unpack (float2* data, unsigned m, size_t batch, bool mirror, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    const unsigned grid = stream_grid (mirror ? batch * m : batch);
    if (grid == 0)
        return hipErrorNoDevice;
    if (mirror)
        hipLaunchKernelGGL (juce_unpack_kernel, dim3 (grid), dim3 (256), 0, stream, data, m, (unsigned) batch, 1);
    else
        hipLaunchKernelGGL (juce_unpack_kernel, dim3 (grid), dim3 (256), 0, stream, data, m, (unsigned) batch, 0);
    return hipGetLastError();
}

hipError_t launch_juce_pack_scale (float2* data, unsigned m, size_t batch, float scale, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    const unsigned grid = stream_grid (batch * m);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (juce_pack_scale_kernel, dim3 (grid), dim3 (256), 0, stream, data, m, (unsigned) batch, scale);
    return hipGetLastError();
}

hipError_t launch_scale (float* data, size_t count, float scale, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    const unsigned grid = stream_grid (count / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (scale_kernel, dim3 (grid), dim3 (256), 0, stream, (float4*) data, count / 4, scale);
    return hipGetLastError();
}
} // namespace cfft::host
