// Layout conversions of the JUCE-engine adapter (juce_adapter.cpp), as streaming kernels.
//
// The reference adapter (chowdsp_fft_juce/chowdsp_fft_juce.cpp:47-86) converts between this library's packed real
// spectrum [X0.re, X(N/2).re, X1.re, X1.im, ...] and JUCE's interleaved layout (N/2+1 complex bins, optionally
// followed by the redundant negative-frequency bins) with scalar loops on the host, one frame per call.  Here the
// same conversions run on a whole batch of 2N-float device buffers.
#include "butterflies.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

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

// ---- frames beyond 2^21 complex points (capi.cpp: huge_transform): tiled transposes of one [rows][cols] complex matrix --------
// out[c][r] = in[r][c] (* W_N^(r c) or its conjugate for the twiddled form).  32 x 32 tiles through LDS, both sides move 256
// contiguous bytes per 32 lanes; rows and cols are multiples of 16 (accepted frame lengths), edges are masked.
// W_N^j = hi[j >> shift] * lo[j & mask]: two tables of about sqrt(N) entries instead of N (tables.cpp).
namespace
{
template <int TW> // 0 plain, 1 times W_N^(r c), 2 times its conjugate
__global__ __launch_bounds__ (256) void transpose_kernel (const float2* __restrict__ in, float2* __restrict__ out, unsigned rows, unsigned cols,
                                                         const float2* __restrict__ lo, const float2* __restrict__ hi, unsigned shift)
{
    __shared__ float2 tile[32][33];
    const unsigned tx = threadIdx.x & 31u, ty = threadIdx.x >> 5; // 32 x 8
    const unsigned tiles_c = (cols + 31u) / 32u;
    const size_t ntiles = (size_t) tiles_c * ((rows + 31u) / 32u);
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x)
    {
        const unsigned r0 = (unsigned) (t / tiles_c) * 32u, c0 = (unsigned) (t % tiles_c) * 32u;
        __syncthreads();
#pragma unroll
        for (unsigned j = 0; j < 32u; j += 8u)
        {
            const unsigned r = r0 + ty + j, c = c0 + tx;
            if (r < rows && c < cols)
            {
                float2 x = in[(size_t) r * cols + c];
                if (TW != 0)
                {
                    const unsigned long long jj = (unsigned long long) r * c; // < N <= 2^27
                    const float2 wl = lo[jj & ((1ull << shift) - 1ull)], wh = hi[jj >> shift];
                    const float2 w = make_float2 (wl.x * wh.x - wl.y * wh.y, wl.x * wh.y + wl.y * wh.x);
                    x = (TW == 1) ? make_float2 (x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x)
                                  : make_float2 (x.x * w.x + x.y * w.y, x.y * w.x - x.x * w.y);
                }
                tile[ty + j][tx] = x;
            }
        }
        __syncthreads();
#pragma unroll
        for (unsigned j = 0; j < 32u; j += 8u)
        {
            const unsigned c = c0 + ty + j, r = r0 + tx;
            if (r < rows && c < cols)
                out[(size_t) c * rows + r] = tile[tx][ty + j];
        }
    }
}
} // namespace

// out[c][r] = in[r][c] for ONE [rows][cols] matrix; twiddle 0 none, +1 times W_N^(r c), -1 times its conjugate (N = rows * cols)
hipError_t launch_transpose (const float2* in, float2* out, unsigned rows, unsigned cols, int twiddle, const float2* tw_lo, const float2* tw_hi, unsigned shift, hipStream_t stream)
{
    if (rows == 0 || cols == 0 || in == out)
        return hipErrorInvalidValue;
    const size_t ntiles = (size_t) ((cols + 31u) / 32u) * ((rows + 31u) / 32u);
    auto k = twiddle == 0 ? transpose_kernel<0> : (twiddle > 0 ? transpose_kernel<1> : transpose_kernel<2>);
    const unsigned grid = persistent_grid (k, 256, 0, ntiles);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (256), 0, stream, in, out, rows, cols, tw_lo, tw_hi, shift);
    return hipGetLastError();
}
} // namespace cfft::host
