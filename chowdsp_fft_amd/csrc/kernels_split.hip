// Instantiation + launch of the split-plane single-pass kernels (N = 32768 complex, real N = 65536).
#include "launch_util.hpp"
#include "split_plane_kernels.hpp"

namespace cfft::host
{
namespace
{
template <class G, class K>
hipError_t launch_split (K kernel, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride, unsigned& grid, unsigned& is, unsigned& os)
{
    if (batch > 0x7FFFFFFFull || in_stride > 0xFFFFFFFFull || out_stride > 0xFFFFFFFFull)
        return hipErrorInvalidValue;
    // persistent over the batch: one 1024-thread workgroup (132 KiB of LDS) or two 512-thread ones (66 KiB each) per CU
    grid = persistent_grid (kernel, G::T, G::lds_bytes, batch);
    if (grid == 0)
        return hipErrorNoDevice;
    is = in_stride ? (unsigned) in_stride : (unsigned) G::N;
    os = out_stride ? (unsigned) out_stride : (unsigned) G::N;
    (void) stream;
    return hipSuccess;
}

template <class G>
hipError_t c2c_split (bool inverse, const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    auto kernel = inverse ? c2c_split_kernel<G, true> : c2c_split_kernel<G, false>;
    unsigned grid = 0, is = 0, os = 0;
    const hipError_t e = launch_split<G> (kernel, batch, stream, in_stride, out_stride, grid, is, os);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, in, out, tw, (unsigned) batch, is, os);
    return hipGetLastError();
}

template <class G>
hipError_t real_split (bool forward, const float2* in, float2* out, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    unsigned grid = 0, is = 0, os = 0;
    if (forward)
    {
        auto kernel = r2c_split_kernel<G, 0>;
        const hipError_t e = launch_split<G> (kernel, batch, stream, in_stride, out_stride, grid, is, os);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, in, out, tw, real_tw, (unsigned) batch, is, os);
    }
    else
    {
        auto kernel = c2r_split_kernel<G, 0>;
        const hipError_t e = launch_split<G> (kernel, batch, stream, in_stride, out_stride, grid, is, os);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, in, out, tw, real_tw, (unsigned) batch, is, os, 1.f);
    }
    return hipGetLastError();
}
// the JUCE engine's layouts by the pair stage itself, in place on frames of 2 G::N float2 (split_plane_kernels.hpp: LAYOUT)
template <class G>
hipError_t real_split_juce (bool forward, float2* inout, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, bool mirror, float scale)
{
    unsigned grid = 0, is = 0, os = 0;
    const size_t stride = 2 * (size_t) G::N;
    if (forward)
    {
        auto kernel = mirror ? r2c_split_kernel<G, 2> : r2c_split_kernel<G, 1>;
        const hipError_t e = launch_split<G> (kernel, batch, stream, stride, stride, grid, is, os);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, (const float2*) inout, inout, tw, real_tw, (unsigned) batch, is, os);
    }
    else
    {
        auto kernel = c2r_split_kernel<G, 1>;
        const hipError_t e = launch_split<G> (kernel, batch, stream, stride, stride, grid, is, os);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, (const float2*) inout, inout, tw, real_tw, (unsigned) batch, is, os, scale);
    }
    return hipGetLastError();
}
template <class G>
hipError_t conv_frames_split (bool real, const float2* in, const float2* H, float2* out, const float2* tw, const float2* real_tw, float scaling, size_t batch, hipStream_t stream)
{
    unsigned grid = 0, is = 0, os = 0;
    if (real)
    {
        auto kernel = conv_frames_real_split_kernel<G>;
        const hipError_t e = launch_split<G> (kernel, batch, stream, 0, 0, grid, is, os);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, in, H, out, tw, real_tw, scaling, (unsigned) batch);
    }
    else
    {
        auto kernel = conv_frames_c2c_split_kernel<G>;
        const hipError_t e = launch_split<G> (kernel, batch, stream, 0, 0, grid, is, os);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::T), G::lds_bytes, stream, in, H, out, tw, scaling, (unsigned) batch);
    }
    return hipGetLastError();
}
} // namespace

bool split_plane_length (int n) { return n == SplitGeom<1024>::N || n == SplitGeom<512>::N || n == SplitGeom<256>::N; }

hipError_t launch_c2c_split (int n, bool inverse, const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (n == SplitGeom<1024>::N)
        return c2c_split<SplitGeom<1024>> (inverse, in, out, tw, batch, stream, in_stride, out_stride);
    if (n == SplitGeom<512>::N)
        return c2c_split<SplitGeom<512>> (inverse, in, out, tw, batch, stream, in_stride, out_stride);
#if CFFT_EXPERIMENTS // option "split8192": the plain 8192-point transforms on the split-plane geometry
    if (n == SplitGeom<256>::N)
        return c2c_split<SplitGeom<256>> (inverse, in, out, tw, batch, stream, in_stride, out_stride);
#endif
    return hipErrorInvalidValue;
}

hipError_t launch_real_split (int m, bool forward, const float2* in, float2* out, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (m == SplitGeom<1024>::N)
        return real_split<SplitGeom<1024>> (forward, in, out, tw, real_tw, batch, stream, in_stride, out_stride);
    if (m == SplitGeom<512>::N)
        return real_split<SplitGeom<512>> (forward, in, out, tw, real_tw, batch, stream, in_stride, out_stride);
#if CFFT_EXPERIMENTS // option "split8192": the plain 8192-point transforms on the split-plane geometry
    if (m == SplitGeom<256>::N)
        return real_split<SplitGeom<256>> (forward, in, out, tw, real_tw, batch, stream, in_stride, out_stride);
#endif
    return hipErrorInvalidValue;
}

hipError_t launch_real_split_juce (int m, bool forward, float2* inout, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, bool mirror, float scale)
{
    if (batch == 0)
        return hipSuccess;
    if (m == SplitGeom<1024>::N)
        return real_split_juce<SplitGeom<1024>> (forward, inout, tw, real_tw, batch, stream, mirror, scale);
    if (m == SplitGeom<512>::N)
        return real_split_juce<SplitGeom<512>> (forward, inout, tw, real_tw, batch, stream, mirror, scale);
    return hipErrorNotSupported;
}

hipError_t launch_conv_frames_split (int n, bool real, const float2* in, const float2* H, float2* out, const float2* tw, const float2* real_tw, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (n == SplitGeom<1024>::N)
        return conv_frames_split<SplitGeom<1024>> (real, in, H, out, tw, real_tw, scaling, batch, stream);
    if (n == SplitGeom<512>::N)
        return conv_frames_split<SplitGeom<512>> (real, in, H, out, tw, real_tw, scaling, batch, stream);
    if (n == SplitGeom<256>::N)
        return conv_frames_split<SplitGeom<256>> (real, in, H, out, tw, real_tw, scaling, batch, stream);
#if CFFT_EXPERIMENTS
    if (n == SplitGeom<64>::N) // option "conv_wave2048": one wavefront per 2048-point frame
        return conv_frames_split<SplitGeom<64>> (real, in, H, out, tw, real_tw, scaling, batch, stream);
#endif
    return hipErrorInvalidValue;
}
} // namespace cfft::host
