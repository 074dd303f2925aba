// Batched complex-to-complex kernels (interleaved fp32 re/im, N <= 4096 per frame, LDS-resident).
//
// Replaces, for a whole batch of frames at once, what the reference does per frame in
// pffft_transform_internal's COMPLEX branches (simd/chowdsp_fft_impl_sse.cpp:1543-1552 forward,
// :1577-1585 backward): de-interleave sweep + cfftf1_ps passes + pffft_cplx_finalize + pffft_zreorder.
// HBM traffic per frame: one read + one write of 8*N bytes; everything else stays in VGPRs / LDS.
#pragma once
#include "stockham.hpp"

namespace cfft
{
// FPB = frames (teams) per workgroup.  Persistent: each workgroup walks the batch with a grid stride so
// that the per-work-item twiddle registers are loaded once per workgroup, not once per frame.
template <class P, int FPB, bool INV>
__global__ __launch_bounds__ (P::T* FPB) void c2c_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ twtab,
                                                         unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<P::N>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    const unsigned team_base = (unsigned) team * P::N; // element offset of this team's frame inside the group
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const bool active = g * FPB + team < batch; // uniform per team
        // group base is workgroup-uniform (SGPR pair); everything per work-item is a 32-bit offset
        const float2* __restrict__ src = in + (size_t) g * (FPB * P::N);
        float2* __restrict__ dst = out + (size_t) g * (FPB * P::N);

        float2 v[P::R];
        if (active)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const float2 x = src[team_base + (unsigned) input_index<P> (t, s)];
                v[s] = INV ? make_float2 (x.y, x.x) : x;
            }
        }
        else
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = make_float2 (0.f, 0.f);
        }
        // the previous frame's last-stage LDS reads must be finished before stage 0 overwrites the image
        team_sync<P::T>();
        fft_core<P> (v, tw, lds, t);
        if (active)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const float2 y = v[s];
                dst[team_base + (unsigned) output_index<P> (t, s)] = INV ? make_float2 (y.y, y.x) : y;
            }
        }
    }
}
} // namespace cfft
