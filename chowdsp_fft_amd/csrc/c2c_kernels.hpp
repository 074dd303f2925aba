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
template <class P, int FPB, bool INV, bool STRIDED>
__global__ __launch_bounds__ (P::T* FPB) void c2c_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ twtab,
                                                         unsigned batch,
                                                         unsigned in_stride_arg,
                                                         unsigned out_stride_arg)
{
    CFFT_FRAME_STRIDES (STRIDED, P::N);
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<P::N>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    // strides: distance between consecutive frames in float2 (N for packed batches; SURVEY 8 f4 strided batches)
    const unsigned team_base = (unsigned) team * P::N; // element offset of this team's frame inside the LDS staging area
    const unsigned team_in = (unsigned) team * in_stride, team_out = (unsigned) team * out_stride;
    constexpr bool STAGED = (P::T < 16);
    float2* stage = smem + FPB * LdsFrame<P::N>::floats2; // staging area (STAGED only): FPB frames, natural order
    StagePrefetch<P::T * FPB, FPB * P::N, P::N> ahead; // (STAGED only) the next group's operands, requested a group ahead
    if constexpr (STAGED && kStagedPrefetch<P::N, false>)
        if (blockIdx.x < ngroups)
            ahead.fetch (in + (size_t) blockIdx.x * FPB * in_stride, in_stride, min ((unsigned) FPB, batch - blockIdx.x * FPB) * P::N, tid);
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const bool active = g * FPB + team < batch; // uniform per team
        // group base is workgroup-uniform (SGPR pair); everything per work-item is a 32-bit offset
        const float2* __restrict__ src = in + (size_t) g * FPB * in_stride;
        float2* __restrict__ dst = out + (size_t) g * FPB * out_stride;
        const unsigned valid = min ((unsigned) FPB, batch - g * FPB) * P::N; // elements of this group that exist

        float2 v[P::R];
        if constexpr (STAGED)
        {
            __syncthreads(); // previous group's copy-out has read the staging area
            if constexpr (kStagedPrefetch<P::N, false>)
            {
                ahead.put (stage, tid);
                const unsigned gn = g + gridDim.x;
                if (gn < ngroups)
                    ahead.fetch (in + (size_t) gn * FPB * in_stride, in_stride, min ((unsigned) FPB, batch - gn * FPB) * P::N, tid);
            }
            else
                stage_in_strided<P::T * FPB, FPB * P::N, P::N> (src, in_stride, stage, valid, tid);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const float2 x = stage[team_base + (unsigned) input_index<P> (t, s)];
                v[s] = INV ? make_float2 (x.y, x.x) : x;
            }
        }
        else if (active)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const float2 x = gload (&src[team_in + (unsigned) input_index<P> (t, s)]);
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
        if constexpr (STAGED)
        {
            // the team's own slice of the staging area: every input was read (and synchronised inside fft_core)
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                stage[team_base + (unsigned) output_index<P> (t, s)] = INV ? make_float2 (v[s].y, v[s].x) : v[s];
            __syncthreads();
            stage_out_strided<P::T * FPB, FPB * P::N, P::N> (dst, out_stride, stage, valid, tid);
        }
        else if (active)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const float2 y = v[s];
                gstore (&dst[team_out + (unsigned) output_index<P> (t, s)], INV ? make_float2 (y.y, y.x) : y);
            }
        }
    }
}

// CFFT_C2C_XCD_ORDER (round-4 A/B): 1 = the frames one XCD works through are CONSECUTIVE in memory (workgroups are dealt round-robin
// over the 8 XCDs, so loop index g belongs to XCD g % 8: XCD x owns ngroups / 8 consecutive groups) instead of every XCD striding
// through the whole batch -- fewer pages and DRAM rows shared between XCDs.  Any bijection is correct.
#ifndef CFFT_C2C_XCD_ORDER
#define CFFT_C2C_XCD_ORDER 0
#endif
#ifndef CFFT_C2C_XOR
#define CFFT_C2C_XOR 0
#endif
__device__ __forceinline__ unsigned c2c_group_map (unsigned g, unsigned ngroups)
{
    if (CFFT_C2C_XOR != 0 && (ngroups % 8u) == 0) // (experiment: which frame residue mod 8 an XCD works on)
        return g ^ (unsigned) CFFT_C2C_XOR;
    if (! CFFT_C2C_XCD_ORDER)
        return g;
    const unsigned q = ngroups / 8u, r = ngroups % 8u, x = g % 8u;
    return x * q + (x < r ? x : r) + g / 8u;
}

// Software-pipelined variant: the loads of the NEXT frame are issued before the current frame is computed and
// stored.  Besides hiding the load latency behind this frame's arithmetic, it puts the loads AHEAD of the
// stores in the wave's in-order vmcnt queue, so waiting for them (s_waitcnt vmcnt(R)) no longer has to wait
// for the previous frame's stores to be acknowledged by memory.  Costs 2*R more VGPRs.
template <class P, int FPB, bool INV, bool STRIDED>
__global__ __launch_bounds__ (P::T* FPB) void c2c_kernel_pf (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ twtab,
                                                            unsigned batch,
                                                            unsigned in_stride_arg,
                                                            unsigned out_stride_arg)
{
    CFFT_FRAME_STRIDES (STRIDED, P::N);
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<P::N>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    const unsigned team_in = (unsigned) team * in_stride, team_out = (unsigned) team * out_stride;

    // One frame: v holds it (already swapped for INV); the next frame's loads go out first, then the
    // arithmetic, then the stores.  The very first frame is peeled so that BOTH ways into the steady-state
    // loop carry the same queue shape (R loads, then R stores) and the loop-top wait is vmcnt(R), not vmcnt(0).
    // (The body is a macro, not a lambda: with a by-reference closure the INV instantiation keeps `v` in scratch.)
    float nre[P::R], nim[P::R]; // prefetch registers (re, im already swapped for INV)
// CFFT_PF_AT (round-4 A/B, three-stage plans only): where the next frame's loads are issued -- 0 at the top of the frame
// (default), 1 behind the first LDS exchange, 2 behind the second, 3 half and half.  A later request keeps fewer bytes in flight
// per CU (the frame-shaped copy of tools/membench5.hip runs 5 % faster with half the bytes in flight).
#ifndef CFFT_PF_AT
#define CFFT_PF_AT 0
#endif
#define CFFT_PF_LOADS(LO, HI)                                                                                       \
    if (gn_ < ngroups)                                                                                              \
    {                                                                                                               \
        const unsigned gnm_ = c2c_group_map (gn_, ngroups);                                                         \
        const bool actn_ = gnm_ * FPB + team < batch;                                                               \
        const float2* __restrict__ srcn_ = in + (size_t) gnm_ * FPB * in_stride;                                    \
        if (FPB == 1 || actn_)                                                                                      \
        {                                                                                                           \
            _Pragma ("unroll") for (int s = (LO); s < (HI); ++s)                                                    \
            {                                                                                                       \
                const float2 x_ = gload (&srcn_[team_in + (unsigned) input_index<P> (t, s)]);                       \
                nre[s] = INV ? x_.y : x_.x;                                                                         \
                nim[s] = INV ? x_.x : x_.y;                                                                         \
            }                                                                                                       \
        }                                                                                                           \
    }
#define CFFT_PF_BODY(G, V)                                                                                          \
    {                                                                                                               \
        const unsigned gm_ = c2c_group_map ((G), ngroups);                                                          \
        const bool active_ = gm_ * FPB + team < batch;                                                              \
        float2* __restrict__ dst_ = out + (size_t) gm_ * FPB * out_stride;                                          \
        const unsigned gn_ = (G) + gridDim.x;                                                                       \
        constexpr int at_ = (P::NS == 3) ? CFFT_PF_AT : 0;                                                          \
        if constexpr (at_ == 0)                                                                                     \
        {                                                                                                           \
            CFFT_PF_LOADS (0, P::R)                                                                                 \
            team_sync<P::T>();                                                                                      \
            fft_core<P> (V, tw, lds, t);                                                                            \
        }                                                                                                           \
        else                                                                                                        \
        {                                                                                                           \
            team_sync<P::T>();                                                                                      \
            stage_compute<P, 0> (V, tw);                                                                            \
            stage_write_lds<P, 0> (V, lds, t);                                                                      \
            team_sync<P::T>();                                                                                      \
            if constexpr (at_ == 1)                                                                                 \
                CFFT_PF_LOADS (0, P::R)                                                                             \
            if constexpr (at_ == 3)                                                                                 \
                CFFT_PF_LOADS (0, P::R / 2)                                                                         \
            stage_read_lds<P, 1> (V, lds, t);                                                                       \
            stage_compute<P, 1> (V, tw);                                                                            \
            team_sync<P::T>();                                                                                      \
            stage_write_lds<P, 1> (V, lds, t);                                                                      \
            team_sync<P::T>();                                                                                      \
            if constexpr (at_ == 2)                                                                                 \
                CFFT_PF_LOADS (0, P::R)                                                                             \
            if constexpr (at_ == 3)                                                                                 \
                CFFT_PF_LOADS (P::R / 2, P::R)                                                                      \
            stage_read_lds<P, 2> (V, lds, t);                                                                       \
            stage_compute<P, 2> (V, tw);                                                                            \
        }                                                                                                           \
        if (FPB == 1 || active_) /* FPB == 1: the store count per iteration is a compile-time constant */           \
        {                                                                                                           \
            _Pragma ("unroll") for (int s = 0; s < P::R; ++s)                                                       \
                gstore (&dst_[team_out + (unsigned) output_index<P> (t, s)],                                        \
                        INV ? make_float2 (V[s].y, V[s].x) : V[s]);                                                 \
        }                                                                                                           \
    }

    unsigned g = blockIdx.x;
    if (g >= ngroups)
        return;
    {
        const unsigned g0 = c2c_group_map (g, ngroups);
        const bool act0 = g0 * FPB + team < batch;
        const float2* __restrict__ src0 = in + (size_t) g0 * FPB * in_stride;
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
        {
            const float2 x = (FPB == 1 || act0) ? gload (&src0[team_in + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
            v[s] = INV ? make_float2 (x.y, x.x) : x;
        }
        CFFT_PF_BODY (g, v)
    }
    for (g += gridDim.x; g < ngroups; g += gridDim.x)
    {
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            v[s] = make_float2 (nre[s], nim[s]);
        CFFT_PF_BODY (g, v)
    }
#undef CFFT_PF_BODY
#undef CFFT_PF_LOADS
}

// One frame per 1024-thread workgroup with streamed twiddles (fft_core_streamed): the N = 16384 plan, whose 128 KiB frame
// still fits the CU's 160 KiB LDS, so it stays a single pass over HBM.  Run-time frame strides (packed or strided).
template <class P, bool INV>
__global__ __launch_bounds__ (P::T) void c2c_kernel_streamed (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ twtab,
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int t0 = threadIdx.x;
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
        {
            const float2 x = gload (&src[(unsigned) input_index<P> (t, s)]);
            v[s] = INV ? make_float2 (x.y, x.x) : x;
        }
        team_sync<P::T>(); // the previous frame's last-stage LDS reads are finished
        fft_core_streamed<P> (v, twtab, smem, t);
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            gstore (&dst[(unsigned) output_index<P> (t, s)], INV ? make_float2 (v[s].y, v[s].x) : v[s]);
    }
}
} // namespace cfft
