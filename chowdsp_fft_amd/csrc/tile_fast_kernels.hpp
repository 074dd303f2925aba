// Two-pass (four-step) transforms of N = n1 x n2 = 2^17 .. 2^20 points: both passes register-direct.
//
// multipass_kernels.hpp's generic tile kernels stage a tile through LDS three times (coalesced load -> per-frame LDS
// FFT with 2-4 exchanges -> coalesced store) and run at 2.0-2.35 TB/s; its 256 x 256 fast path (global loads straight into
// VGPRs, two register stages, ONE exchange) is 40 % faster per pass.  This file is that fast path for core lengths
// n = 256, 512 and 1024, built like split_plane_kernels.hpp:
//   * a tile = 16 adjacent columns (pass A) / rows (pass B); a column is handled by a TEAM of n/32 work-items with 32 points
//     each, the 16 teams INTERLEAVED across lanes (lane = column + 16 * team member), so that one wave-instruction touches
//     4 x 128 contiguous bytes of the strided side;
//   * n = 32 x (n/32): stage 0 = one radix-32 butterfly per work-item, stage 1 = 32 butterflies of radix n/32 per column
//     (32 / (n/32) per work-item), ONE exchange between them through LDS, one component at a time (the plane of a tile is
//     16 x (n + n/32) floats: 68 KiB for n = 1024, two workgroups per CU);
//   * pass A re-maps its work-items across the exchange (lanes then run along the butterfly index, 32 consecutive output
//     indices kb per half-wave), which makes the plain layout W[a][kb] of the intermediate contiguous for pass A's stores
//     (256 bytes per half-wave) AND for pass B's loads (16 rows kb of one column a: 128 bytes);
//   * stage-1 twiddles W_n^(a k) in registers, factorised like the butterflies (split_plane_kernels.hpp): 10 / 6 / 4 values
//     per butterfly index for n = 1024 / 512 / 256.
// Replaces, per frame, the reference's in-cache pass loop (cfftf1_ps, simd/chowdsp_fft_impl_sse.cpp:372-432;
// chowdsp_fft.h:131-134 on scratch for large N).
#pragma once
#include "split_plane_kernels.hpp"

#ifndef CFFT_PARF_GROUP
#define CFFT_PARF_GROUP(N2) ((N2) >= 1024 ? 4 : 2)
#endif
namespace cfft
{
template <int N_>
struct FastTile
{
    static constexpr int n = N_;
    static constexpr int Tn = N_ / 32;        // work-items per column / row
    static constexpr int R1 = N_ / 32;        // radix of stage 1
    static constexpr int Q = 32 / Tn;         // stage-1 butterflies per work-item
    static constexpr int threads = 16 * Tn;   // 128 / 256 / 512
    // distance of consecutive teams' images in the plane: >= n + n/32 and = 2 (mod 32), so that the 16 teams x 2 team
    // members of a half-wave fall on 32 different banks (bank = 2 c + member + const)
    static constexpr int S = ((N_ + N_ / 32 + 31) / 32) * 32 + 2;
    static constexpr size_t lds_bytes = (size_t) 16 * S * sizeof (float);
    // minimum waves per SIMD asked of the compiler: two 512-thread workgroups per CU for n = 1024 (4 waves per SIMD, 128
    // VGPRs), CFFT_FAST_TILE_WAVES overrides (A/B)
#ifndef CFFT_FAST_TILE_WAVES
#define CFFT_FAST_TILE_WAVES 0
#endif
    static constexpr int min_waves = CFFT_FAST_TILE_WAVES > 0 ? CFFT_FAST_TILE_WAVES : 2; // (4 for n = 1024 spills: 2^19 -11 %)
    static constexpr int tw_rows = R1 == 32 ? 10 : (R1 == 16 ? 6 : 4); // rows [j][k < 32] of the stage-1 table
};

// 8-point forward DFT with the stage twiddle W^(a k) applied first, a = n1 + 4 n2 (fft_reg<8> is 2 (n2) x 4 (n1)):
// w[0..2] = W^(n1 k), n1 = 1..3, after the inner 2-point transforms; w[3] = W^(4 k) before them.
__device__ __forceinline__ void fft_reg8_tw (float2 (&u)[8], const float2 (&w)[4])
{
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1)
        u[n1 + 4] = cmul (u[n1 + 4], w[3]);
    bfly2 (u[0], u[4]);
    bfly2 (u[1], u[5]);
    bfly2 (u[2], u[6]);
    bfly2 (u[3], u[7]);
    // now u[n1] = A[n1][k2 = 0], u[n1 + 4] = A[n1][k2 = 1]
#pragma unroll
    for (int n1 = 1; n1 < 4; ++n1)
    {
        u[n1] = cmul (u[n1], w[n1 - 1]);
        u[n1 + 4] = cmul (u[n1 + 4], w[n1 - 1]);
    }
    u[5] = mul_w8_1 (u[5]);
    u[6] = mul_mi (u[6]);
    u[7] = mul_w8_3 (u[7]);
    bfly4 (u[0], u[1], u[2], u[3]);
    bfly4 (u[4], u[5], u[6], u[7]);
    const float2 e0 = u[0], e1 = u[1], e2 = u[2], e3 = u[3];
    const float2 o0 = u[4], o1 = u[5], o2 = u[6], o3 = u[7];
    u[0] = e0; u[1] = o0; u[2] = e1; u[3] = o1;
    u[4] = e2; u[5] = o2; u[6] = e3; u[7] = o3;
}

// XCD-aware order of work items (speed only; any bijection is correct): blocks are dealt round-robin over the 8 XCDs, so
// block b and b + 8 share an L2.  Item v of the persistent loop (v = blockIdx + k gridDim, gridDim a multiple of 8 on a full
// chip) is mapped so that the items one XCD works through are CONSECUTIVE tiles: the rows whose mirrored halves straddle a
// 128-byte line then share that line inside one L2 instead of fetching / writing it from two.
// the mirrored halves' accesses: default cache policy (0), so that the two tiles that share a line meet in L2, or nontemporal (1, round 3)
#ifndef CFFT_REAL4_MIRROR_NT
#define CFFT_REAL4_MIRROR_NT 0
#endif
__device__ __forceinline__ void real4_mirror_store (float2* p, float2 v)
{
    if (CFFT_REAL4_MIRROR_NT)
        gstore (p, v);
    else
        *p = v;
}
__device__ __forceinline__ size_t xcd_order (size_t v, size_t nwork)
{
    const size_t per = (nwork + 7) / 8;
    return (v % 8) * per + v / 8;
}

// the stage-1 twiddles of butterfly index i (rows [j][i] of the table)
template <class FT>
__device__ __forceinline__ void ft_load_twiddles (float2 (&w)[10], const float2* __restrict__ tw, int i)
{
#pragma unroll
    for (int j = 0; j < FT::tw_rows; ++j)
        w[j] = tw[j * 32 + i];
}

// the stage-1 butterfly number q (radix FT::R1) on v[q R1 .. q R1 + R1)
template <class FT, int Q_>
__device__ __forceinline__ void ft_stage1 (float2 (&v)[32], const float2 (&w)[10])
{
    float2 u[FT::R1];
#pragma unroll
    for (int a = 0; a < FT::R1; ++a)
        u[a] = v[Q_ * FT::R1 + a];
    if constexpr (FT::R1 == 32)
        fft_reg32<true> (u, w);
    else if constexpr (FT::R1 == 16)
    {
        const float2 w6[6] = { w[0], w[1], w[2], w[3], w[4], w[5] };
        fft_reg16_tw (u, w6);
    }
    else
    {
        const float2 w4[4] = { w[0], w[1], w[2], w[3] };
        fft_reg8_tw (u, w4);
    }
#pragma unroll
    for (int a = 0; a < FT::R1; ++a)
        v[Q_ * FT::R1 + a] = u[a];
}
template <class FT>
__device__ __forceinline__ void ft_stage1_all (float2 (&v)[32], const float2 (&w0)[10], const float2 (&w1)[10], const float2 (&w2)[10], const float2 (&w3)[10])
{
    ft_stage1<FT, 0> (v, w0);
    if constexpr (FT::Q > 1)
        ft_stage1<FT, 1> (v, w1);
    if constexpr (FT::Q > 2)
    {
        ft_stage1<FT, 2> (v, w2);
        ft_stage1<FT, 3> (v, w3);
    }
}

// The exchange.  Stage-0 role (c, m): team c = tid & 15, member m = tid >> 4, holds Y_c[32 m + b] in v[b].
// Stage-1 role: SWAPPED (pass A): butterfly i = tid & 31 of the columns cc_q = (tid >> 5) + (Tn / 2) q;
//               plain   (pass B): butterflies i_q = m + Tn q of team c.   v[q R1 + a] = X[i + 32 a].
// Addresses: team image at c S; write phys (32 m + b) = 33 m + b; read phys (i + 32 a) = i + 33 a  (i < 32).
template <class FT, bool SWAPPED>
__device__ __forceinline__ void ft_exchange (float2 (&v)[32], float* __restrict__ plane, int tid)
{
    const int c = tid & 15, m = tid >> 4;
    float* __restrict__ wp = plane + c * FT::S + 33 * m;
    const float* __restrict__ rp = SWAPPED ? plane + (tid >> 5) * FT::S + (tid & 31) : plane + c * FT::S + m;
    constexpr int rq = SWAPPED ? (FT::Tn / 2) * FT::S : FT::Tn; // distance between a work-item's butterflies q
    __syncthreads(); // the previous tile's reads of the plane are finished
#pragma unroll
    for (int b = 0; b < 32; ++b)
        wp[b] = v[b].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FT::Q; ++q)
#pragma unroll
        for (int a = 0; a < FT::R1; ++a)
            v[q * FT::R1 + a].x = rp[q * rq + 33 * a];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 32; ++b)
        wp[b] = v[b].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FT::Q; ++q)
#pragma unroll
        for (int a = 0; a < FT::R1; ++a)
            v[q * FT::R1 + a].y = rp[q * rq + 33 * a];
}

// ---- pass A: n2-point column FFTs of 16 adjacent columns, times W_N^{a kb}, into W[a][kb] ------------------------------
template <int N2, bool SWAP_IN>
__global__ __launch_bounds__ (FastTile<N2>::threads, FastTile<N2>::min_waves) void pass_a_fast_kernel (const float2* __restrict__ in,
                                                                           float2* __restrict__ work,
                                                                           const float2* __restrict__ stage_tw, // [rows][32] of the n2 core
                                                                           const float2* __restrict__ big_tw,   // [a][kb] = W_N^{a kb}
                                                                           unsigned n1,
                                                                           unsigned batch,
                                                                           size_t in_stride)
{
    using FT = FastTile<N2>;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[10];
    ft_load_twiddles<FT> (w, stage_tw, tid0 & 31); // stage-1 role: butterfly index i = tid & 31
    const unsigned tiles = n1 / 16;
    const size_t frame_len = (size_t) n1 * N2;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t wk = blockIdx.x; wk < nwork; wk += gridDim.x)
    {
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned a0 = (unsigned) (wk % tiles) * 16u;
        const float2* __restrict__ src = in + (size_t) f * in_stride + a0 + (unsigned) (tid & 15);
        const unsigned m = (unsigned) tid >> 4;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
        {
            const float2 x = gload (&src[(size_t) (m + FT::Tn * a) * n1]);
            v[a] = SWAP_IN ? make_float2 (x.y, x.x) : x;
        }
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v, w);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, true> (v, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v, w, w, w, w);
        CFFT_PHASE_FENCE();
        // column cc_q, output indices kb = i + 32 b: 32 consecutive kb per half-wave
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        const unsigned i = (unsigned) ts & 31u;
#pragma unroll
        for (int q = 0; q < FT::Q; ++q)
        {
            const unsigned a = a0 + ((unsigned) ts >> 5) + (unsigned) (FT::Tn / 2) * q;
            const float2* __restrict__ twp = big_tw + (size_t) a * N2 + i;
            float2* __restrict__ dst = work + (size_t) f * frame_len + (size_t) a * N2 + i;
#pragma unroll
            for (int b = 0; b < FT::R1; ++b)
                dst[32 * b] = cmul (v[q * FT::R1 + b], twp[32 * b]); // plain store: the workspace should stay cached
        }
    }
}

// ---- pass B: n1-point row FFTs of 16 adjacent rows kb, transposed store X[n2 ka + kb] ----------------------------------
// MULH (round 3, fft_convolve_frames_batch of these sizes): every bin is multiplied by H[bin] * scaling on its way out, so
// that forward transform + product are one sweep pair instead of two (the stand-alone conv_mac pass reads and writes the
// whole batch once more); the backward transform follows as usual.
template <int N1, bool SWAP_OUT, bool MULH = false>
__global__ __launch_bounds__ (FastTile<N1>::threads, FastTile<N1>::min_waves) void pass_b_fast_kernel (const float2* __restrict__ work,
                                                                           float2* __restrict__ out,
                                                                           const float2* __restrict__ stage_tw, // [rows][32] of the n1 core
                                                                           unsigned n2,
                                                                           unsigned batch,
                                                                           size_t out_stride,
                                                                           const float2* __restrict__ H = nullptr,
                                                                           float scaling = 1.f)
{
    using FT = FastTile<N1>;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[FT::Q][10];
#pragma unroll
    for (int q = 0; q < FT::Q; ++q)
        ft_load_twiddles<FT> (w[q], stage_tw, (tid0 >> 4) + FT::Tn * q); // butterfly index i_q = member + Tn q
    const unsigned tiles = n2 / 16;
    const size_t frame_len = (size_t) N1 * n2;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t wk = blockIdx.x; wk < nwork; wk += gridDim.x)
    {
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned kb0 = (unsigned) (wk % tiles) * 16u;
        const unsigned m = (unsigned) tid >> 4;
        const float2* __restrict__ src = work + (size_t) f * frame_len + kb0 + (unsigned) (tid & 15);
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v[a] = src[(size_t) (m + FT::Tn * a) * n2];
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v, w[0]);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, false> (v, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v, w[0], w[FT::Q > 1 ? 1 : 0], w[FT::Q > 2 ? 2 : 0], w[FT::Q > 2 ? 3 : 0]);
        CFFT_PHASE_FENCE();
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        float2* __restrict__ dst = out + (size_t) f * out_stride + kb0 + ((unsigned) ts & 15u);
        const unsigned ms = (unsigned) ts >> 4;
#pragma unroll
        for (int q = 0; q < FT::Q; ++q)
#pragma unroll
            for (int b = 0; b < FT::R1; ++b)
            {
                float2 y = v[q * FT::R1 + b];
                if constexpr (MULH)
                {
                    const float2 h = H[(size_t) (ms + FT::Tn * q + 32 * b) * n2 + kb0 + ((unsigned) ts & 15u)];
                    y = cmul (y, make_float2 (h.x * scaling, h.y * scaling));
                }
                gstore (&dst[(size_t) (ms + FT::Tn * q + 32 * b) * n2], SWAP_OUT ? make_float2 (y.y, y.x) : y);
            }
    }
}

// =====================================================================================================
// N = 2^21 = 2 L, L = 1024 x 1024 (round 4): one radix-2 decimation in time AROUND the register-direct two-pass transform of
// the two half-length sub-frames x_p[j] = x[2 j + p] -- no third sweep, no 2048-point tile (which these kernels cannot hold):
//   pass A  one 16-byte load brings element j of BOTH sub-frames (adjacent in the frame: 256 contiguous bytes per half-wave row
//           instead of 128); the two column tiles are transformed one after the other through the same LDS plane and stored as
//           two consecutive intermediates W_p[a][kb] of L points;
//   pass B  the row tiles of both intermediates -> E[k], O[k] in the same slots of one work-item, then
//           X[k] = E[k] + W_N^k O[k],  X[k + L] = E[k] - W_N^k O[k]  -- two plain 128-byte runs per row, natural order.
//           W_N^k for k = n2 (ms + 32 b) + kb:  (W_N^kb W_2048^ms) W_64^b = one table product per tile times a constant.
// Before: 2048 x 1024 on the LDS-staged generic tiles, 0.18 of 8 TB/s.  Reference: the same radix-2 split is the first pass of
// cfftf1_ps for any even N (simd/chowdsp_fft_impl_sse.cpp:372-432, passf2_ps :98-137).
// =====================================================================================================
// W_64^b = exp (-2 pi i b / 64), b < 32 (rounded from double)
static __device__ const float kW64re[32] = { 1.0f, 0.9951847266721969f, 0.9807852804032304f, 0.9569403357322088f, 0.9238795325112867f, 0.881921264348355f, 0.8314696123025452f, 0.773010453362737f, 0.7071067811865476f, 0.6343932841636455f, 0.5555702330196023f, 0.4713967368259978f, 0.38268343236508984f, 0.29028467725446233f, 0.19509032201612833f, 0.09801714032956077f, 0.0f, -0.09801714032956065f, -0.1950903220161282f, -0.29028467725446216f, -0.3826834323650897f, -0.4713967368259977f, -0.555570233019602f, -0.6343932841636454f, -0.7071067811865475f, -0.773010453362737f, -0.8314696123025453f, -0.8819212643483549f, -0.9238795325112867f, -0.9569403357322088f, -0.9807852804032304f, -0.9951847266721968f };
static __device__ const float kW64im[32] = { -0.0f, -0.0980171403295606f, -0.19509032201612825f, -0.29028467725446233f, -0.3826834323650898f, -0.47139673682599764f, -0.5555702330196022f, -0.6343932841636455f, -0.7071067811865475f, -0.773010453362737f, -0.8314696123025452f, -0.8819212643483549f, -0.9238795325112867f, -0.9569403357322089f, -0.9807852804032304f, -0.9951847266721968f, -1.0f, -0.9951847266721969f, -0.9807852804032304f, -0.9569403357322089f, -0.9238795325112867f, -0.881921264348355f, -0.8314696123025455f, -0.7730104533627371f, -0.7071067811865476f, -0.6343932841636455f, -0.5555702330196022f, -0.47139673682599786f, -0.3826834323650899f, -0.2902846772544624f, -0.1950903220161286f, -0.09801714032956083f };

template <int N2, bool SWAP_IN>
__global__ __launch_bounds__ (FastTile<N2>::threads, FastTile<N2>::min_waves) void pass_a_fast_r2_kernel (const float4* __restrict__ in, // the frame as L pairs (x[2 j], x[2 j + 1])
                                                                              float2* __restrict__ work, // per frame: W_0[a][kb], W_1[a][kb]
                                                                              const float2* __restrict__ stage_tw, // [rows][32] of the n2 core
                                                                              const float2* __restrict__ big_tw,   // [a][kb] = W_L^{a kb}
                                                                              unsigned n1,
                                                                              unsigned batch,
                                                                              size_t in_stride4)
{
    using FT = FastTile<N2>;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[10];
    ft_load_twiddles<FT> (w, stage_tw, tid0 & 31);
    const unsigned tiles = n1 / 16;
    const size_t sub_len = (size_t) n1 * N2;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t wk = blockIdx.x; wk < nwork; wk += gridDim.x)
    {
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned a0 = (unsigned) (wk % tiles) * 16u;
        const float4* __restrict__ src = in + (size_t) f * in_stride4 + a0 + (unsigned) (tid & 15);
        const unsigned m = (unsigned) tid >> 4;
        float2 v0[32], v1[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
        {
            const float4 q = gload4 (&src[(size_t) (m + FT::Tn * a) * n1]);
            v0[a] = SWAP_IN ? make_float2 (q.y, q.x) : make_float2 (q.x, q.y);
            v1[a] = SWAP_IN ? make_float2 (q.w, q.z) : make_float2 (q.z, q.w);
        }
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        const unsigned i = (unsigned) ts & 31u;
        // the inter-pass twiddles are the same for both sub-frames (same column a, same kb): loaded once, behind the first
        // sub-frame's arithmetic
        float2 bt[32];
#define CFFT_R2_PASS_A(V, P)                                                                                              \
    {                                                                                                                     \
        CFFT_PHASE_FENCE();                                                                                               \
        fft_reg32<false> (V, w);                                                                                          \
        CFFT_PHASE_FENCE();                                                                                               \
        ft_exchange<FT, true> (V, ft_plane, tid);                                                                         \
        CFFT_PHASE_FENCE();                                                                                               \
        if ((P) == 0)                                                                                                     \
        {                                                                                                                 \
            _Pragma ("unroll") for (int q = 0; q < FT::Q; ++q)                                                            \
            {                                                                                                             \
                const unsigned a = a0 + ((unsigned) ts >> 5) + (unsigned) (FT::Tn / 2) * q;                               \
                const float2* __restrict__ twp = big_tw + (size_t) a * N2 + i;                                            \
                _Pragma ("unroll") for (int b = 0; b < FT::R1; ++b)                                                       \
                    bt[q * FT::R1 + b] = twp[32 * b];                                                                     \
            }                                                                                                             \
        }                                                                                                                 \
        CFFT_PHASE_FENCE();                                                                                               \
        ft_stage1_all<FT> (V, w, w, w, w);                                                                                \
        CFFT_PHASE_FENCE();                                                                                               \
        _Pragma ("unroll") for (int q = 0; q < FT::Q; ++q)                                                                \
        {                                                                                                                 \
            const unsigned a = a0 + ((unsigned) ts >> 5) + (unsigned) (FT::Tn / 2) * q;                                   \
            float2* __restrict__ dst = work + ((size_t) f * 2 + (P)) * sub_len + (size_t) a * N2 + i;                     \
            _Pragma ("unroll") for (int b = 0; b < FT::R1; ++b)                                                           \
                dst[32 * b] = cmul (V[q * FT::R1 + b], bt[q * FT::R1 + b]); /* plain store: the workspace should stay cached */ \
        }                                                                                                                 \
    }
        CFFT_R2_PASS_A (v0, 0)
        CFFT_R2_PASS_A (v1, 1)
#undef CFFT_R2_PASS_A
    }
}

// MULH (fft_convolve_frames_batch, forward transform only): every bin times H[bin] * scaling on its way out, as in pass_b_fast_kernel
template <int N1, bool SWAP_OUT, bool MULH = false>
__global__ __launch_bounds__ (FastTile<N1>::threads, FastTile<N1>::min_waves) void pass_b_fast_r2_kernel (const float2* __restrict__ work,
                                                                              float2* __restrict__ out,
                                                                              const float2* __restrict__ stage_tw, // [rows][32] of the n1 core
                                                                              const float2* __restrict__ r2_tw,    // W_N^kb, kb < n2; then W_(2 N1)^j, j < 32
                                                                              unsigned n2,
                                                                              unsigned batch,
                                                                              size_t out_stride,
                                                                              const float2* __restrict__ H = nullptr,
                                                                              float scaling = 1.f)
{
    using FT = FastTile<N1>;
    static_assert (N1 == 1024 && FT::Q == 1 && FT::Tn == 32, "the radix-2 twiddle is factorised for 1024-point rows: W_N^(n2 (ms + 32 b)) = W_2048^ms W_64^b");
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[10];
    ft_load_twiddles<FT> (w, stage_tw, tid0 >> 4); // butterfly index i = member
    const float2 wms = r2_tw[n2 + ((unsigned) tid0 >> 4)]; // W_2048^ms
    const unsigned tiles = n2 / 16;
    const size_t sub_len = (size_t) N1 * n2;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t wk = blockIdx.x; wk < nwork; wk += gridDim.x)
    {
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned kb0 = (unsigned) (wk % tiles) * 16u;
        const unsigned m = (unsigned) tid >> 4;
        const float2* __restrict__ src = work + (size_t) f * 2 * sub_len + kb0 + (unsigned) (tid & 15);
        float2 v0[32], v1[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v0[a] = src[(size_t) (m + FT::Tn * a) * n2];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v1[a] = src[sub_len + (size_t) (m + FT::Tn * a) * n2];
        const float2 wkb = r2_tw[kb0 + ((unsigned) tid & 15u)]; // W_N^kb
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v0, w);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, false> (v0, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v0, w, w, w, w);
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v1, w);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, false> (v1, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v1, w, w, w, w);
        CFFT_PHASE_FENCE();
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        float2* __restrict__ dst = out + (size_t) f * out_stride + kb0 + ((unsigned) ts & 15u);
        const unsigned ms = (unsigned) ts >> 4;
        const float2 wbase = cmul (wkb, wms); // W_N^(kb + n2 ms)
#pragma unroll
        for (int b = 0; b < 32; ++b)
        {
            const float2 wk2 = cmul (wbase, make_float2 (kW64re[b], kW64im[b])); // W_N^k, k = n2 (ms + 32 b) + kb
            const float2 o = cmul (v1[b], wk2);
            float2 x0 = make_float2 (v0[b].x + o.x, v0[b].y + o.y), x1 = make_float2 (v0[b].x - o.x, v0[b].y - o.y);
            if constexpr (MULH)
            {
                const float2* __restrict__ hp = &H[(size_t) (ms + 32 * b) * n2 + kb0 + ((unsigned) ts & 15u)];
                const float2 h0 = hp[0], h1 = hp[sub_len];
                x0 = cmul (x0, make_float2 (h0.x * scaling, h0.y * scaling));
                x1 = cmul (x1, make_float2 (h1.x * scaling, h1.y * scaling));
            }
            float2* __restrict__ d = &dst[(size_t) (ms + 32 * b) * n2];
            gstore (d, SWAP_OUT ? make_float2 (x0.y, x0.x) : x0);
            gstore (d + sub_len, SWAP_OUT ? make_float2 (x1.y, x1.x) : x1);
        }
    }
}

// =====================================================================================================
// Long REAL frames (N = R1 x R2 = 2^17 .. 2^20 samples) on the same register-direct tiles: the real four-step of
// multipass_kernels.hpp (pairs of real columns as complex columns, separation, rows kb <= R2/2 only, packed spectrum
// written directly) with the column / row transforms done as above.
//   forward : pass A (columns, re-mapped across the exchange) -> Wf[a][kb], NRp = R2/2 + 16 entries per real column a
//             pass B (rows) -> packed spectrum
//   backward: pass B (rows, re-mapped) -> Wi[kb][a];  pass A (columns) -> the real frame
// The partner Z_c[R2 - kb] of the separation sits in the SAME 32-lane group as Z_c[kb] after the re-mapping (butterfly index
// 32 - i, slot R1c - 1 - b): one ds_bpermute per component, no LDS traffic, no barrier.
// =====================================================================================================
// (Round-4 A/B, not kept: TWO frames per work item sharing every inter-pass twiddle load, as the radix-2 plans do with their two sub-frames --
// twice the registers, half the workgroups per CU, no gain: profiles/r04_radix2_wrap.txt section 5.)
template <int N2>
__global__ __launch_bounds__ (FastTile<N2>::threads, FastTile<N2>::min_waves) void pass_a_real_fast_kernel (const float2* __restrict__ in, // real frame as N/2 float2
                                                                                                       float2* __restrict__ work, // Wf[a][kb]
                                                                                                       const float2* __restrict__ stage_tw,
                                                                                                       const float2* __restrict__ big_tw, // [a][kb < NRp]
                                                                                                       unsigned n1c, // complex columns = R1 / 2
                                                                                                       unsigned batch,
                                                                                                       size_t in_stride)
{
    using FT = FastTile<N2>;
    constexpr unsigned NRp = N2 / 2 + 16;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[10];
    ft_load_twiddles<FT> (w, stage_tw, tid0 & 31);
    const unsigned tiles = n1c / 16;
    const size_t ws_len = (size_t) (2 * n1c) * NRp;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t wk = blockIdx.x; wk < nwork; wk += gridDim.x)
    {
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned a0 = (unsigned) (wk % tiles) * 16u;
        const float2* __restrict__ src = in + (size_t) f * in_stride + a0 + (unsigned) (tid & 15);
        const unsigned m = (unsigned) tid >> 4;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v[a] = gload (&src[(size_t) (m + FT::Tn * a) * n1c]);
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v, w);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, true> (v, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v, w, w, w, w);
        CFFT_PHASE_FENCE();
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        const unsigned i = (unsigned) ts & 31u;
        const int partner = (int) ((32u - i) & 31u);
        // Steps (q, b) in groups of four, the two twiddles of a group requested one group ahead of its stores: gfx9 counts loads
        // and stores in one in-order counter, so a twiddle load issued after a store and waited for at once drains that store
        // (tools/isa_vmem_scan.py; it cost the N2 = 1024 form a store round trip per element).
        constexpr int NB = FT::R1 / 2 + 1, STEPS = FT::Q * NB, GROUP = CFFT_PARF_GROUP (N2), NG = (STEPS + GROUP - 1) / GROUP;
        float2* __restrict__ dst = work + (size_t) f * ws_len;
        const unsigned cc0 = a0 + ((unsigned) ts >> 5);
        float2 te[2][GROUP], to[2][GROUP];
#define CFFT_PARF_LOAD(G, SLOT)                                                                                          \
    _Pragma ("unroll") for (int c = 0; c < GROUP; ++c)                                                                   \
    {                                                                                                                    \
        constexpr int dummy_ = 0;                                                                                        \
        const int st_ = (G) * GROUP + c + dummy_;                                                                        \
        if (st_ < STEPS)                                                                                                 \
        {                                                                                                                \
            const int q_ = st_ / NB, b_ = st_ % NB;                                                                      \
            const size_t re_ = (size_t) (2 * (cc0 + (unsigned) (FT::Tn / 2) * q_)) * NRp + i;                            \
            const size_t off_ = (b_ < FT::R1 / 2 || i == 0) ? (size_t) (32 * b_) : 0; /* lanes that do not store read slot 0 */ \
            te[SLOT][c] = big_tw[re_ + off_];                                                                            \
            to[SLOT][c] = big_tw[re_ + NRp + off_];                                                                      \
        }                                                                                                                \
    }
        CFFT_PARF_LOAD (0, 0)
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
            if (g + 1 < NG)
            {
                if ((g & 1) == 0)
                    CFFT_PARF_LOAD (g + 1, 1)
                else
                    CFFT_PARF_LOAD (g + 1, 0)
            }
            CFFT_PHASE_FENCE();
#pragma unroll
            for (int c = 0; c < GROUP; ++c)
            {
                const int st = g * GROUP + c;
                if (st < STEPS)
                {
                    const int q = st / NB, b = st % NB;
                    const size_t row_e = (size_t) (2 * (cc0 + (unsigned) (FT::Tn / 2) * q)) * NRp + i, row_o = row_e + NRp;
                    // kb = i + 32 b <= R2 / 2: b < R1c / 2 for every lane, b = R1c / 2 for i = 0 only
                    const float2 zk = v[q * FT::R1 + (b % FT::R1)];
                    const float2 own = v[q * FT::R1 + ((FT::R1 - b) % FT::R1)];         // lane 0: Z[R2 - 32 b] is its own slot R1c - b
                    const float2 oth = v[q * FT::R1 + ((2 * FT::R1 - 1 - b) % FT::R1)]; // others: slot R1c - 1 - b of lane 32 - i
                    float2 zm = make_float2 (__shfl (oth.x, partner, 32), __shfl (oth.y, partner, 32));
                    if (i == 0)
                        zm = own;
                    if (b < FT::R1 / 2 || i == 0)
                    {
                        const float2 ce = make_float2 (0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                        const float2 co = make_float2 (0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                        dst[row_e + 32 * b] = cmul (ce, te[g & 1][c]);
                        dst[row_o + 32 * b] = cmul (co, to[g & 1][c]);
                    }
                }
            }
            CFFT_PHASE_FENCE();
        }
#undef CFFT_PARF_LOAD
    }
}

// MULH: every packed bin times H[bin] * scaling on its way out (bin 0 = DC / Nyquist: two real products, reference
// simd/chowdsp_fft_impl_sse.cpp:1640-1644) -- fft_convolve_frames_batch of these sizes, as pass_b_fast_kernel<..., MULH>
template <int R1_, bool MULH = false>
__global__ __launch_bounds__ (FastTile<R1_>::threads, FastTile<R1_>::min_waves) void pass_b_real_fast_kernel (const float2* __restrict__ work, // Wf[a][kb]
                                                                                                         float2* __restrict__ out, // packed spectrum
                                                                                                         const float2* __restrict__ stage_tw,
                                                                                                         unsigned n2, // R2
                                                                                                         unsigned batch,
                                                                                                         size_t out_stride,
                                                                                                         const float2* __restrict__ H = nullptr,
                                                                                                         float scaling = 1.f)
{
    using FT = FastTile<R1_>;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[FT::Q][10];
#pragma unroll
    for (int q = 0; q < FT::Q; ++q)
        ft_load_twiddles<FT> (w[q], stage_tw, (tid0 >> 4) + FT::Tn * q);
    const unsigned NR = n2 / 2 + 1, NRp = n2 / 2 + 16;
    const unsigned tiles = NRp / 16;
    const size_t ws_len = (size_t) R1_ * NRp;
    const size_t nwork = (size_t) batch * tiles;
    const size_t nloop = ((nwork + 7) / 8) * 8;
    for (size_t wv = blockIdx.x; wv < nloop; wv += gridDim.x)
    {
        const size_t wk = xcd_order (wv, nwork);
        if (wk >= nwork)
            continue;
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned kb = (unsigned) (wk % tiles) * 16u + ((unsigned) tid & 15u);
        const bool row_ok = kb < NR;
        const unsigned m = (unsigned) tid >> 4;
        const float2* __restrict__ src = work + (size_t) f * ws_len + kb;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v[a] = row_ok ? src[(size_t) (m + FT::Tn * a) * NRp] : make_float2 (0.f, 0.f);
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v, w[0]);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, false> (v, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v, w[0], w[FT::Q > 1 ? 1 : 0], w[FT::Q > 2 ? 2 : 0], w[FT::Q > 2 ? 3 : 0]);
        CFFT_PHASE_FENCE();
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        const unsigned ms = (unsigned) ts >> 4;
        float2* __restrict__ dst = out + (size_t) f * out_stride;
        if (row_ok)
        {
            const bool inner = kb != 0 && 2 * kb != n2; // rows 0 and R2/2 are their own mirrors
#pragma unroll
            for (int q = 0; q < FT::Q; ++q)
#pragma unroll
                for (int b = 0; b < FT::R1; ++b)
                {
                    const unsigned ka = ms + (unsigned) (FT::Tn * q + 32 * b);
                    const float2 y = v[q * FT::R1 + b];
                    if (32 * b + FT::Tn * q < R1_ / 2) // ka < R1 / 2 for every member (Tn q + 32 b is a multiple of Tn; ms < Tn)
                    {
                        if (q == 0 && b == 0 && kb == 0 && ms == 0)
                        {
                            float2 p0 = make_float2 (y.x, v[FT::R1 / 2].x); // (X0, X_{N/2}): Y_0[0] and Y_0[R1/2], both real, both here
                            if constexpr (MULH)
                                p0 = make_float2 (p0.x * H[0].x * scaling, p0.y * H[0].y * scaling);
                            gstore (&dst[0], p0);
                        }
                        else
                        {
                            const size_t k = (size_t) ka * n2 + kb;
                            if constexpr (MULH)
                                gstore (&dst[k], cmul (y, make_float2 (H[k].x * scaling, H[k].y * scaling)));
                            else
                                gstore (&dst[k], y);
                        }
                    }
                    else if (inner)
                    {
                        const size_t k = (size_t) (R1_ - 1 - ka) * n2 + (n2 - kb);
                        if constexpr (MULH)
                            real4_mirror_store (&dst[k], cmul (make_float2 (y.x, -y.y), make_float2 (H[k].x * scaling, H[k].y * scaling)));
                        else
                            real4_mirror_store (&dst[k], make_float2 (y.x, -y.y));
                    }
                }
        }
    }
}

// backward, first sweep: rows gathered from the packed spectrum (conjugate mirror for ka >= R1/2), transformed backward,
// times conj W_N^{a kb}, into Wi[kb][a] (work-items re-mapped across the exchange: lanes along a)
template <int R1_>
__global__ __launch_bounds__ (FastTile<R1_>::threads, FastTile<R1_>::min_waves) void pass_b_real_inv_fast_kernel (const float2* __restrict__ in, // packed spectrum
                                                                                                             float2* __restrict__ work, // Wi[kb][a]
                                                                                                             const float2* __restrict__ stage_tw,
                                                                                                             const float2* __restrict__ big_tw, // [kb][a]
                                                                                                             unsigned n2,
                                                                                                             unsigned batch,
                                                                                                             size_t in_stride)
{
    using FT = FastTile<R1_>;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[10];
    ft_load_twiddles<FT> (w, stage_tw, tid0 & 31);
    const unsigned NR = n2 / 2 + 1, NRp = n2 / 2 + 16;
    const unsigned tiles = NRp / 16;
    const size_t ws_len = (size_t) R1_ * NRp;
    const size_t nwork = (size_t) batch * tiles;
    const size_t nloop = ((nwork + 7) / 8) * 8;
    for (size_t wv = blockIdx.x; wv < nloop; wv += gridDim.x)
    {
        const size_t wk = xcd_order (wv, nwork);
        if (wk >= nwork)
            continue;
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned kb0 = (unsigned) (wk % tiles) * 16u;
        const unsigned kb = kb0 + ((unsigned) tid & 15u);
        const bool row_ok = kb < NR;
        const unsigned m = (unsigned) tid >> 4;
        const float2* __restrict__ src = in + (size_t) f * in_stride;
        // Every load is issued unconditionally (rows past NR read row 0 and are never stored): a load inside a divergent
        // branch is followed by its own s_waitcnt, which serialises the 32 round trips of a tile.
        const unsigned kbl = row_ok ? kb : 0u;
        const bool first = (kbl == 0u) && (m == 0u); // holds X0 / X_{N/2} of the packed layout
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
        {
            const unsigned ka = m + (unsigned) (FT::Tn * a);
            float2 x;
            if (FT::Tn * a < R1_ / 2) // ka < R1 / 2: P[n2 ka + kb]
            {
                x = gload (&src[(size_t) ka * n2 + kbl]);
                if (a == 0)
                    x.y = first ? 0.f : x.y; // X0 (the slot's second float is X_{N/2})
            }
            else // conj P[n2 (R1 - 1 - ka) + n2 - kb]; X_{N/2} for (ka, kb) = (R1 / 2, 0)
            {
                const bool nyq = (FT::Tn * a == R1_ / 2) && first;
                const size_t mi = nyq ? (size_t) 0 : (size_t) (R1_ - ka) * n2 - kbl;
                // (round 4) DEFAULT cache policy, not nontemporal: the 16 mirrored bins of a tile straddle two 128-byte lines, each
                // of which the neighbouring tile -- same XCD, about the same time (xcd_order) -- needs too.  Nontemporal loads
                // left nothing in L2 for it: FETCH_SIZE of this kernel was 1.4 x (R1 = 512) / 2.2 x (R1 = 1024) its input bytes
                // (profiles/r04_real_four_step_traffic.txt).  CFFT_REAL4_MIRROR_NT=1 brings the old form back.
                const float2 p = CFFT_REAL4_MIRROR_NT ? gload (&src[mi]) : src[mi];
                x = make_float2 (nyq ? p.y : p.x, nyq ? 0.f : -p.y);
            }
            v[a] = make_float2 (x.y, x.x); // swapped: backward = swap . forward . swap
        }
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v, w);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, true> (v, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v, w, w, w, w);
        CFFT_PHASE_FENCE();
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        const unsigned i = (unsigned) ts & 31u;
#pragma unroll
        for (int q = 0; q < FT::Q; ++q)
        {
            const unsigned row = kb0 + ((unsigned) ts >> 5) + (unsigned) (FT::Tn / 2) * q;
            if (row < NR)
            {
                const size_t base = (size_t) row * R1_ + i;
                float2* __restrict__ dst = work + (size_t) f * ws_len + base;
                const float2* __restrict__ twp = big_tw + base;
#pragma unroll
                for (int b = 0; b < FT::R1; ++b)
                {
                    const float2 ys = v[q * FT::R1 + b];
                    dst[32 * b] = cmulc (make_float2 (ys.y, ys.x), twp[32 * b]); // un-swap, times conj W_N^{a kb}
                }
            }
        }
    }
}

// backward, second sweep: columns.  Z_c[kb] = C_2c[kb] + i C_2c+1[kb] (kb <= R2/2), conj C_2c[R2-kb] + i conj C_2c+1[R2-kb] above.
template <int N2>
__global__ __launch_bounds__ (FastTile<N2>::threads, FastTile<N2>::min_waves) void pass_a_real_inv_fast_kernel (const float2* __restrict__ work, // Wi[kb][a]
                                                                                                           float2* __restrict__ out, // real frame as N/2 float2
                                                                                                           const float2* __restrict__ stage_tw,
                                                                                                           unsigned n1c,
                                                                                                           unsigned batch,
                                                                                                           size_t out_stride)
{
    using FT = FastTile<N2>;
    constexpr unsigned NRp = N2 / 2 + 16;
    extern __shared__ __attribute__ ((aligned (16))) float ft_plane[];
    const int tid0 = threadIdx.x;
    float2 w[FT::Q][10];
#pragma unroll
    for (int q = 0; q < FT::Q; ++q)
        ft_load_twiddles<FT> (w[q], stage_tw, (tid0 >> 4) + FT::Tn * q);
    const unsigned tiles = n1c / 16;
    const size_t ws_len = (size_t) (2 * n1c) * NRp;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t wk = blockIdx.x; wk < nwork; wk += gridDim.x)
    {
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        const unsigned f = (unsigned) (wk / tiles);
        const unsigned a0 = (unsigned) (wk % tiles) * 16u;
        const unsigned m = (unsigned) tid >> 4;
        // (C_2c, C_2c+1) of row kb: one float4 at Wi[kb][2 (a0 + c)]
        const float4* __restrict__ src = reinterpret_cast<const float4*> (work + (size_t) f * ws_len) + a0 + ((unsigned) tid & 15u);
        float2 v[32];
        // two batches of 16 float4 loads, each issued as a group before any of its values is used (left alone, the scheduler
        // trades the loads' 128 registers for one wait per load: 32 serial round trips per tile)
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            float4 p[16];
#pragma unroll
            for (int a = 0; a < 16; ++a)
            {
                const unsigned kb = m + (unsigned) (FT::Tn * (a + 16 * h));
                const bool upper = (unsigned) (FT::Tn * (a + 16 * h)) > (unsigned) (N2 / 2) || ((unsigned) (FT::Tn * (a + 16 * h)) == (unsigned) (N2 / 2) && m != 0);
                const unsigned row = upper ? N2 - kb : kb;
                p[a] = src[(size_t) row * n1c];
            }
            CFFT_PHASE_FENCE();
#pragma unroll
            for (int a = 0; a < 16; ++a)
            {
                const bool upper = (unsigned) (FT::Tn * (a + 16 * h)) > (unsigned) (N2 / 2) || ((unsigned) (FT::Tn * (a + 16 * h)) == (unsigned) (N2 / 2) && m != 0);
                const float2 z = upper ? make_float2 (p[a].x + p[a].w, p[a].z - p[a].y) : make_float2 (p[a].x - p[a].w, p[a].y + p[a].z);
                v[a + 16 * h] = make_float2 (z.y, z.x); // swapped for the backward transform
            }
            CFFT_PHASE_FENCE();
        }
        CFFT_PHASE_FENCE();
        fft_reg32<false> (v, w[0]);
        CFFT_PHASE_FENCE();
        ft_exchange<FT, false> (v, ft_plane, tid);
        CFFT_PHASE_FENCE();
        ft_stage1_all<FT> (v, w[0], w[FT::Q > 1 ? 1 : 0], w[FT::Q > 2 ? 2 : 0], w[FT::Q > 2 ? 3 : 0]);
        CFFT_PHASE_FENCE();
        int ts = tid0;
        CFFT_LAUNDER_INDEX (ts);
        float2* __restrict__ dst = out + (size_t) f * out_stride + a0 + ((unsigned) ts & 15u);
        const unsigned ms = (unsigned) ts >> 4;
#pragma unroll
        for (int q = 0; q < FT::Q; ++q)
#pragma unroll
            for (int b = 0; b < FT::R1; ++b)
            {
                const float2 y = v[q * FT::R1 + b];
                gstore (&dst[(size_t) (ms + FT::Tn * q + 32 * b) * n1c], make_float2 (y.y, y.x));
            }
    }
}
} // namespace cfft
