// Mixed-radix (2, 3, 4, 5, 8) frames that are not powers of two: N = 2^a * 3^b * 5^c, the rest of the reference's
// accepted size set (simd/chowdsp_fft_impl_common.hpp:51-82, 216-225; exercised by the reference's "Other sizes"
// tests, test/test.cpp:279-285: 96, 192, 384, 480, 640, 768, 9216).  SURVEY section 8 row f2.
//
// Replaces passf3_ps / passf5_ps (simd/chowdsp_fft_impl_sse.cpp:182-219, 298-370) and, through the same
// half-length-complex + pair-stage scheme as real_kernels.hpp, radf3/5 and radb3/5 (:588-625, :704-786, :1126-1170,
// :1262-1356).  A workgroup owns F whole frames at a time (F > 1 for short frames so that every work-item has a
// butterfly to do); the frames ping-pong between two LDS images, one Stockham stage per radix with a RUN-TIME radix
// list (these sizes are an open set, so there is no per-size template plan); per-stage twiddles come from a table in
// L2.  Long frames (two images > 64 KiB) run with 1024 work-items per workgroup because only one workgroup fits a CU.
// This path favours generality over the last factor of two in bandwidth: the power-of-two sizes -- the hot path --
// never take it.
#pragma once
#include "real_kernels.hpp"

#include <type_traits>

namespace cfft
{
static constexpr int kGenericMaxStages = 10;
static constexpr int kGenericMaxN = 20000; // one LDS image of 8*N bytes (in-place stages, see below) within the CU's 160 KiB

struct GenericPlan
{
    int n;                            // complex points per frame
    int ns;                           // number of stages
    int frames;                       // frames per workgroup pass (F)
    unsigned in_stride, out_stride;   // float2 between consecutive frames in global memory (n for packed batches)
    int radix[kGenericMaxStages];     // each in {2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16}
    int tw_offset[kGenericMaxStages]; // float2 offset of the stage's table: (radix-1) * p entries, [a-1][k]
    int pair_lds;                     // generic_kernel, real modes: the pair twiddles ride in LDS too (behind the stage twiddles)
    int lds_stages;                   // generic_inplace_kernel: the tables of stages [0, lds_stages) ride in LDS behind the image
    int lds_tw_len;                   // ... that many float2 (the tables are contiguous from tw_offset[0])
};

template <>
__device__ __forceinline__ void fft_reg<3> (float2 (&u)[3])
{
    constexpr float s60 = 0.86602540378443864676f;
    const float2 t = cadd (u[1], u[2]);
    const float2 d = csub (u[1], u[2]);
    const float2 m = make_float2 (u[0].x - 0.5f * t.x, u[0].y - 0.5f * t.y);
    const float2 rot = make_float2 (s60 * d.y, -s60 * d.x); // -i * s60 * d
    u[0] = cadd (u[0], t);
    u[1] = cadd (m, rot);
    u[2] = csub (m, rot);
}

template <>
__device__ __forceinline__ void fft_reg<5> (float2 (&u)[5])
{
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 a1 = cadd (u[1], u[4]), b1 = csub (u[1], u[4]);
    const float2 a2 = cadd (u[2], u[3]), b2 = csub (u[2], u[3]);
    const float2 m1 = make_float2 (u[0].x + c1 * a1.x + c2 * a2.x, u[0].y + c1 * a1.y + c2 * a2.y);
    const float2 m2 = make_float2 (u[0].x + c2 * a1.x + c1 * a2.x, u[0].y + c2 * a1.y + c1 * a2.y);
    // forward: X1 = m1 - i (s1 b1 + s2 b2),  X2 = m2 - i (s2 b1 - s1 b2)
    const float2 q1 = make_float2 (s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 q2 = make_float2 (s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    u[0] = make_float2 (u[0].x + a1.x + a2.x, u[0].y + a1.y + a2.y);
    u[1] = make_float2 (m1.x + q1.y, m1.y - q1.x);
    u[4] = make_float2 (m1.x - q1.y, m1.y + q1.x);
    u[2] = make_float2 (m2.x + q2.y, m2.y - q2.x);
    u[3] = make_float2 (m2.x - q2.y, m2.y + q2.x);
}

// ---- composite radices 6, 9, 10, 12, 15 (round 2) ---------------------------------------------------------------------
// One stage of radix A * B instead of a stage of radix A and a stage of radix B: one LDS round trip less per merged pair
// (N = 9216: 3 3 16 8 8 -> 12 12 8 8; 96: 3 8 4 -> 12 8; 1536: 3 8 8 8 -> 12 16 8; 480: 5 3 8 4 -> 15 8 4).  Cooley-Tukey
// inside the registers, n = n1 + A n2, k = B k1 + k2:
//     X[B k1 + k2] = sum_{n1 < A} W_A^{n1 k1} [ W_{AB}^{n1 k2} sum_{n2 < B} x[n1 + A n2] W_B^{n2 k2} ]
// with the inner twiddles W_{AB}^m as compile-time constants (evaluated in double by the constexpr series below).
namespace ct
{
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double sin_series (double x) // |x| <= pi/4
{
    const double x2 = x * x;
    double term = x, sum = x;
    for (int i = 1; i < 12; ++i)
    {
        term *= -x2 / (double) ((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double cos_series (double x) // |x| <= pi/4
{
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 12; ++i)
    {
        term *= -x2 / (double) ((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
// cos / sin of 2 pi num / den by octant reduction (exact symmetries, series argument <= pi/4)
constexpr double cos_2pi (int num, int den)
{
    num %= den;
    if (num < 0)
        num += den;
    if (2 * num > den) // cos(2 pi - t) = cos t
        num = den - num;
    if (4 * num > den) // cos(pi - t) = -cos t
        return -cos_2pi (den - 2 * num, 2 * den);
    // 0 <= 2 pi num / den <= pi/2
    if (8 * num > den) // cos t = sin(pi/2 - t)
        return sin_series (kPi / 2.0 - 2.0 * kPi * (double) num / (double) den);
    return cos_series (2.0 * kPi * (double) num / (double) den);
}
constexpr double sin_2pi (int num, int den) { return cos_2pi (4 * num - den, 4 * den); } // sin t = cos(t - pi/2)
template <int DEN, int M>
__device__ __forceinline__ float2 mul_root (float2 u) // u * exp(-2 pi i M / DEN)
{
    constexpr int m = ((M % DEN) + DEN) % DEN;
    if constexpr (m == 0)
        return u;
    else if constexpr (4 * m == DEN)
        return make_float2 (u.y, -u.x);
    else if constexpr (2 * m == DEN)
        return make_float2 (-u.x, -u.y);
    else if constexpr (4 * m == 3 * DEN)
        return make_float2 (-u.y, u.x);
    else
    {
        constexpr float c = (float) cos_2pi (m, DEN), sn = (float) sin_2pi (m, DEN);
        return cmul (u, make_float2 (c, -sn));
    }
}
template <int A, int B, int N1, int K2>
__device__ __forceinline__ void inner_twiddle_row (float2 (&y)[A * B])
{
    if constexpr (K2 < B)
    {
        y[N1 + A * K2] = mul_root<A * B, N1 * K2> (y[N1 + A * K2]);
        inner_twiddle_row<A, B, N1, K2 + 1> (y);
    }
}
template <int A, int B, int N1>
__device__ __forceinline__ void inner_twiddles (float2 (&y)[A * B])
{
    if constexpr (N1 < A)
    {
        inner_twiddle_row<A, B, N1, 1> (y);
        inner_twiddles<A, B, N1 + 1> (y);
    }
}
// natural order in and out
template <int A, int B>
__device__ __forceinline__ void fft_reg_ct (float2 (&u)[A * B])
{
    // B-point transforms over n2 for each n1: y[n1 + A k2]
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1)
    {
        float2 t[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2)
            t[n2] = u[n1 + A * n2];
        fft_reg<B> (t);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2)
            u[n1 + A * k2] = t[k2];
    }
    inner_twiddles<A, B, 1> (u);
    // A-point transforms over n1 for each k2: X[B k1 + k2]
    float2 o[A * B];
#pragma unroll
    for (int k2 = 0; k2 < B; ++k2)
    {
        float2 t[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1)
            t[n1] = u[n1 + A * k2];
        fft_reg<A> (t);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1)
            o[B * k1 + k2] = t[k1];
    }
#pragma unroll
    for (int i = 0; i < A * B; ++i)
        u[i] = o[i];
}
} // namespace ct
template <> __device__ __forceinline__ void fft_reg<6> (float2 (&u)[6]) { ct::fft_reg_ct<3, 2> (u); }
template <> __device__ __forceinline__ void fft_reg<9> (float2 (&u)[9]) { ct::fft_reg_ct<3, 3> (u); }
template <> __device__ __forceinline__ void fft_reg<10> (float2 (&u)[10]) { ct::fft_reg_ct<5, 2> (u); }
template <> __device__ __forceinline__ void fft_reg<12> (float2 (&u)[12]) { ct::fft_reg_ct<3, 4> (u); }
template <> __device__ __forceinline__ void fft_reg<15> (float2 (&u)[15]) { ct::fft_reg_ct<5, 3> (u); }

// x / d for 0 <= x < 2^16-ish and a run-time, workgroup-uniform d: one multiply-high with M = floor((2^32 - 1) / d) + 1 (exact while
// x * d < 2^32: here x < 65536 work items and d <= 20000).  The compiler's 32-bit division is ~35 instructions, and the stage
// loops need one or two per butterfly -- as much arithmetic as a radix-6 butterfly itself.
__device__ __forceinline__ unsigned div_magic (unsigned d) { return 0xFFFFFFFFu / d + 1u; } // d >= 2
__device__ __forceinline__ int fast_div (int x, unsigned magic) { return (int) __umulhi ((unsigned) x, magic); }

// One Stockham stage of radix R over `nframes` frames of n points: src -> dst.
// IO 0: LDS image -> LDS image (frame f at f*n).  IO 1: global memory (frame stride gs, optional re<->im swap) -> LDS:
// the first stage of a complex transform reads its operands where they are, x[i + a*m] is lane-contiguous.
// IO 2: LDS -> global memory: the last stage stores y[j + b*p] straight to the result (lane-contiguous in k).
// IO 3: global -> global (single-stage plans).
// STATIC: n and p are compile-time constants at the call (static_core below): everything derived from them folds.
// CONV (fused convolution, generic_conv_kernels.hpp): this is the forward transform's last stage and it writes the LDS image;
// every output is multiplied by H[its index] * scaling and stored re <-> im swapped, ready for the backward transform.
template <int R, int BLOCK, int IO, bool SWAP, bool STATIC = false, bool CONV = false>
__device__ __forceinline__ void generic_stage (const float2* __restrict__ src, float2* __restrict__ dst,
                                               const float2* __restrict__ tw, int n, int p, int nframes, int tid,
                                               unsigned gs_in, unsigned gs_out, unsigned lfs,
                                               const float2* __restrict__ H = nullptr, float scaling = 1.f)
{
    // lfs: distance between consecutive frames inside an LDS image (n, or n + 1 where lanes walk across frames)
    if constexpr (! STATIC)
        asm volatile ("; generic_stage R=%1 IO=%2" : "+s"(n) : "n"(R), "n"(IO)); // keeps the radix cases' offsets inside their case (see inplace_stage)
    const int m = n / R;
    const bool p_pow2 = (p & (p - 1)) == 0;
    const unsigned fs_in = (IO & 1) ? gs_in : lfs, fs_out = (IO & 2) ? gs_out : lfs;
    const unsigned magic_m = m > 1 ? div_magic ((unsigned) m) : 0u, magic_p = p > 1 ? div_magic ((unsigned) p) : 0u;
    for (int idx = tid; idx < nframes * m; idx += BLOCK)
    {
        // one frame per workgroup pass (n > 512) and power-of-two p are the common cases: no integer division there
        const int f = (nframes == 1) ? 0 : (m == 1 ? idx : fast_div (idx, magic_m));
        const int i = idx - f * m;
        const int k = p_pow2 ? (i & (p - 1)) : (i - fast_div (i, magic_p) * p);
        const int j = (i - k) * R + k;
        const float2* s = src + (size_t) f * fs_in;
        float2* d = dst + (size_t) f * fs_out;
        float2 u[R];
#pragma unroll
        for (int a = 0; a < R; ++a)
        {
            const float2 x = (IO & 1) ? gload (&s[i + a * m]) : s[i + a * m];
            u[a] = ((IO & 1) && SWAP) ? make_float2 (x.y, x.x) : x;
        }
        if (p > 1)
        {
#pragma unroll
            for (int a = 1; a < R; ++a)
                u[a] = cmul (u[a], tw[(a - 1) * p + k]);
        }
        fft_reg<R> (u);
#pragma unroll
        for (int b = 0; b < R; ++b)
        {
            if (IO & 2)
                gstore (&d[j + b * p], SWAP ? make_float2 (u[b].y, u[b].x) : u[b]);
            else if constexpr (CONV)
            {
                const float2 h = H[j + b * p];
                const float2 y = cmul (u[b], make_float2 (h.x * scaling, h.y * scaling));
                d[j + b * p] = make_float2 (y.y, y.x);
            }
            else
                d[j + b * p] = u[b];
        }
    }
}

template <int BLOCK, int IO, bool SWAP>
__device__ __forceinline__ void generic_stage_dispatch (int r, const float2* __restrict__ src, float2* __restrict__ dst,
                                                        const float2* __restrict__ tw, int n, int p, int nframes, int tid,
                                                        unsigned gs_in, unsigned gs_out, unsigned lfs)
{
    switch (r)
    {
        case 2: generic_stage<2, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 3: generic_stage<3, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 4: generic_stage<4, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 5: generic_stage<5, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 6: generic_stage<6, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 8: generic_stage<8, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 9: generic_stage<9, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 10: generic_stage<10, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 12: generic_stage<12, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        case 15: generic_stage<15, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
        default: generic_stage<16, BLOCK, IO, SWAP> (src, dst, tw, n, p, nframes, tid, gs_in, gs_out, lfs); break;
    }
}

// All stages.  IN_GLOBAL: stage 0 reads the frames from `gin` (frame stride plan.in_stride) instead of image `a`.
// OUT_GLOBAL: the last stage writes to `gout`; otherwise the image holding the result is returned.
// SWAP: re<->im swap on the global side(s) (inverse transform).
template <int BLOCK, bool IN_GLOBAL, bool OUT_GLOBAL, bool SWAP>
__device__ __forceinline__ float2* generic_core (const GenericPlan& plan, const float2* __restrict__ twtab, float2* a, float2* b,
                                                  const float2* __restrict__ gin, float2* __restrict__ gout, int nframes, int tid,
                                                  unsigned lfs, int tw_base = 0)
{
    // tw_base: table offset of twtab[0] (plan.tw_offset[0] when twtab is the workgroup's LDS copy of this plan's n - 1 twiddles)
    float2 *src = a, *dst = b;
    int p = 1;
    for (int s = 0; s < plan.ns; ++s)
    {
        const int r = plan.radix[s];
        const float2* tw = twtab + (plan.tw_offset[s] - tw_base);
        const bool first = IN_GLOBAL && s == 0, last = OUT_GLOBAL && s == plan.ns - 1;
        if (first && last)
            generic_stage_dispatch<BLOCK, 3, SWAP> (r, gin, gout, tw, plan.n, p, nframes, tid, plan.in_stride, plan.out_stride, lfs);
        else if (first)
            generic_stage_dispatch<BLOCK, 1, SWAP> (r, gin, dst, tw, plan.n, p, nframes, tid, plan.in_stride, plan.out_stride, lfs);
        else if (last)
            generic_stage_dispatch<BLOCK, 2, SWAP> (r, src, gout, tw, plan.n, p, nframes, tid, plan.in_stride, plan.out_stride, lfs);
        else
            generic_stage_dispatch<BLOCK, 0, false> (r, src, dst, tw, plan.n, p, nframes, tid, plan.in_stride, plan.out_stride, lfs);
        if (! last)
            __syncthreads();
        float2* t = src;
        src = dst;
        dst = t;
        p *= r;
    }
    return src;
}

// ---- compile-time stage lists for every mixed-radix length up to 2048 (round 2) --------------------------------------------
// The reference's "other sizes" (test/test.cpp:279-285: 96, 192, 384, 480, 640, 768 complex and real points, i.e. complex
// lengths 48 ... 768) and the other 2-3-5-smooth multiples of 16 that the two-image kernel serves -- thirty lengths -- get the
// stage list the planner gives them (tables.cpp; checked at launch, since the twiddle table is laid out by that list; the list
// below is generated from fft_gpu_plan_describe) as a template: no radix switch, no stage loop, no run-time division, every
// offset a constant.  Against the run-time list that is about half the vector and a fortieth of the scalar instructions
// (profiles/r02_sq_counters_mixed_radix.txt): N = 1920 0.51 -> 0.74 of 8 TB/s, N = 48 0.34 -> 0.52.  Longer frames (the
// in-place kernel) keep the run-time list.
template <int N_, int... Rs>
struct StaticPlan
{
    static constexpr int n = N_;
    static constexpr int ns = (int) sizeof...(Rs);
    static constexpr int radix (int s)
    {
        constexpr int r[] = { Rs... };
        return r[s];
    }
    static constexpr int period (int s) // product of the radices before stage s
    {
        int p = 1;
        for (int i = 0; i < s; ++i)
            p *= radix (i);
        return p;
    }
    static constexpr int tw_offset (int s) // of stage s's table inside the plan's n - 1 twiddles
    {
        int o = 0;
        for (int i = 0; i < s; ++i)
            o += (radix (i) - 1) * period (i);
        return o;
    }
};
// two-stage lists are ordered for full-line stores in the last stage where one order gives them (tables.cpp: generic_radices)
#ifndef CFFT_TWO_STAGE_FULL_LINE
#define CFFT_TWO_STAGE_FULL_LINE 1
#endif
// CFFT_GENERIC_LISTS_V2 (round 5): the one-pass lists of 160, 192 and 240 points re-chosen by measurement -- three stages of radix <= 8 instead of
// two stages of radix 10 ... 16 (12 ... 16 work-items per frame, first-stage reads in runs of 12 / 15 points): c192 +14 %, c240 +14 %, real N = 320
// +10 % (profiles/r05_generic_lists.txt; the other lengths keep their lists: the same treatment measured -2 ... -9 % on their complex transforms).
// tables.cpp (generic_radices) carries the same table.
#ifndef CFFT_GENERIC_LISTS_V2
#define CFFT_GENERIC_LISTS_V2 1
#endif
template <int N_> struct StaticPlanFor { using type = void; };
template <> struct StaticPlanFor<48> { using type = StaticPlan<48, 6, 8>; };
template <> struct StaticPlanFor<80> { using type = StaticPlan<80, 10, 8>; };
template <> struct StaticPlanFor<96> { using type = StaticPlan<96, 12, 8>; };
template <> struct StaticPlanFor<144> { using type = StaticPlan<144, 12, 12>; };
#if CFFT_GENERIC_LISTS_V2
template <> struct StaticPlanFor<160> { using type = StaticPlan<160, 4, 8, 5>; };
#else
template <> struct StaticPlanFor<160> { using type = StaticPlan<160, 10, 16>; };
#endif
#if CFFT_GENERIC_LISTS_V2
template <> struct StaticPlanFor<192> { using type = StaticPlan<192, 6, 8, 4>; };
#else
template <> struct StaticPlanFor<192> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<192, 16, 12>, StaticPlan<192, 12, 16>>::type; };
#endif
#if CFFT_GENERIC_LISTS_V2
template <> struct StaticPlanFor<240> { using type = StaticPlan<240, 6, 8, 5>; };
#else
template <> struct StaticPlanFor<240> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<240, 16, 15>, StaticPlan<240, 15, 16>>::type; };
#endif
template <> struct StaticPlanFor<288> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<288, 6, 8, 6>, StaticPlan<288, 6, 6, 8>>::type; };
template <> struct StaticPlanFor<320> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<320, 8, 8, 5>, StaticPlan<320, 5, 8, 8>>::type; };
template <> struct StaticPlanFor<384> { using type = StaticPlan<384, 6, 8, 8>; };
template <> struct StaticPlanFor<400> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<400, 10, 8, 5>, StaticPlan<400, 10, 5, 8>>::type; };
template <> struct StaticPlanFor<432> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<432, 6, 8, 9>, StaticPlan<432, 9, 6, 8>>::type; };
template <> struct StaticPlanFor<480> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<480, 10, 8, 6>, StaticPlan<480, 10, 6, 8>>::type; };
template <> struct StaticPlanFor<576> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<576, 8, 8, 9>, StaticPlan<576, 9, 8, 8>>::type; };
template <> struct StaticPlanFor<640> { using type = StaticPlan<640, 10, 8, 8>; };
template <> struct StaticPlanFor<720> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<720, 10, 8, 9>, StaticPlan<720, 10, 9, 8>>::type; };
template <> struct StaticPlanFor<768> { using type = StaticPlan<768, 12, 8, 8>; };
template <> struct StaticPlanFor<800> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<800, 10, 8, 10>, StaticPlan<800, 10, 10, 8>>::type; };
template <> struct StaticPlanFor<864> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<864, 12, 8, 9>, StaticPlan<864, 12, 9, 8>>::type; };
template <> struct StaticPlanFor<960> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<960, 12, 8, 10>, StaticPlan<960, 12, 10, 8>>::type; };
template <> struct StaticPlanFor<1152> { using type = StaticPlan<1152, 12, 12, 8>; };
template <> struct StaticPlanFor<1200> { using type = StaticPlan<1200, 12, 10, 10>; };
template <> struct StaticPlanFor<1280> { using type = StaticPlan<1280, 10, 16, 8>; };
template <> struct StaticPlanFor<1296> { using type = StaticPlan<1296, 12, 12, 9>; };
template <> struct StaticPlanFor<1440> { using type = StaticPlan<1440, 12, 12, 10>; };
template <> struct StaticPlanFor<1536> { using type = StaticPlan<1536, 12, 16, 8>; };
template <> struct StaticPlanFor<1600> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<1600, 10, 16, 10>, StaticPlan<1600, 10, 10, 16>>::type; };
template <> struct StaticPlanFor<1728> { using type = StaticPlan<1728, 12, 12, 12>; };
template <> struct StaticPlanFor<1920> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<1920, 12, 16, 10>, StaticPlan<1920, 12, 10, 16>>::type; };
template <> struct StaticPlanFor<2000> { using type = std::conditional<CFFT_TWO_STAGE_FULL_LINE != 0, StaticPlan<2000, 10, 5, 8, 5>, StaticPlan<2000, 10, 5, 5, 8>>::type; };
// lengths that only occur as the column / row length of a long plan's tile kernels (kernels_tile_static*.hip): every other
// 2-3-5-smooth length from 100 to 1125, same planner rule (tables.cpp: generic_radices; the launcher checks the list)
template <> struct StaticPlanFor<100> { using type = StaticPlan<100, 10, 10>; };
template <> struct StaticPlanFor<108> { using type = StaticPlan<108, 12, 9>; };
template <> struct StaticPlanFor<120> { using type = StaticPlan<120, 12, 10>; };
template <> struct StaticPlanFor<125> { using type = StaticPlan<125, 5, 5, 5>; };
template <> struct StaticPlanFor<128> { using type = StaticPlan<128, 16, 8>; };
template <> struct StaticPlanFor<135> { using type = StaticPlan<135, 15, 9>; };
template <> struct StaticPlanFor<150> { using type = StaticPlan<150, 15, 10>; };
template <> struct StaticPlanFor<162> { using type = StaticPlan<162, 9, 6, 3>; };
template <> struct StaticPlanFor<180> { using type = StaticPlan<180, 15, 12>; };
template <> struct StaticPlanFor<200> { using type = StaticPlan<200, 5, 5, 8>; };
template <> struct StaticPlanFor<216> { using type = StaticPlan<216, 6, 6, 6>; };
template <> struct StaticPlanFor<225> { using type = StaticPlan<225, 15, 15>; };
template <> struct StaticPlanFor<243> { using type = StaticPlan<243, 9, 9, 3>; };
template <> struct StaticPlanFor<250> { using type = StaticPlan<250, 10, 5, 5>; };
template <> struct StaticPlanFor<256> { using type = StaticPlan<256, 16, 16>; };
template <> struct StaticPlanFor<270> { using type = StaticPlan<270, 9, 6, 5>; };
template <> struct StaticPlanFor<300> { using type = StaticPlan<300, 10, 6, 5>; };
template <> struct StaticPlanFor<324> { using type = StaticPlan<324, 9, 6, 6>; };
template <> struct StaticPlanFor<360> { using type = StaticPlan<360, 10, 6, 6>; };
template <> struct StaticPlanFor<375> { using type = StaticPlan<375, 15, 5, 5>; };
template <> struct StaticPlanFor<405> { using type = StaticPlan<405, 9, 9, 5>; };
template <> struct StaticPlanFor<450> { using type = StaticPlan<450, 10, 9, 5>; };
template <> struct StaticPlanFor<486> { using type = StaticPlan<486, 9, 9, 6>; };
template <> struct StaticPlanFor<500> { using type = StaticPlan<500, 10, 10, 5>; };
template <> struct StaticPlanFor<512> { using type = StaticPlan<512, 8, 8, 8>; };
template <> struct StaticPlanFor<540> { using type = StaticPlan<540, 10, 9, 6>; };
template <> struct StaticPlanFor<600> { using type = StaticPlan<600, 10, 10, 6>; };
template <> struct StaticPlanFor<625> { using type = StaticPlan<625, 5, 5, 5, 5>; };
template <> struct StaticPlanFor<648> { using type = StaticPlan<648, 9, 9, 8>; };
template <> struct StaticPlanFor<675> { using type = StaticPlan<675, 15, 9, 5>; };
template <> struct StaticPlanFor<729> { using type = StaticPlan<729, 9, 9, 9>; };
template <> struct StaticPlanFor<750> { using type = StaticPlan<750, 15, 10, 5>; };
template <> struct StaticPlanFor<810> { using type = StaticPlan<810, 10, 9, 9>; };
template <> struct StaticPlanFor<900> { using type = StaticPlan<900, 10, 10, 9>; };
template <> struct StaticPlanFor<972> { using type = StaticPlan<972, 12, 9, 9>; };
template <> struct StaticPlanFor<1000> { using type = StaticPlan<1000, 10, 10, 10>; };
template <> struct StaticPlanFor<1024> { using type = StaticPlan<1024, 16, 8, 8>; };
template <> struct StaticPlanFor<1080> { using type = StaticPlan<1080, 12, 10, 9>; };
template <> struct StaticPlanFor<1125> { using type = StaticPlan<1125, 15, 15, 5>; };
template <> struct StaticPlanFor<2048> { using type = StaticPlan<2048, 16, 16, 8>; };
// the tile kernels' lists (stages LDS to LDS): the sorted order for the two lengths whose whole-frame kernels use the full-line order
template <int N_> struct StaticTilePlanFor { using type = typename StaticPlanFor<N_>::type; };
#if CFFT_GENERIC_LISTS_V2 // (tile kernels keep the planner's tile lists: frames_override > 0 in generic_radices)
template <> struct StaticTilePlanFor<160> { using type = StaticPlan<160, 10, 16>; };
#endif
template <> struct StaticTilePlanFor<192> { using type = StaticPlan<192, 12, 16>; };
template <> struct StaticTilePlanFor<240> { using type = StaticPlan<240, 15, 16>; };
template <> struct StaticTilePlanFor<288> { using type = StaticPlan<288, 6, 6, 8>; };
template <> struct StaticTilePlanFor<320> { using type = StaticPlan<320, 5, 8, 8>; };
template <> struct StaticTilePlanFor<400> { using type = StaticPlan<400, 10, 5, 8>; };
template <> struct StaticTilePlanFor<432> { using type = StaticPlan<432, 9, 6, 8>; };
template <> struct StaticTilePlanFor<480> { using type = StaticPlan<480, 10, 6, 8>; };
template <> struct StaticTilePlanFor<576> { using type = StaticPlan<576, 9, 8, 8>; };
template <> struct StaticTilePlanFor<720> { using type = StaticPlan<720, 10, 9, 8>; };
template <> struct StaticTilePlanFor<800> { using type = StaticPlan<800, 10, 10, 8>; };
template <> struct StaticTilePlanFor<864> { using type = StaticPlan<864, 12, 9, 8>; };
template <> struct StaticTilePlanFor<960> { using type = StaticPlan<960, 12, 10, 8>; };
template <> struct StaticTilePlanFor<1600> { using type = StaticPlan<1600, 10, 10, 16>; };
template <> struct StaticTilePlanFor<1920> { using type = StaticPlan<1920, 12, 10, 16>; };
template <> struct StaticTilePlanFor<2000> { using type = StaticPlan<2000, 10, 5, 5, 8>; };

// all stages of a static plan; twl: the workgroup's LDS copy of the plan's twiddles.  Returns the image that holds the result.
// ---- software pipeline of the compile-time lists (round 3): the first stage's operands of the NEXT group of frames are
// requested while this group's other stages run.  A first stage reads global memory, has no twiddles (p = 1) and writes
// output i R + b; its butterfly number tid (the only one a work-item has for every length but the shortest) takes its
// operands from `pre`, requested by generic_first_prefetch one group earlier; further butterflies (idx >= BLOCK) load as before.
template <int R, int BLOCK, bool SWAP>
__device__ __forceinline__ void generic_first_prefetch (float2 (&pre)[R], const float2* __restrict__ src, int n, int nframes, int tid, unsigned gs_in)
{
    const int m = n / R;
    if (tid < nframes * m)
    {
        const int f = (nframes == 1) ? 0 : (m == 1 ? tid : fast_div (tid, div_magic ((unsigned) (m > 1 ? m : 2))));
        const int i = tid - f * m;
        const float2* s = src + (size_t) f * gs_in;
#pragma unroll
        for (int a = 0; a < R; ++a)
        {
            const float2 x = gload (&s[i + a * m]);
            pre[a] = SWAP ? make_float2 (x.y, x.x) : x;
        }
    }
}
template <int R, int BLOCK, bool SWAP>
__device__ __forceinline__ void generic_first_stage_pre (const float2 (&pre)[R], const float2* __restrict__ src, float2* __restrict__ dst, int n, int nframes, int tid,
                                                         unsigned gs_in, unsigned lfs)
{
    const int m = n / R;
    const unsigned magic_m = m > 1 ? div_magic ((unsigned) m) : 0u;
    for (int idx = tid; idx < nframes * m; idx += BLOCK)
    {
        const int f = (nframes == 1) ? 0 : (m == 1 ? idx : fast_div (idx, magic_m));
        const int i = idx - f * m;
        float2 u[R];
        if (idx == tid)
        {
#pragma unroll
            for (int a = 0; a < R; ++a)
                u[a] = pre[a];
        }
        else
        {
            const float2* s = src + (size_t) f * gs_in;
#pragma unroll
            for (int a = 0; a < R; ++a)
            {
                const float2 x = gload (&s[i + a * m]);
                u[a] = SWAP ? make_float2 (x.y, x.x) : x;
            }
        }
        fft_reg<R> (u);
        float2* d = dst + (size_t) f * lfs + i * R;
#pragma unroll
        for (int b = 0; b < R; ++b)
            d[b] = u[b];
    }
}

// CONV: the last stage (which then writes an LDS image: OUT_GLOBAL false) multiplies by H * scaling and swaps, see generic_stage
// LFS: distance of consecutive frames inside an LDS image where it is not n (the tile kernels pad every frame by one point)
template <class SP, int BLOCK, bool IN_GLOBAL, bool OUT_GLOBAL, bool SWAP, int S = 0, bool CONV = false, int LFS = 0>
__device__ __forceinline__ float2* static_core (const float2* twl, float2* a, float2* b, const float2* __restrict__ gin, float2* __restrict__ gout,
                                                int nframes, int tid, unsigned gs_in, unsigned gs_out,
                                                const float2* __restrict__ H = nullptr, float scaling = 1.f)
{
    if constexpr (S == SP::ns)
        return a;
    else
    {
        constexpr int R = SP::radix (S), P = SP::period (S);
        constexpr bool first = IN_GLOBAL && S == 0, last = OUT_GLOBAL && S == SP::ns - 1;
        constexpr int IO = (first ? 1 : 0) | (last ? 2 : 0);
        constexpr bool conv_here = CONV && ! OUT_GLOBAL && S == SP::ns - 1;
        generic_stage<R, BLOCK, IO, IO != 0 && SWAP, true, conv_here> (first ? gin : a, last ? gout : b, twl + SP::tw_offset (S), SP::n, P, nframes, tid, gs_in, gs_out, (unsigned) (LFS != 0 ? LFS : SP::n), H, scaling);
        if constexpr (! last)
            __syncthreads();
        return static_core<SP, BLOCK, IN_GLOBAL, OUT_GLOBAL, SWAP, S + 1, CONV, LFS> (twl, b, a, gin, gout, nframes, tid, gs_in, gs_out, H, scaling);
    }
}

// 48 and 80 (round 3, tools/gpu_r3_staged_48_80_ab.sh, direct / staged, fraction of 8 TB/s): c48 0.538 / 0.561, c80 0.546 / 0.558,
// real N = 96 0.537 / 0.594, real N = 160 0.540 / 0.620 (round 2 had measured no difference for the complex ones)
#ifndef CFFT_GENERIC_STAGED_48_80
#define CFFT_GENERIC_STAGED_48_80 1
#endif
#ifndef CFFT_GENERIC_STAGED_160
#define CFFT_GENERIC_STAGED_160 1
#endif
#ifndef CFFT_GENERIC_STAGED_MAX_N
#define CFFT_GENERIC_STAGED_MAX_N 256
#endif
// Lengths whose complex transforms go through the staged-I/O path of generic_kernel (below), by measurement on MI355X
// (profiles/r02_generic_staged_io_ab.txt, staged against direct, GB/s): 96 5419 / 4509, 144 5169 / 4304, 160 5494 / 4556 -- but
// 48 4142 / 4189, 80 4139 / 4214 (no difference) and 192 3858 / 4442, 240 3813 / 4396 (their radix-16 last stage writes the
// image at a pitch of 12 / 15 points: LDS conflicts cost more than the half-used lines did).
__host__ __device__ constexpr bool generic_staged_io (int n) { return n <= CFFT_GENERIC_STAGED_MAX_N && (n == 96 || n == 144 || (CFFT_GENERIC_STAGED_160 != 0 && n == 160) || (CFFT_GENERIC_STAGED_48_80 != 0 && (n == 48 || n == 80))); }
// ... and whose REAL transforms (N = 2 n) do: real N = 192 4677 / 4423, 288 4767 / 4407 -- but 320 3368 / 3856
__host__ __device__ constexpr bool generic_staged_io_real (int n) { return n <= CFFT_GENERIC_STAGED_MAX_N && (n == 96 || n == 144 || (CFFT_GENERIC_STAGED_48_80 != 0 && (n == 48 || n == 80))); }
template <class SP>
struct SP_N_OR_ZERO
{
    static constexpr int value = SP::n;
    static constexpr int r0 = SP::radix (0); // radix of the first stage
};
template <>
struct SP_N_OR_ZERO<void>
{
    static constexpr int value = 1 << 30;
    static constexpr int r0 = 1;
};
// CFFT_GENERIC_PIPE / CFFT_GENERIC_CONV_PIPE (A/B switches): compile-time lists request the next group's first-stage operands one
// group ahead (generic_first_prefetch above).  Measured on MI355X, pipelined against not, same process order: the plain
// transforms LOSE more than they gain (c768 0.76 -> 0.69, c1920 0.75 -> 0.68, c2000 0.67 -> 0.63, c480 0.60 -> 0.58; c192, c640,
// r768, r1920 + 1-3 %): off.  The fused convolutions, with two transforms between a group's loads and stores, gain 1-4.5 %
// (c480 0.497 -> 0.511, c768 0.456 -> 0.477, c1536 0.516 -> 0.528): on.
#ifndef CFFT_GENERIC_PIPE
#define CFFT_GENERIC_PIPE 0
#endif
#ifndef CFFT_GENERIC_CONV_PIPE
#define CFFT_GENERIC_CONV_PIPE 1
#endif

// Copy of a twiddle table to LDS at kernel start, eight loads per work-item in flight (the plain loop is one round trip per
// BLOCK entries: a 1919-entry table on 256 work-items costs eight serial trips to L2 before the first frame).
template <int BLOCK>
__device__ __forceinline__ void lds_table_copy (float2* __restrict__ dst, const float2* __restrict__ src, int count, int tid)
{
#ifndef CFFT_TABLE_COPY_BATCH
#define CFFT_TABLE_COPY_BATCH 8
#endif
    constexpr int U = CFFT_TABLE_COPY_BATCH;
    for (int i0 = tid; i0 < count; i0 += U * BLOCK)
    {
        float2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = src[min (i0 + u * BLOCK, count - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * BLOCK < count)
                dst[i0 + u * BLOCK] = v[u];
    }
}

// ---- pair stage of the real transforms, kRealPairBatch pairs per work-item at a time ------------------------------------------
// The plain `for (k = tid; k < half; k += BLOCK)` loops compiled to one round trip per pair: the operands (backward: two loads
// of the spectrum per pair) and the pair twiddle sit behind a branch on k and an LDS access, so every iteration waited vmcnt(0) --
// and in the forward direction that wait also drained the previous pair's stores (one in-order counter).  Here the loads of a
// batch are issued first and unconditionally (indices clamped into the range), then the batch is computed and stored.
#ifndef CFFT_REAL_PAIR_BATCH
#define CFFT_REAL_PAIR_BATCH 4
#endif
static constexpr int kRealPairBatch = CFFT_REAL_PAIR_BATCH;

// the mirrored bins' LOADS of the backward transform: default cache policy where a workgroup holds five or more short frames (real N = 96 ... 800: +1 ... +5 %,
// the two pieces of a line are asked for by neighbouring instructions and meet in L2), nontemporal otherwise (real N = 960 at 2 GiB per buffer: -2 ... -4 %
// with plain loads; longer frames: 0).  profiles/r05_real_mirror_store.txt; CFFT_REAL_MIRROR_LOAD_PLAIN = 0: nontemporal everywhere (rounds 1-4).
#ifndef CFFT_REAL_MIRROR_LOAD_PLAIN
#define CFFT_REAL_MIRROR_LOAD_PLAIN 1
#endif
// backward: spectrum frames in global memory (frame stride `is`) -> Z, re <-> im swapped, in the LDS image (frame stride n)
template <int BLOCK>
__device__ __forceinline__ void real_merge_frames (const float2* __restrict__ src, unsigned is, float2* __restrict__ img, const float2* wp, int n,
                                                   int nframes, int tid)
{
    const int half = n / 2 + 1, total = nframes * half;
    const unsigned magic_half = div_magic ((unsigned) half);
    constexpr int U = kRealPairBatch;
    for (int idx0 = tid; idx0 < total; idx0 += U * BLOCK)
    {
        float2 A[U], X[U], W[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int idx = min (idx0 + u * BLOCK, total - 1);
            const int f = nframes == 1 ? 0 : fast_div (idx, magic_half), k = idx - f * half;
            const int kp = (k == 0) ? 0 : n - k;
            A[u] = gload (&src[(size_t) f * is + k]);
            X[u] = (CFFT_REAL_MIRROR_LOAD_PLAIN && nframes >= 5) ? src[(size_t) f * is + kp] : gload (&src[(size_t) f * is + kp]);
            W[u] = wp[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int idx = idx0 + u * BLOCK;
            if (idx < total)
            {
                const int f = nframes == 1 ? 0 : fast_div (idx, magic_half), k = idx - f * half;
                const int kp = (k == 0) ? 0 : n - k;
                float2 zk, zp;
                if (k == 0)
                {
                    zk = make_float2 (A[u].x + A[u].y, A[u].x - A[u].y); // Z[0] = (X0 + XM) + i (X0 - XM)
                    zp = zk;
                }
                else if (2 * k == n)
                {
                    zk = make_float2 (2.f * A[u].x, -2.f * A[u].y); // Z[n/2] = 2 conj(X[n/2])
                    zp = zk;
                }
                else
                    real_merge_pair (A[u], X[u], W[u], zk, zp);
                img[f * n + k] = make_float2 (zk.y, zk.x);
                if (kp != k)
                    img[f * n + kp] = make_float2 (zp.y, zp.x);
            }
        }
    }
}

// The mirrored half X[n - k] of the packed spectrum leaves in descending runs that start one element off a 128-byte line.  With the default cache policy
// (plain = true) the pieces of a line written by neighbouring instructions meet in L2 before they go out; nontemporal sends each piece on its own.
// Measured (round 5, profiles/r05_real_mirror_store.txt): workgroups that hold several short frames gain 3-5 % with plain stores (real N = 384 ... 1920),
// one-frame workgroups lose 1-3 % -- the caller says which it is.  CFFT_REAL_MIRROR_PLAIN = 0: nontemporal everywhere (rounds 1-4).
#ifndef CFFT_REAL_MIRROR_PLAIN
#define CFFT_REAL_MIRROR_PLAIN 1
#endif
__device__ __forceinline__ void mirror_store (float2* p, float2 v, bool plain)
{
    if (CFFT_REAL_MIRROR_PLAIN && plain)
        *p = v;
    else
        gstore (p, v);
}
// forward: Z in the LDS image (frame stride n) -> packed spectrum frames in global memory (frame stride `os`)
template <int BLOCK>
__device__ __forceinline__ void real_split_frames (const float2* __restrict__ img, float2* __restrict__ dst, unsigned os, const float2* wp, int n,
                                                   int nframes, int tid, bool plain_mirror = false)
{
    const int half = n / 2 + 1, total = nframes * half;
    const unsigned magic_half = div_magic ((unsigned) half);
    constexpr int U = kRealPairBatch;
    for (int idx0 = tid; idx0 < total; idx0 += U * BLOCK)
    {
        float2 W[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int idx = min (idx0 + u * BLOCK, total - 1);
            const int f = nframes == 1 ? 0 : fast_div (idx, magic_half);
            W[u] = wp[idx - f * half];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int idx = idx0 + u * BLOCK;
            if (idx < total)
            {
                const int f = nframes == 1 ? 0 : fast_div (idx, magic_half), k = idx - f * half;
                const int kp = (k == 0) ? 0 : n - k;
                const float2 A = img[f * n + k];
                const float2 Zp = img[f * n + kp];
                float2 Xk, Xp;
                if (k == 0)
                {
                    Xk = make_float2 (A.x + A.y, A.x - A.y); // [X0.re, X(N/2).re]
                    Xp = Xk;
                }
                else if (2 * k == n)
                {
                    Xk = make_float2 (A.x, -A.y); // X[n/2] = conj(Z[n/2])
                    Xp = Xk;
                }
                else
                    real_split_pair (A, Zp, W[u], Xk, Xp);
                gstore (&dst[(size_t) f * os + k], Xk);
                if (kp != k)
                    mirror_store (&dst[(size_t) f * os + kp], Xp, plain_mirror);
            }
        }
    }
}

// MODE 0: complex <-> complex (INV: backward).  MODE 1: real forward (packed spectrum out).  MODE 2: real backward.
// For the real modes plan.n = N/2 and wtab holds W_N^k, k <= N/4 (pair stage formulas: real_kernels.hpp).
// TWLDS: the plan's stage twiddles (n - 1 values: sum of (r - 1) p over the stages) are copied to LDS once per workgroup, behind
// the two images.  From L2 they cost 25-35 % (MI355X, n = 480 .. 1536: every butterfly waits for its own twiddle loads, and in
// the last stage -- gfx9 counts loads and stores in one in-order counter -- for the previous butterfly's stores as well).
template <int MODE, bool INV, int BLOCK, bool TWLDS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void generic_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ twtab,
                                                         const float2* __restrict__ wtab,
                                                         GenericPlan plan,
                                                         unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    constexpr bool STATIC = ! std::is_void<SP>::value; // compile-time stage list (static_core): n is a constant
    static_assert (! STATIC || TWLDS, "static plans take their twiddles from LDS");
    const int n = [&] {
        if constexpr (STATIC)
            return (int) SP::n;
        else
            return plan.n;
    }();
    const int F = plan.frames;
    const int half = n / 2 + 1; // pair-stage work items per frame: k = 0 .. n/2
    float2* bufA = smem;
    float2* bufB = smem + (size_t) F * n;
    float2* twl = smem + (size_t) 2 * F * n;
    // real transforms: the pair-stage twiddles W_N^k, k <= n / 2, ride behind the stage twiddles where that costs no workgroup
    // per CU (plan.pair_lds, decided by the launcher; read from L2 at their use they sit in front of the spectrum's stores, and
    // every "load W; wait" drains the stores before it: real N = 384 0.46 -> 0.58 of 8 TB/s)
    float2* wlds = twl + (n - 1);
    const bool pair_lds = TWLDS && MODE != 0 && plan.pair_lds != 0;
    if constexpr (TWLDS)
    {
        lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n - 1, tid);
        if (pair_lds)
        {
            lds_table_copy<BLOCK> (wlds, wtab, half, tid);
        }
        // the first barrier of the group loop orders these writes before any stage
    }
    const unsigned ngroups = (batch + F - 1) / F;
    constexpr int R0 = SP_N_OR_ZERO<SP>::r0;
    constexpr bool PIPE = STATIC && MODE != 2 && CFFT_GENERIC_PIPE != 0;
    float2 pre[R0];
    if constexpr (PIPE)
    {
        if (blockIdx.x < ngroups)
            generic_first_prefetch<R0, BLOCK, INV> (pre, in + (size_t) blockIdx.x * F * plan.in_stride, n, (int) min ((unsigned) F, batch - blockIdx.x * F), tid, plan.in_stride);
    }
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int nframes = (int) min ((unsigned) F, batch - g * F);
        const unsigned is = plan.in_stride, os = plan.out_stride;
        const float2* __restrict__ src = in + (size_t) g * F * is;
        float2* __restrict__ dst = out + (size_t) g * F * os;
        __syncthreads(); // previous group's reads of the images are finished
        if (MODE == 2)
        {
            // merge pairs straight from global memory (k and n-k), stored re<->im swapped for the inverse core
            // wp: the pair twiddles, in LDS or in L2 (one instantiation each)
            const auto merge_all = [&] (const float2* wp) { real_merge_frames<BLOCK> (src, is, bufA, wp, n, nframes, tid); };
            if (pair_lds)
                merge_all (wlds);
            else
                merge_all (wtab);
        }
        // Short frames (compile-time lists, complex, packed batch): with direct global I/O one first-stage load instruction
        // fetches eight 64-byte pieces (n = 96: x[i + 8 a], eight butterflies per frame) and one last-stage store writes 96-byte
        // runs at a 768-byte pitch -- half-used lines on both sides.  Here the workgroup's frames are ONE contiguous block:
        // it is copied to the first image with whole-workgroup 16-byte accesses, every stage runs LDS -> LDS, and the result
        // leaves the same way (two more LDS passes for fully used lines; the lengths that gain: generic_staged_io above;
        // CFFT_GENERIC_STAGED_MAX_N = 0: direct I/O everywhere).
        if constexpr (STATIC && MODE == 0)
        {
            if constexpr (generic_staged_io (SP_N_OR_ZERO<SP>::value))
            {
                if (is == (unsigned) n && os == (unsigned) n)
                {
                    const int quads = nframes * n / 2;
                    const float4* __restrict__ s4 = reinterpret_cast<const float4*> (src);
                    float4* a4 = reinterpret_cast<float4*> (bufA);
                    for (int e = tid; e < quads; e += BLOCK)
                    {
                        const float4 x = gload4 (&s4[e]);
                        a4[e] = INV ? make_float4 (x.y, x.x, x.w, x.z) : x;
                    }
                    __syncthreads();
                    const float2* r2 = static_core<SP, BLOCK, false, false, false> (twl, bufA, bufB, src, dst, nframes, tid, is, os);
                    const float4* r4 = reinterpret_cast<const float4*> (r2);
                    float4* __restrict__ d4 = reinterpret_cast<float4*> (dst);
                    for (int e = tid; e < quads; e += BLOCK)
                    {
                        const float4 x = r4[e];
                        gstore4 (&d4[e], INV ? make_float4 (x.y, x.x, x.w, x.z) : x);
                    }
                    continue;
                }
            }
        }
        // MODE 0 / 1: stage 0 reads the frames straight from global memory
        if (MODE == 2)
            __syncthreads();
        constexpr bool IN_GLOBAL = (MODE != 2), OUT_GLOBAL = (MODE != 1);
        const float2* res;
        // real transforms of the same lengths: the forward transform's samples come in by the staged copy, the backward
        // transform's leave by it (the pair stage keeps its own global accesses)
        bool staged_real = false;
        if constexpr (STATIC && MODE != 0)
        {
            if constexpr (generic_staged_io_real (SP_N_OR_ZERO<SP>::value))
                staged_real = (MODE == 1) ? (is == (unsigned) n) : (os == (unsigned) n);
        }
        if constexpr (STATIC && MODE != 0 && generic_staged_io_real (SP_N_OR_ZERO<SP>::value))
        {
            if (staged_real)
            {
                const int quads = nframes * n / 2;
                if constexpr (MODE == 1)
                {
                    const float4* __restrict__ s4 = reinterpret_cast<const float4*> (src);
                    float4* a4 = reinterpret_cast<float4*> (bufA);
                    for (int e = tid; e < quads; e += BLOCK)
                        a4[e] = gload4 (&s4[e]);
                    __syncthreads();
                    res = static_core<SP, BLOCK, false, false, false> (twl, bufA, bufB, src, dst, nframes, tid, plan.in_stride, plan.out_stride);
                }
                else
                {
                    // the merged pairs are in bufA, re <-> im swapped; the result leaves swapped back
                    const float2* r2 = static_core<SP, BLOCK, false, false, false> (twl, bufA, bufB, src, dst, nframes, tid, plan.in_stride, plan.out_stride);
                    const float4* r4 = reinterpret_cast<const float4*> (r2);
                    float4* __restrict__ d4 = reinterpret_cast<float4*> (dst);
                    for (int e = tid; e < quads; e += BLOCK)
                    {
                        const float4 x = r4[e];
                        gstore4 (&d4[e], make_float4 (x.y, x.x, x.w, x.z));
                    }
                    continue;
                }
            }
        }
        if (staged_real)
        {
            // MODE 1: res is set; fall through to the pair stage
        }
        else if constexpr (PIPE)
        {
            // first stage from the operands requested one group ago; the next group's are requested now and stay in flight
            // while the other stages of this group run
            generic_first_stage_pre<R0, BLOCK, INV> (pre, src, bufB, n, nframes, tid, is, (unsigned) n);
            __syncthreads();
            const unsigned gn = g + gridDim.x;
            if (gn < ngroups)
                generic_first_prefetch<R0, BLOCK, INV> (pre, in + (size_t) gn * F * is, n, (int) min ((unsigned) F, batch - gn * F), tid, is);
            res = static_core<SP, BLOCK, IN_GLOBAL, OUT_GLOBAL, INV, 1> (twl, bufB, bufA, src, dst, nframes, tid, is, os);
        }
        else if constexpr (STATIC)
            res = static_core<SP, BLOCK, IN_GLOBAL, OUT_GLOBAL, (MODE == 2) || INV> (twl, bufA, bufB, src, dst, nframes, tid, plan.in_stride, plan.out_stride);
        else if constexpr (TWLDS)
            res = generic_core<BLOCK, IN_GLOBAL, OUT_GLOBAL, (MODE == 2) || INV> (plan, twl, bufA, bufB, src, dst, nframes, tid, (unsigned) n, plan.tw_offset[0]);
        else
            res = generic_core<BLOCK, IN_GLOBAL, OUT_GLOBAL, (MODE == 2) || INV> (plan, twtab, bufA, bufB, src, dst, nframes, tid, (unsigned) n);
        if (MODE == 1)
        {
            const auto split_all = [&] (const float2* wp) { real_split_frames<BLOCK> (res, dst, os, wp, n, nframes, tid, F >= 2); }; // (several frames per workgroup: plain mirrored stores)
            if (pair_lds)
                split_all (wlds);
            else
                split_all (wtab);
        }
    }
}

// ---- long single frames (n > 2048): ONE LDS image, stages in place ------------------------------------------------------
// A Stockham stage cannot overwrite its input while other work-items still read it, which is why generic_kernel ping-pongs
// between two images.  With one frame per workgroup every work-item can afford to hold ALL of its butterflies of a stage
// in registers (at most 20 points per work-item: the workgroup size grows with n), so a stage becomes: read everything,
// barrier, twiddle + butterfly + write everything, barrier -- on a single image.  Half the LDS means two workgroups per
// CU up to n = 10240 (their load / compute / store phases overlap), and frames of up to 20000 points fit at all.
static constexpr int kInplaceMaxPoints = 20; // points per work-item and stage

template <int R, int BLOCK, int IO, bool SWAP, bool STATIC = false, bool CONV = false>
__device__ __forceinline__ void inplace_stage (const float2* __restrict__ gin, float2* __restrict__ gout, float2* __restrict__ buf,
                                               const float2* __restrict__ tw, int n, int p, int tid,
                                               const float2* __restrict__ H = nullptr, float scaling = 1.f)
{
    constexpr int J = (kInplaceMaxPoints + R - 1) / R; // butterflies per work-item
    // n passes through an opaque (volatile) statement: everything below derives from it, so the loop-invariant code motion of
    // the stage loop cannot lift the operand / twiddle offsets of every radix case out of the loop and hold them all across it
    // (with eleven radix cases that doubled the kernel's registers: 80 for the worst single case, 179-210 for the kernel).
    if constexpr (! STATIC) // (compile-time n and p at the call: static_inplace_core)
        asm volatile ("; inplace_stage R=%1 IO=%2" : "+s"(n) : "n"(R), "n"(IO));
    const int m = n / R;
    if constexpr ((IO & 1) != 0)
    {
        // the stage that reads global memory is always the first one (p == 1: no twiddles, output index i * R + b) and
        // overwrites nothing it reads: U butterflies at a time, all their loads issued before the first butterfly
#ifndef CFFT_INPLACE_FIRST_U_BIG
#define CFFT_INPLACE_FIRST_U_BIG 1 // (2: no change for the compile-time lists, 24-32 more VGPRs for the run-time ones)
#endif
        constexpr int U = R <= 4 ? 4 : (R <= 8 ? 2 : CFFT_INPLACE_FIRST_U_BIG);
#pragma unroll 1
        for (int i0 = tid; i0 < m; i0 += U * BLOCK)
        {
            float2 u[U][R];
#pragma unroll
            for (int q = 0; q < U; ++q)
            {
                const int i = i0 + q * BLOCK;
                if (i < m)
                {
#pragma unroll
                    for (int a = 0; a < R; ++a)
                    {
                        const float2 x = gload (&gin[(unsigned) (i + a * m)]); // 32-bit offset from a uniform base
                        u[q][a] = SWAP ? make_float2 (x.y, x.x) : x;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < U; ++q)
            {
                const int i = i0 + q * BLOCK;
                if (i < m)
                {
                    fft_reg<R> (u[q]);
#pragma unroll
                    for (int b = 0; b < R; ++b)
                    {
                        if (IO & 2)
                            gstore (&gout[(unsigned) (i * R + b)], SWAP ? make_float2 (u[q][b].y, u[q][b].x) : u[q][b]);
                        else
                            buf[i * R + b] = u[q][b];
                    }
                }
            }
        }
        if (! (IO & 2))
            __syncthreads();
        return;
    }
    else
    {
        const bool p_pow2 = (p & (p - 1)) == 0;
        const unsigned magic_p = p > 1 ? div_magic ((unsigned) p) : 0u;
        float2 u[J][R];
#pragma unroll
        for (int j = 0; j < J; ++j)
        {
            const int i = tid + j * BLOCK;
            if (i < m)
            {
#pragma unroll
                for (int a = 0; a < R; ++a)
                    u[j][a] = buf[i + a * m];
            }
        }
        __syncthreads(); // every operand of this stage has left the image
#pragma unroll
        for (int j = 0; j < J; ++j)
        {
            const int i = tid + j * BLOCK;
            if (i < m)
            {
                const int k = p_pow2 ? (i & (p - 1)) : (i - fast_div (i, magic_p) * p);
                const int jj = (i - k) * R + k;
                if (p > 1)
                {
#pragma unroll
                    for (int a = 1; a < R; ++a)
                        u[j][a] = cmul (u[j][a], tw[(a - 1) * p + k]);
                }
                fft_reg<R> (u[j]);
#pragma unroll
                for (int b = 0; b < R; ++b)
                {
                    if (IO & 2)
                        gstore (&gout[(unsigned) (jj + b * p)], SWAP ? make_float2 (u[j][b].y, u[j][b].x) : u[j][b]);
                    else if constexpr (CONV)
                    {
                        // fused convolution: the forward transform's last stage multiplies by H * scaling and swaps (generic_stage)
                        const float2 h = H[jj + b * p];
                        const float2 y = cmul (u[j][b], make_float2 (h.x * scaling, h.y * scaling));
                        buf[jj + b * p] = make_float2 (y.y, y.x);
                    }
                    else
                        buf[jj + b * p] = u[j][b];
                }
            }
        }
        if (! (IO & 2))
            __syncthreads(); // the image holds this stage's result
    }
}

template <int BLOCK, int IO, bool SWAP, bool CONV = false>
__device__ __forceinline__ void inplace_stage_dispatch (int r, const float2* __restrict__ gin, float2* __restrict__ gout, float2* __restrict__ buf,
                                                        const float2* __restrict__ tw, int n, int p, int tid,
                                                        const float2* __restrict__ H = nullptr, float scaling = 1.f)
{
    switch (r)
    {
        case 2: inplace_stage<2, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 3: inplace_stage<3, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 4: inplace_stage<4, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 5: inplace_stage<5, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 6: inplace_stage<6, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 8: inplace_stage<8, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 9: inplace_stage<9, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 10: inplace_stage<10, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 12: inplace_stage<12, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        case 15: inplace_stage<15, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
        default: inplace_stage<16, BLOCK, IO, SWAP, false, CONV> (gin, gout, buf, tw, n, p, tid, H, scaling); break;
    }
}

// TWLDS: stages [0, plan.lds_stages) take their twiddles from the workgroup's LDS copy `twl` (twl[0] = table entry
// plan.tw_offset[0]), the rest from the table in L2.  Two call sites per kind of stage so that each keeps its address space.
// CONV: the last stage (which then writes the image: OUT_GLOBAL false) multiplies by H * scaling and swaps (inplace_stage)
template <int BLOCK, bool IN_GLOBAL, bool OUT_GLOBAL, bool SWAP, bool TWLDS, bool CONV = false>
__device__ __forceinline__ void inplace_core (const GenericPlan& plan, const float2* __restrict__ twtab, const float2* twl, float2* buf,
                                              const float2* __restrict__ gin, float2* __restrict__ gout, int tid,
                                              const float2* __restrict__ H = nullptr, float scaling = 1.f)
{
    int p = 1, s0 = 0;
    if constexpr (IN_GLOBAL)
    {
        // the stage that reads global memory stands outside the stage loop (p == 1: no twiddles): with its radix cases
        // inside, the loop carried 40 more registers
        const int r = plan.radix[0];
        if (OUT_GLOBAL && plan.ns == 1)
            inplace_stage_dispatch<BLOCK, 3, SWAP> (r, gin, gout, buf, twtab, plan.n, 1, tid);
        else
            inplace_stage_dispatch<BLOCK, 1, SWAP> (r, gin, gout, buf, twtab, plan.n, 1, tid);
        p = r;
        s0 = 1;
    }
    for (int s = s0; s < plan.ns; ++s)
    {
        const int r = plan.radix[s];
        const float2* tw = twtab + plan.tw_offset[s];
        const bool last = OUT_GLOBAL && s == plan.ns - 1;
        if (CONV && ! OUT_GLOBAL && s == plan.ns - 1) // (a plan's last table never rides in LDS with the earlier ones missing: take it from L2)
            inplace_stage_dispatch<BLOCK, 0, false, CONV> (r, gin, gout, buf, tw, plan.n, p, tid, H, scaling);
        else if (TWLDS && s < plan.lds_stages)
        {
            const float2* tl = twl + (plan.tw_offset[s] - plan.tw_offset[0]);
            if (last)
                inplace_stage_dispatch<BLOCK, 2, SWAP> (r, gin, gout, buf, tl, plan.n, p, tid);
            else
                inplace_stage_dispatch<BLOCK, 0, false> (r, gin, gout, buf, tl, plan.n, p, tid);
        }
        else if (last)
            inplace_stage_dispatch<BLOCK, 2, SWAP> (r, gin, gout, buf, tw, plan.n, p, tid);
        else
            inplace_stage_dispatch<BLOCK, 0, false> (r, gin, gout, buf, tw, plan.n, p, tid);
        p *= r;
    }
}

// The same with a compile-time stage list (StaticPlan): for the long frames of the reference's test list (9216 complex points, and
// 4608 for the real transform of 9216) and a handful of lengths audio hosts use.  Whether a stage's twiddles come from the LDS
// copy or from L2 stays a run-time (uniform) choice: the launcher decides how many tables fit.
template <class SP, int BLOCK, bool IN_GLOBAL, bool OUT_GLOBAL, bool SWAP, bool TWLDS, int S = 0, bool CONV = false>
__device__ __forceinline__ void static_inplace_core (const GenericPlan& plan, const float2* __restrict__ twtab, const float2* twl, float2* buf,
                                                     const float2* __restrict__ gin, float2* __restrict__ gout, int tid,
                                                     const float2* __restrict__ H = nullptr, float scaling = 1.f)
{
    if constexpr (S < SP::ns)
    {
        constexpr int R = SP::radix (S), P = SP::period (S);
        constexpr bool first = IN_GLOBAL && S == 0, last = OUT_GLOBAL && S == SP::ns - 1;
        constexpr int IO = (first ? 1 : 0) | (last ? 2 : 0);
        constexpr bool conv_here = CONV && ! OUT_GLOBAL && S == SP::ns - 1;
        if constexpr (first)
            inplace_stage<R, BLOCK, IO, SWAP, true> (gin, gout, buf, twtab, SP::n, 1, tid); // p == 1: no twiddles
        else
        {
            bool from_lds = false;
            if constexpr (TWLDS)
                from_lds = S < plan.lds_stages;
            if (from_lds)
                inplace_stage<R, BLOCK, IO, IO != 0 && SWAP, true, conv_here> (gin, gout, buf, twl + SP::tw_offset (S), SP::n, P, tid, H, scaling);
            else
                inplace_stage<R, BLOCK, IO, IO != 0 && SWAP, true, conv_here> (gin, gout, buf, twtab + plan.tw_offset[0] + SP::tw_offset (S), SP::n, P, tid, H, scaling);
        }
        static_inplace_core<SP, BLOCK, IN_GLOBAL, OUT_GLOBAL, SWAP, TWLDS, S + 1, CONV> (plan, twtab, twl, buf, gin, gout, tid, H, scaling);
    }
}
template <int N_> struct StaticInplacePlanFor { using type = void; };
// (the lists: generic_static_inplace.hpp, generated from the planner for every length of this kernel)

// same MODEs as generic_kernel; one frame per workgroup pass
// 512 work-items: at most 128 VGPRs (4 waves per SIMD), so that two workgroups share a CU where their images fit
template <int MODE, bool INV, int BLOCK, bool TWLDS, class SP = void>
__global__ __launch_bounds__ (BLOCK, BLOCK == 512 ? 4 : 1) void generic_inplace_kernel (const float2* __restrict__ in,
                                                                 float2* __restrict__ out,
                                                                 const float2* __restrict__ twtab,
                                                                 const float2* __restrict__ wtab,
                                                                 GenericPlan plan,
                                                                 unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid0 = threadIdx.x;
    constexpr bool STATIC = ! std::is_void<SP>::value;
    const int n = [&] {
        if constexpr (STATIC)
            return (int) SP::n;
        else
            return plan.n;
    }();
    float2* buf = smem;
    float2* twl = smem + n;
    if constexpr (TWLDS)
    {
        lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], plan.lds_tw_len, tid0);
        // the first barrier of the frame loop orders these writes before any stage
    }
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        const float2* __restrict__ src = in + (size_t) g * plan.in_stride;
        float2* __restrict__ dst = out + (size_t) g * plan.out_stride;
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid); // per frame: nothing derived from the work-item index is lifted out of the frame loop and held
        __syncthreads(); // previous frame's reads of the image are finished
        if (MODE == 2)
        {
            real_merge_frames<BLOCK> (src, 0u, buf, wtab, n, 1, tid);
            __syncthreads();
        }
        constexpr bool IN_GLOBAL = (MODE != 2), OUT_GLOBAL = (MODE != 1);
        if constexpr (STATIC)
            static_inplace_core<SP, BLOCK, IN_GLOBAL, OUT_GLOBAL, (MODE == 2) || INV, TWLDS> (plan, twtab, twl, buf, src, dst, tid);
        else
            inplace_core<BLOCK, IN_GLOBAL, OUT_GLOBAL, (MODE == 2) || INV, TWLDS> (plan, twtab, twl, buf, src, dst, tid);
        if (MODE == 1)
            real_split_frames<BLOCK> (buf, dst, 0u, wtab, n, 1, tid);
    }
}

// Loads of a tile, kTileLoadsInFlight per work-item at a time: the plain `for (e = tid; e < total; e += BLOCK)` loop compiles to
// one global_load followed by s_waitcnt vmcnt(0) per iteration (run-time trip count, LDS store behind every load), i.e. 9 full
// round trips per work-item for a 16 x 144 tile and 80 % of the wave cycles parked (profiles/r03_general_two_pass.txt).
// Every address is clamped into the tile so that the loads are unconditional (loads inside divergent branches get serialised
// the same way); what does not exist is replaced by zero afterwards.
#ifndef CFFT_TILE_LOADS_IN_FLIGHT
#define CFFT_TILE_LOADS_IN_FLIGHT 8
#endif
static constexpr int kTileLoadsInFlight = CFFT_TILE_LOADS_IN_FLIGHT;
#ifndef CFFT_TILE_PREFETCH_B
#define CFFT_TILE_PREFETCH_B 4
#endif
static constexpr int kTilePrefetchB = CFFT_TILE_PREFETCH_B;
#ifndef CFFT_TILE_PREFETCH_A
#define CFFT_TILE_PREFETCH_A 8
#endif
static constexpr int kTilePrefetchA = CFFT_TILE_PREFETCH_A;
static constexpr int kTileStoreBatch = 10;

// ---- in-place stages over a TILE of C frames (compile-time lists only) ----------------------------------------------------------
// The two-image Stockham form needs 2 x C x (n + 1) points of LDS: beyond n = 639 that forces 8-column tiles (64-byte pieces of
// the callers' rows) and from n = 324 on it leaves ONE workgroup per CU.  Here every work-item reads the operands of its
// butterflies of a stage into registers, the workgroup meets at a barrier, and the results overwrite the SAME image (the scheme
// of inplace_stage above, across C frames): one image, so 16-column tiles up to n = 1200 and two or three workgroups per CU up
// to n = 600.  J x R <= 24 points per work-item and stage for every length of the lists.
template <int R, int BLOCK, int C, int N, int P, int LFS>
__device__ __forceinline__ void tile_inplace_stage (float2* __restrict__ buf, const float2* __restrict__ tw, int tid)
{
    constexpr int M = N / R;        // butterflies per frame
    constexpr int TOTAL = C * M;    // ... per tile
    constexpr int J = (TOTAL + BLOCK - 1) / BLOCK;
    float2 u[J][R];
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
        const int idx = tid + j * BLOCK;
        if (J * BLOCK == TOTAL || idx < TOTAL)
        {
            const int f = idx / M, i = idx - f * M;
#pragma unroll
            for (int a = 0; a < R; ++a)
                u[j][a] = buf[f * LFS + i + a * M];
        }
    }
    __syncthreads(); // every operand of this stage has left the image
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
        const int idx = tid + j * BLOCK;
        if (J * BLOCK == TOTAL || idx < TOTAL)
        {
            const int f = idx / M, i = idx - f * M;
            const int k = i % P, jj = (i - k) * R + k;
            if constexpr (P > 1)
            {
#pragma unroll
                for (int a = 1; a < R; ++a)
                    u[j][a] = cmul (u[j][a], tw[(a - 1) * P + k]);
            }
            fft_reg<R> (u[j]);
#pragma unroll
            for (int b = 0; b < R; ++b)
                buf[f * LFS + jj + b * P] = u[j][b];
        }
    }
    __syncthreads(); // the image holds this stage's result
}

template <class SP, int BLOCK, int C, int LFS, int S = 0>
__device__ __forceinline__ void static_tile_inplace_core (const float2* twl, float2* buf, int tid)
{
    if constexpr (S < SP::ns)
    {
        tile_inplace_stage<SP::radix (S), BLOCK, C, SP::n, SP::period (S), LFS> (buf, twl + SP::tw_offset (S), tid);
        static_tile_inplace_core<SP, BLOCK, C, LFS, S + 1> (twl, buf, tid);
    }
}

// XCD-aware order of the tiles (speed only: a bijection of [0, nwork)).  Workgroups are dealt round-robin over the 8 XCDs, each
// with an L2 of its own; item v of the persistent loops (v = blockIdx + k gridDim, gridDim a multiple of 8 on a full chip) is
// mapped so that the tiles ONE XCD works on at a time are consecutive ones: where a tile is 8 or 4 columns wide (64- / 32-byte
// pieces of 128-byte lines) the neighbours' pieces of a line then meet in one L2 instead of being fetched / written by two.
// Measured on MI355X against the plain order (profiles/r03_general_two_pass.txt): the in-place column pass of the 2-D transforms
// gains 17-20 % with narrow tiles (1024 x 1024 0.214 -> 0.253 of 8 TB/s, 1080 x 1920 0.215 -> 0.250, 2048 x 2048 0.092 -> 0.111)
// and nothing with 16 columns; the two passes of the long 1-D plans, which hand a workspace from one order to the other, move
// by -7 .. +6 % with no pattern (c655360 +4 %, c2^21 -7 %, c1000000 -4 %) and keep the plain order.
#ifndef CFFT_TILE_XCD_ORDER
#define CFFT_TILE_XCD_ORDER 1 // bit 0: the column pass, bit 1: pass A / pass B
#endif
template <int WHICH>
__device__ __forceinline__ size_t tile_xcd_order (size_t v, size_t nwork)
{
    if ((CFFT_TILE_XCD_ORDER & WHICH) == 0)
        return v;
    const size_t q = nwork / 8, r = nwork % 8, x = v % 8;
    return x * q + (x < r ? x : r) + v / 8; // XCD x owns q (+ 1 while x < r) consecutive tiles from x q + min (x, r)
}

// SP (tile kernels): compile-time stage list of the tile's frame length (the lists of StaticPlanFor, checked against the
// planner's by the launcher): the radix switch, the stage loop and every division fold away, and the tile is requested with
// exactly as many loads per work-item as it has.  SP = void: the run-time list of GenericPlan.
template <class SP>
__device__ __forceinline__ int tile_frame_len (const GenericPlan& plan)
{
    if constexpr (std::is_void<SP>::value)
        return plan.n;
    else
        return SP::n;
}
// operands (or results) of a tile per work-item; the one-image kernels hold up to 19 and request at most `cap` of them ahead
template <class SP, int C, int BLOCK>
constexpr int tile_slots (int dynamic, int cap = 1 << 20)
{
    if constexpr (std::is_void<SP>::value)
        return dynamic;
    else
        return (C * SP::n + BLOCK - 1) / BLOCK < cap ? (C * SP::n + BLOCK - 1) / BLOCK : cap;
}
#ifndef CFFT_TILE_INPLACE_PREFETCH
#define CFFT_TILE_INPLACE_PREFETCH 20
#endif

// ---- long mixed-radix frames (complex length > 20000): pass A of the two-pass plan -----------------------------------
// N = n1 * n2 as in multipass_kernels.hpp, with n2 = (odd part) * 2^j a mixed-radix length and n1 a power of two:
// this kernel does the n2-point COLUMN FFTs of a tile of CA (16, or 8 for n2 > 640) adjacent columns with the run-time stage list above and
// writes them, times W_N^{a kb}, in the same tile-major workspace layout as pass_a_kernel; pass B is then the ordinary
// power-of-two pass_b_kernel.  LDS: two images of 16 frames of n2 + 1 points (the + 1 keeps the 16 lanes that walk
// across frames while the tile is loaded / stored off each other's banks).
// BLOCK: 256 work-items, or 1024 where the two images leave room for only one workgroup per CU (4 waves would be all a CU has)
// PIPE: the first kTileLoadsInFlight operands per work-item of the NEXT tile are requested as soon as this tile's have been
// committed to the image, and stay in registers across the stages and the stores (the tile kernels were parked on memory for
// 65-80 % of their wave cycles: a workgroup had loads in flight only during its own load phase).  vmcnt is one in-order
// counter, so nothing between the request and the end of the stages may wait on a younger load: the stage twiddles are then
// read from an LDS copy behind the images (plan.n - 1 entries; the launcher grants PIPE where that costs no workgroup per CU).
template <bool SWAP_IN, int CA, int BLOCK, bool PIPE = false, class SP = void, bool INPLACE = false>
__global__ __launch_bounds__ (BLOCK, PIPE ? 4 : 1) void pass_a_generic_kernel (const float2* __restrict__ in,
                                                              float2* __restrict__ work,
                                                              const float2* __restrict__ twtab,
                                                              const float2* __restrict__ big_tw,
                                                              GenericPlan plan, // n = n2, frames = CA
                                                              unsigned n1,
                                                              unsigned batch,
                                                              size_t in_stride,
                                                              unsigned lfs, // frame stride inside an image: n2 + 1, or n2 where the padded images would not fit the CU's LDS
                                                              size_t ws_frame_len) // float2 per frame in the workspace: tiles * n2 * CA (>= n1 * n2: the last tile may be ragged)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid0 = threadIdx.x;
    constexpr bool STATIC = ! std::is_void<SP>::value;
    static_assert (! STATIC || PIPE, "compile-time lists come with the pipelined form only");
    static_assert (! INPLACE || STATIC, "one-image tiles exist for compile-time lists only");
    const int n2 = tile_frame_len<SP> (plan);
    if constexpr (STATIC)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs; // (INPLACE: no second image, the twiddle copy sits here)
    float2* twl = smem + (size_t) (INPLACE ? 1 : 2) * CA * lfs;
    if constexpr (PIPE)
    {
        lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n2 - 1, tid0);
    }
    const unsigned tiles = (n1 + CA - 1) / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2;
    // Compile-time lists request the whole tile ahead (exactly its count of operands per work-item); the run-time list requests
    // kTilePrefetchA = 8 of the at most 10 a tile shape has (16 x 160 / 256, 16 x 639 / 1024, 8 x 1280 / 1024, 4 x 2559 / 1024: ten
    // spill 8-12 bytes in the narrow-tile variants) and loads the rest when the tile is committed.  Slots beyond the tile re-read
    // its last element: a test around a load -- even a workgroup-uniform one -- makes the compiler wait for every load before the next.
    constexpr int PF = tile_slots<SP, CA, BLOCK> (kTilePrefetchA, INPLACE ? CFFT_TILE_INPLACE_PREFETCH : 1 << 20);
    constexpr int SB = tile_slots<SP, CA, BLOCK> (CA == 16 ? kTileStoreBatch : 8, INPLACE ? 10 : 1 << 20); // (run-time list: 10 spills a few bytes in the narrow-tile variants)
    float2 pre[PF];
    const auto request = [&] (size_t v) {
        int tid = tid0;
        if constexpr (INPLACE) // (per tile: nothing derived from the work-item index is lifted out of the tile loop and held -- the
            CFFT_LAUNDER_INDEX (tid); // one-image kernels spill otherwise; the two-image ones are 10-15 % faster with the offsets held)
        const size_t w = tile_xcd_order<2> (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const int valid = (int) min ((unsigned) CA, n1 - tile * CA);
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        const auto slot = [&] (int u) {
            const int e = min (tid + u * BLOCK, total - 1);
            const int c = min (e % CA, valid - 1), j = e / CA;
            pre[u] = gload (&src[(size_t) j * n1 + c]);
        };
        // run-time list: short tiles on 1024 work-items have 2-4 operands each -- one branch around the whole group of loads
        // (not one per load) keeps the other six from re-reading the tile's last element
        if (! STATIC && PF > 4 && total <= 4 * BLOCK)
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                slot (u);
        }
        else
        {
#pragma unroll
            for (int u = 0; u < PF; ++u)
                slot (u);
        }
    };
    if constexpr (PIPE)
    {
        if (blockIdx.x < nwork)
            request (blockIdx.x);
    }
    for (size_t v = blockIdx.x; v < nwork; v += gridDim.x)
    {
        int tid = tid0;
        if constexpr (INPLACE)
            CFFT_LAUNDER_INDEX (tid);
        const size_t w = tile_xcd_order<2> (v, nwork);
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const int valid = (int) min ((unsigned) CA, n1 - tile * CA); // columns of this tile that exist (n1 need not be a multiple of CA)
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) n2 * CA);
        const float2* __restrict__ twp = big_tw + (size_t) tile * ((size_t) n2 * CA);
        __syncthreads(); // previous tile fully stored
        if constexpr (PIPE)
        {
#pragma unroll
            for (int u = 0; u < PF; ++u)
            {
                const int e = tid + u * BLOCK;
                const int c = e % CA, j = e / CA;
                if (e < total)
                    bufA[c * lfs + j] = c < valid ? (SWAP_IN ? make_float2 (pre[u].y, pre[u].x) : pre[u]) : make_float2 (0.f, 0.f);
            }
        }
        for (int e0 = tid + (PIPE ? PF * BLOCK : 0); e0 < total; e0 += BLOCK * kTileLoadsInFlight)
        {
            float2 x[kTileLoadsInFlight];
#pragma unroll
            for (int u = 0; u < kTileLoadsInFlight; ++u)
            {
                const int e = min (e0 + u * BLOCK, total - 1);
                const int c = min (e % CA, valid - 1), j = e / CA;
                x[u] = gload (&src[(size_t) j * n1 + c]);
            }
#pragma unroll
            for (int u = 0; u < kTileLoadsInFlight; ++u)
            {
                const int e = e0 + u * BLOCK;
                const int c = e % CA, j = e / CA;
                if (e < total)
                    bufA[c * lfs + j] = c < valid ? (SWAP_IN ? make_float2 (x[u].y, x[u].x) : x[u]) : make_float2 (0.f, 0.f);
            }
        }
        __syncthreads();
        const float2* res;
        if constexpr (PIPE)
        {
            if (v + gridDim.x < nwork)
                request (v + gridDim.x);
            if constexpr (INPLACE)
            {
                static_tile_inplace_core<SP, BLOCK, CA, SP_N_OR_ZERO<SP>::value + 1> (twl, bufA, tid);
                res = bufA;
            }
            else if constexpr (STATIC)
                res = static_core<SP, BLOCK, false, false, false, 0, false, SP_N_OR_ZERO<SP>::value + 1> (twl, bufA, bufB, nullptr, nullptr, CA, tid, 0u, 0u);
            else
                res = generic_core<BLOCK, false, false, false> (plan, twl, bufA, bufB, nullptr, nullptr, CA, tid, lfs, plan.tw_offset[0]);
        }
        else
            res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, CA, tid, lfs);
        // the inter-pass twiddles of the whole tile first, then the stores: a twiddle load behind stores would wait for them
        // too (one in-order counter) -- a store drain per tile
        const auto store_tile = [&] (auto batch) {
            constexpr int NB = decltype (batch)::value;
            for (int e0 = tid; e0 < total; e0 += BLOCK * NB)
            {
                float2 tw[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    tw[u] = twp[min (e0 + u * BLOCK, total - 1)];
#pragma unroll
                for (int u = 0; u < NB; ++u)
                {
                    const int e = e0 + u * BLOCK;
                    const int c = e % CA, k = e / CA;
                    if (e < total)
                        dst[e] = cmul (res[c * lfs + k], tw[u]); // plain store: the workspace should stay cached (padding columns: zeros)
                }
            }
        };
        if (! STATIC && CA == 16 && total <= 4 * BLOCK) // (the narrow-tile variants spill with the second form)
            store_tile (std::integral_constant<int, 4> {});
        else
            store_tile (std::integral_constant<int, SB> {});
    }
}

// Row pass of the same plans when the row length n1 is not a power of two either (and for powers of two beyond the
// compile-time cores): n1-point FFTs of CB adjacent rows kb with the run-time stage list, read from pass A's tile-major
// workspace [tile tA][kb][c] (CA columns per tile), stored transposed to X[n2 ka + kb].  n2 need not be a multiple of CB.
template <bool SWAP_OUT, int CB, int BLOCK, bool PIPE = false, class SP = void, bool INPLACE = false>
__global__ __launch_bounds__ (BLOCK) void pass_b_generic_kernel (const float2* __restrict__ work,
                                                              float2* __restrict__ out,
                                                              const float2* __restrict__ twtab,
                                                              GenericPlan plan, // n = n1, frames = CB
                                                              unsigned n2,
                                                              unsigned ca, // pass A's tile width
                                                              unsigned batch,
                                                              size_t out_stride,
                                                              unsigned lfs,
                                                              size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid0 = threadIdx.x;
    constexpr bool STATIC = ! std::is_void<SP>::value;
    static_assert (! STATIC || PIPE, "compile-time lists come with the pipelined form only");
    static_assert (! INPLACE || STATIC, "one-image tiles exist for compile-time lists only");
    const int n1 = tile_frame_len<SP> (plan);
    if constexpr (STATIC)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs; // (INPLACE: no second image, the twiddle copy sits here)
    float2* twl = smem + (size_t) (INPLACE ? 1 : 2) * CB * lfs; // PIPE (see pass_a_generic_kernel): the stage twiddles' LDS copy
    if constexpr (PIPE)
    {
        lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n1 - 1, tid0);
    }
    const unsigned tiles = (n2 + CB - 1) / CB;
    const unsigned magic_n1 = div_magic ((unsigned) n1), magic_ca = ca > 1 ? div_magic (ca) : 0u;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * n1;
    constexpr int PF = tile_slots<SP, CB, BLOCK> (kTilePrefetchB, INPLACE ? CFFT_TILE_INPLACE_PREFETCH : 1 << 20); // (run-time list: 8 costs pass B a workgroup per CU with 256 work-items and spills with 1024)
    float2 pre[PF];
    // element (row kbl, column a): c = a % ca fastest (contiguous in the workspace), then the row, then pass A's tile
    const auto element = [&] (const float2* __restrict__ src, unsigned kb0, int valid, int e) {
        const unsigned kbl = (unsigned) fast_div (e, magic_n1), a = (unsigned) e - kbl * (unsigned) n1;
        const unsigned tA = (unsigned) fast_div ((int) a, magic_ca), c = a - tA * ca;
        return src[((size_t) tA * n2 + kb0 + min (kbl, (unsigned) (valid - 1))) * ca + c];
    };
    const auto request = [&] (size_t v) {
        int tid = tid0;
        if constexpr (INPLACE) // (per tile: nothing derived from the work-item index is lifted out of the tile loop and held -- the
            CFFT_LAUNDER_INDEX (tid); // one-image kernels spill otherwise; the two-image ones are 10-15 % faster with the offsets held)
        const size_t w = tile_xcd_order<2> (v, nwork);
        const unsigned f = (unsigned) (w / tiles), kb0 = (unsigned) (w % tiles) * CB;
        const int valid = (int) min ((unsigned) CB, n2 - kb0);
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
#pragma unroll
        for (int u = 0; u < PF; ++u)
            pre[u] = element (src, kb0, valid, min (tid + u * BLOCK, total - 1));
    };
    if constexpr (PIPE)
    {
        if (blockIdx.x < nwork)
            request (blockIdx.x);
    }
    for (size_t v = blockIdx.x; v < nwork; v += gridDim.x)
    {
        int tid = tid0;
        if constexpr (INPLACE)
            CFFT_LAUNDER_INDEX (tid);
        const size_t w = tile_xcd_order<2> (v, nwork);
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, n2 - kb0);
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
        float2* __restrict__ dst = out + (size_t) f * out_stride + kb0;
        __syncthreads();
        if constexpr (PIPE)
        {
#pragma unroll
            for (int u = 0; u < PF; ++u)
            {
                const int e = tid + u * BLOCK;
                const unsigned kbl = (unsigned) fast_div (e, magic_n1), a = (unsigned) e - kbl * (unsigned) n1;
                if (e < total)
                    bufA[kbl * lfs + a] = (int) kbl < valid ? pre[u] : make_float2 (0.f, 0.f);
            }
        }
        for (int e0 = tid + (PIPE ? PF * BLOCK : 0); e0 < total; e0 += BLOCK * kTileLoadsInFlight)
        {
            float2 x[kTileLoadsInFlight];
#pragma unroll
            for (int u = 0; u < kTileLoadsInFlight; ++u)
                x[u] = element (src, kb0, valid, min (e0 + u * BLOCK, total - 1));
#pragma unroll
            for (int u = 0; u < kTileLoadsInFlight; ++u)
            {
                const int e = e0 + u * BLOCK;
                const unsigned kbl = (unsigned) fast_div (e, magic_n1), a = (unsigned) e - kbl * (unsigned) n1;
                if (e < total)
                    bufA[kbl * lfs + a] = (int) kbl < valid ? x[u] : make_float2 (0.f, 0.f);
            }
        }
        __syncthreads();
        const float2* res;
        if constexpr (PIPE)
        {
            if (v + gridDim.x < nwork)
                request (v + gridDim.x);
            if constexpr (INPLACE)
            {
                static_tile_inplace_core<SP, BLOCK, CB, SP_N_OR_ZERO<SP>::value + 1> (twl, bufA, tid);
                res = bufA;
            }
            else if constexpr (STATIC)
                res = static_core<SP, BLOCK, false, false, false, 0, false, SP_N_OR_ZERO<SP>::value + 1> (twl, bufA, bufB, nullptr, nullptr, CB, tid, 0u, 0u);
            else
                res = generic_core<BLOCK, false, false, false> (plan, twl, bufA, bufB, nullptr, nullptr, CB, tid, lfs, plan.tw_offset[0]);
        }
        else
            res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, CB, tid, lfs);
#pragma unroll 4 // (compile-time trip counts: fully unrolled, the 64-bit store addresses of up to 19 rounds are formed early and held)
        for (int e = tid; e < CB * n1; e += BLOCK)
        {
            const int c = e % CB, k = e / CB;
            if (c < valid)
            {
                const float2 y = res[c * lfs + k];
                gstore (&dst[(size_t) k * n2 + c], SWAP_OUT ? make_float2 (y.y, y.x) : y);
            }
        }
    }
}

// ---- 2-D transforms: the column pass ---------------------------------------------------------
// `batch` row-major [rows][cols] complex arrays whose rows have already been transformed: rows-point FFTs along every
// column, in place.  Same tile scheme as the two-pass column pass above -- C adjacent columns staged through LDS so that
// every global access touches C * 8 contiguous bytes, the FFTs by the run-time stage list -- without the inter-pass
// twiddle, and stored back where it came from.  cols need not be a multiple of C.
// PIPE / SP: as in pass_a_generic_kernel (the next tile requested across the stages, stage twiddles in LDS; compile-time list).
template <bool SWAP, int C, int BLOCK, bool PIPE = false, class SP = void, bool INPLACE = false>
__global__ __launch_bounds__ (BLOCK, PIPE ? 4 : 1) void column_fft_kernel (float2* data, // (no __restrict__: the source below may be this array)
                                                          const float2* __restrict__ twtab,
                                                          GenericPlan plan, // n = rows, frames = C
                                                          unsigned cols,
                                                          unsigned batch,
                                                          size_t array_stride, // float2 between consecutive arrays
                                                          unsigned lfs,
                                                          // round 4 (frames beyond 2^22 points as columns x rows, capi.cpp huge_transform): the arrays are read from
                                                          // `src_arg` (nullptr: in place) and every result is multiplied by W_N^(k col), N = rows * cols, formed from the
                                                          // two-level table W_N^j = tw_hi[j >> tw_shift] * tw_lo[j & mask] (nullptr: no twiddle)
                                                          const float2* src_arg,
                                                          const float2* __restrict__ tw_lo,
                                                          const float2* __restrict__ tw_hi,
                                                          unsigned tw_shift)
{
    const float2* srcdata = src_arg != nullptr ? src_arg : data;
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid0 = threadIdx.x;
    constexpr bool STATIC = ! std::is_void<SP>::value;
    static_assert (! STATIC || PIPE, "compile-time lists come with the pipelined form only");
    static_assert (! INPLACE || STATIC, "one-image tiles exist for compile-time lists only");
    const int rows = tile_frame_len<SP> (plan);
    if constexpr (STATIC)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) C * lfs; // (INPLACE: no second image, the twiddle copy sits here)
    float2* twl = smem + (size_t) (INPLACE ? 1 : 2) * C * lfs;
    if constexpr (PIPE)
    {
        lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], rows - 1, tid0);
    }
    const unsigned tiles = (cols + C - 1) / C;
    const size_t nwork = (size_t) batch * tiles;
    const int total = C * rows;
    constexpr int PF = tile_slots<SP, C, BLOCK> (kTilePrefetchA, INPLACE ? CFFT_TILE_INPLACE_PREFETCH : 1 << 20);
    float2 pre[PF];
    const auto request = [&] (size_t v) {
        int tid = tid0;
        if constexpr (INPLACE) // (per tile: nothing derived from the work-item index is lifted out of the tile loop and held -- the
            CFFT_LAUNDER_INDEX (tid); // one-image kernels spill otherwise; the two-image ones are 10-15 % faster with the offsets held)
        const size_t w = tile_xcd_order<1> (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const int valid = (int) min ((unsigned) C, cols - tile * C);
        const float2* src = srcdata + (size_t) f * array_stride + (size_t) tile * C;
        const auto slot = [&] (int u) {
            const int e = min (tid + u * BLOCK, total - 1);
            const int c = min (e % C, valid - 1), j = e / C;
            pre[u] = gload (&src[(size_t) j * cols + c]);
        };
        if (! STATIC && PF > 4 && total <= 4 * BLOCK) // (one branch around the group of loads, see pass_a_generic_kernel)
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                slot (u);
        }
        else
        {
#pragma unroll
            for (int u = 0; u < PF; ++u)
                slot (u);
        }
    };
    if constexpr (PIPE)
    {
        if (blockIdx.x < nwork)
            request (blockIdx.x);
    }
    for (size_t v = blockIdx.x; v < nwork; v += gridDim.x)
    {
        int tid = tid0;
        if constexpr (INPLACE)
            CFFT_LAUNDER_INDEX (tid);
        const size_t w = tile_xcd_order<1> (v, nwork);
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const int valid = (int) min ((unsigned) C, cols - tile * C);
        float2* base = data + (size_t) f * array_stride + (size_t) tile * C;
        const float2* sbase = srcdata + (size_t) f * array_stride + (size_t) tile * C;
        __syncthreads(); // previous tile fully stored
        if constexpr (PIPE)
        {
#pragma unroll
            for (int u = 0; u < PF; ++u)
            {
                const int e = tid + u * BLOCK;
                const int c = e % C, j = e / C;
                if (e < total)
                    bufA[c * lfs + j] = c < valid ? (SWAP ? make_float2 (pre[u].y, pre[u].x) : pre[u]) : make_float2 (0.f, 0.f);
            }
        }
        for (int e0 = tid + (PIPE ? PF * BLOCK : 0); e0 < total; e0 += BLOCK * kTileLoadsInFlight)
        {
            float2 x[kTileLoadsInFlight];
#pragma unroll
            for (int u = 0; u < kTileLoadsInFlight; ++u)
            {
                const int e = min (e0 + u * BLOCK, total - 1);
                const int c = min (e % C, valid - 1), j = e / C;
                x[u] = gload (&sbase[(size_t) j * cols + c]);
            }
#pragma unroll
            for (int u = 0; u < kTileLoadsInFlight; ++u)
            {
                const int e = e0 + u * BLOCK;
                const int c = e % C, j = e / C;
                if (e < total)
                    bufA[c * lfs + j] = c < valid ? (SWAP ? make_float2 (x[u].y, x[u].x) : x[u]) : make_float2 (0.f, 0.f);
            }
        }
        __syncthreads();
        const float2* res;
        if constexpr (PIPE)
        {
            if (v + gridDim.x < nwork)
                request (v + gridDim.x); // another tile: never the lines this one is about to store
            if constexpr (INPLACE)
            {
                static_tile_inplace_core<SP, BLOCK, C, SP_N_OR_ZERO<SP>::value + 1> (twl, bufA, tid);
                res = bufA;
            }
            else if constexpr (STATIC)
                res = static_core<SP, BLOCK, false, false, false, 0, false, SP_N_OR_ZERO<SP>::value + 1> (twl, bufA, bufB, nullptr, nullptr, C, tid, 0u, 0u);
            else
                res = generic_core<BLOCK, false, false, false> (plan, twl, bufA, bufB, nullptr, nullptr, C, tid, lfs, plan.tw_offset[0]);
        }
        else
            res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, C, tid, lfs);
        if (tw_lo == nullptr)
        {
#pragma unroll 4
            for (int e = tid; e < total; e += BLOCK)
            {
                const int c = e % C, k = e / C;
                if (c < valid)
                {
                    const float2 y = res[c * lfs + k];
                    gstore (&base[(size_t) k * cols + c], SWAP ? make_float2 (y.y, y.x) : y);
                }
            }
        }
        else
        {
            // ... times W_N^(k col): the forward twiddle in BOTH directions -- the backward transform is this forward algorithm on
            // swapped data, and the swap back happens behind the product
            const unsigned col0 = tile * C;
            const unsigned mask = (1u << tw_shift) - 1u;
#pragma unroll 2
            for (int e = tid; e < total; e += BLOCK)
            {
                const int c = e % C, k = e / C;
                if (c < valid)
                {
                    const unsigned jj = (unsigned) k * (col0 + (unsigned) c); // < N <= 2^27
                    const float2 w = cmul (tw_lo[jj & mask], tw_hi[jj >> tw_shift]);
                    const float2 y = cmul (res[c * lfs + k], w);
                    gstore (&base[(size_t) k * cols + c], SWAP ? make_float2 (y.y, y.x) : y);
                }
            }
        }
    }
}
} // namespace cfft
