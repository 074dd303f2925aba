// LDS-resident Stockham (autosort) FFT core for one frame of N <= 4096 complex points.
//
// A "team" of T work-items owns one frame; every work-item keeps R = N/T complex points in VGPRs.
// The transform is  N = r0 * r1 * ... (radices <= 16); stage s with radix r and p = r0*...*r(s-1):
//
//     butterfly i in [0, N/r):   k = i mod p,  j = (i - k) * r + k
//         u[a]  = X[i + a*N/r]                 a = 0..r-1      (stride N/r: lane-contiguous)
//         u[a] *= W_{p*r}^{a*k}
//         v     = DFT_r(u)
//         Y[j + b*p] = v[b]                    b = 0..r-1
//
// Stage 0 reads straight from global memory, the last stage writes straight to global memory
// (both lane-contiguous), the stages in between exchange through a padded LDS image.
// This is the MI355X counterpart of the reference's pass driver cfftf1_ps
// (simd/chowdsp_fft_impl_sse.cpp:372-432): same autosort idea, but the ping-pong buffers are LDS,
// not memory, the V-lane finalize (`pffft_cplx_finalize`, :434-496) and `pffft_zreorder` (:1466-1515)
// have no counterpart because the index algebra already ends in natural order.
//
// Per-stage twiddles W_{p*r}^{a*k} depend only on (work-item, butterfly slot), not on the frame, so a
// persistent team loads them ONCE into VGPRs and reuses them for every frame it transforms.
#pragma once
#include "butterflies.hpp"

#include <type_traits>

namespace cfft
{
// ---- compile-time plan -------------------------------------------------------------------------
template <int N_, int T_, int R0, int R1 = 1, int R2 = 1, int R3 = 1>
struct Plan
{
    static constexpr int N = N_;
    static constexpr int T = T_;           // work-items per frame
    static constexpr int R = N_ / T_;      // points per work-item
    static constexpr int NS = (R1 == 1 ? 1 : (R2 == 1 ? 2 : (R3 == 1 ? 3 : 4)));
    static constexpr int radix (int s) { return s == 0 ? R0 : (s == 1 ? R1 : (s == 2 ? R2 : R3)); }
    static constexpr int prod (int s) // product of radices before stage s
    {
        int p = 1;
        for (int i = 0; i < s; ++i)
            p *= radix (i);
        return p;
    }
    // offset (in float2) of stage s inside the twiddle table: stage s>=1 stores (r-1)*p entries [a-1][k]
    static constexpr int tw_offset (int s)
    {
        int o = 0;
        for (int i = 1; i < s; ++i)
            o += (radix (i) - 1) * prod (i);
        return o;
    }
    static constexpr int tw_total = tw_offset (NS);
    // twiddle registers per work-item
    static constexpr int tw_regs (int s) { return (radix (s) - 1) * (R / radix (s)); }
    static_assert (R0 * R1 * R2 * R3 == N_, "radices must multiply to N");
    static_assert (N_ % T_ == 0, "T must divide N");
    static_assert (R % R0 == 0 && R % R1 == 0 && R % R2 == 0 && R % R3 == 0, "each radix must divide R");
};

// LDS padding: one float2 of padding after every 16 (keeps b64 accesses conflict-light for both the
// stride-r writes of stage 0 and the p-run writes of later stages; see DESIGN.md "LDS image").
__device__ __forceinline__ constexpr int lds_phys (int idx) { return idx + (idx >> 4); }
template <int N>
struct LdsFrame
{
    static constexpr int floats2 = N + (N >> 4);
};

template <int T>
__device__ __forceinline__ void team_sync()
{
    if constexpr (T > 64)
    {
        __syncthreads();
    }
    else
    {
        // the whole team lives in one wavefront: LDS operations of a wave execute in order, only the
        // compiler has to be kept from moving them across this point.
        __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Which butterfly of a stage a work-item's slot q computes.  PlainMap: i = t + T q (lane-contiguous: what the stages that
// touch global memory need).  A stage that exchanges through LDS on both sides may use any bijection instead -- the fused
// real convolution assigns its LAST stage so that a bin and its mirror bin end up in the same work-item (RealPairMap,
// conv_kernels.hpp).  MapOf<P, S, MapLast> = MapLast for the last stage, PlainMap otherwise.
struct PlainMap
{
    template <class P>
    static __device__ __forceinline__ constexpr int index (int t, int q) { return t + P::T * q; }
};
template <class P, int S, class MapLast>
using MapOf = std::conditional_t<S == P::NS - 1, MapLast, PlainMap>;

// ---- twiddle registers -------------------------------------------------------------------------
template <class P>
struct TwiddleRegs
{
    // worst case: NS-1 twiddled stages, each needs < R registers
    float2 w[(P::NS > 1 ? P::NS - 1 : 1)][P::R];
};

// The same with the twiddles of stage 1 read from an LDS table at their use instead of living in registers (round 4: the
// fused real convolution trades these (r1 - 1) * Q1 register pairs for its resident pair-map coefficients).  Stage 1's
// twiddles depend on k = i mod r0 only: a table of (r1 - 1) * r0 entries [a - 1][k] per workgroup, read as broadcasts.
template <class P>
struct TwiddleRegsL1
{
    const float2* l1; // (plans with at least three stages: stage 1 must not be the last one, whose assignment may be mapped)
    float2 w[(P::NS > 1 ? P::NS - 1 : 1)][P::R]; // row 0 stays unused (never loaded: the optimiser drops it)
};
template <class P, int S>
__device__ __forceinline__ float2 tw_get (const TwiddleRegs<P>& tw, int q, int a, int)
{
    return tw.w[S > 0 ? S - 1 : 0][q * (P::radix (S) - 1) + (a - 1)];
}
template <class P, int S>
__device__ __forceinline__ float2 tw_get (const TwiddleRegsL1<P>& tw, int q, int a, int t)
{
    if constexpr (S == 1)
        return tw.l1[(a - 1) * P::prod (1) + ((t + P::T * q) & (P::prod (1) - 1))];
    else
        return tw.w[S > 0 ? S - 1 : 0][q * (P::radix (S) - 1) + (a - 1)];
}
template <class P, int T_THREADS>
__device__ __forceinline__ void fill_l1_table (float2* __restrict__ l1, const float2* __restrict__ table, int tid)
{
    constexpr int n = (P::radix (1) - 1) * P::prod (1);
    for (int j = tid; j < n; j += T_THREADS)
        l1[j] = table[P::tw_offset (1) + j];
}

template <class P, int S, class Map = PlainMap, class TW = TwiddleRegs<P>>
__device__ __forceinline__ void load_twiddles_stage (TW& tw, const float2* __restrict__ table, int t)
{
    constexpr int r = P::radix (S);
    constexpr int p = P::prod (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        const int i = Map::template index<P> (t, q);
        const int k = i & (p - 1);
#pragma unroll
        for (int a = 1; a < r; ++a)
        {
#if defined(CFFT_EXPERIMENT_FEW_TWIDDLES) && CFFT_EXPERIMENT_FEW_TWIDDLES
            // TIMING EXPERIMENT ONLY (wrong results): one twiddle stands in for the stage's fifteen, which frees ~28 VGPRs of the
            // N = 4096 plan: what would a third workgroup per CU be worth if the registers were there?
            if constexpr (P::N == 4096 && S == 2)
            {
                tw.w[S - 1][q * (r - 1) + (a - 1)] = table[P::tw_offset (S) + k];
                continue;
            }
#endif
            tw.w[S - 1][q * (r - 1) + (a - 1)] = table[P::tw_offset (S) + (a - 1) * p + k];
        }
    }
}

template <class P, class MapLast = PlainMap>
__device__ __forceinline__ void load_twiddles (TwiddleRegsL1<P>& tw, const float2* __restrict__ table, int t, const float2* l1)
{
    tw.l1 = l1;
    if constexpr (P::NS > 2) load_twiddles_stage<P, 2, MapOf<P, 2, MapLast>> (tw, table, t);
    if constexpr (P::NS > 3) load_twiddles_stage<P, 3, MapOf<P, 3, MapLast>> (tw, table, t);
}
template <class P, class MapLast = PlainMap>
__device__ __forceinline__ void load_twiddles (TwiddleRegs<P>& tw, const float2* __restrict__ table, int t)
{
    if constexpr (P::NS > 1) load_twiddles_stage<P, 1, MapOf<P, 1, MapLast>> (tw, table, t);
    if constexpr (P::NS > 2) load_twiddles_stage<P, 2, MapOf<P, 2, MapLast>> (tw, table, t);
    if constexpr (P::NS > 3) load_twiddles_stage<P, 3, MapOf<P, 3, MapLast>> (tw, table, t);
}

// ---- one stage ---------------------------------------------------------------------------------
// compute part: twiddle (S>0) + DFT_r on every butterfly slot of this work-item
// w: this stage's twiddles of this work-item, [q * (r-1) + (a-1)] (unused for S == 0)
template <class P, int S>
__device__ __forceinline__ void stage_compute_row (float2 (&v)[P::R], const float2 (&w)[P::R])
{
    constexpr int r = P::radix (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        float2 u[r];
#pragma unroll
        for (int a = 0; a < r; ++a)
            u[a] = v[q * r + a];
        if constexpr (S > 0)
        {
#pragma unroll
            for (int a = 1; a < r; ++a)
                u[a] = cmul (u[a], w[q * (r - 1) + (a - 1)]);
        }
        fft_reg<r> (u);
#pragma unroll
        for (int a = 0; a < r; ++a)
            v[q * r + a] = u[a];
    }
}

template <class P, int S>
__device__ __forceinline__ void stage_compute (float2 (&v)[P::R], const TwiddleRegs<P>& tw)
{
    stage_compute_row<P, S> (v, tw.w[S > 0 ? S - 1 : 0]);
}
// ... with the twiddles through tw_get (registers, or stage 1 from its LDS table)
template <class P, int S, class TW>
__device__ __forceinline__ void stage_compute_tw (float2 (&v)[P::R], const TW& tw, int t)
{
    constexpr int r = P::radix (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        float2 u[r];
#pragma unroll
        for (int a = 0; a < r; ++a)
            u[a] = v[q * r + a];
        if constexpr (S > 0)
        {
#pragma unroll
            for (int a = 1; a < r; ++a)
                u[a] = cmul (u[a], tw_get<P, S> (tw, q, a, t));
        }
        fft_reg<r> (u);
#pragma unroll
        for (int a = 0; a < r; ++a)
            v[q * r + a] = u[a];
    }
}

// scatter the outputs of stage S into the LDS image (Y[j + b*p])
template <class P, int S, class Map = PlainMap>
__device__ __forceinline__ void stage_write_lds (const float2 (&v)[P::R], float2* __restrict__ lds, int t)
{
    constexpr int r = P::radix (S);
    constexpr int p = P::prod (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        const int i = Map::template index<P> (t, q);
        const int k = i & (p - 1);
        const int j = (i - k) * r + k;
#pragma unroll
        for (int b = 0; b < r; ++b)
            lds[lds_phys (j + b * p)] = v[q * r + b];
    }
}

// gather the inputs of stage S from the LDS image (X[i + a*N/r])
template <class P, int S, class Map = PlainMap>
__device__ __forceinline__ void stage_read_lds (float2 (&v)[P::R], const float2* __restrict__ lds, int t)
{
    constexpr int r = P::radix (S);
    constexpr int Q = P::R / r;
    constexpr int m = P::N / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        const int i = Map::template index<P> (t, q);
#pragma unroll
        for (int a = 0; a < r; ++a)
            v[q * r + a] = lds[lds_phys (i + a * m)];
    }
}

// ---- the TRANSPOSED network (round 4) ---------------------------------------------------------------------------
// The DFT matrix is symmetric, so the transpose of the network above -- every stage run backwards: gather where it
// scattered, DFT_r (symmetric itself), multiply by the SAME twiddles after the butterfly instead of before, scatter where
// it gathered -- computes the same transform, but takes its operands in the forward network's OUTPUT arrangement (slot
// q r + b of the last stage = index i_q + b N / r) and leaves its results in the forward network's INPUT arrangement
// (input_index).  A kernel that runs a forward transform, works on the spectrum in registers and transforms back
// (the fused convolutions) therefore needs no re-ordering pass through LDS between the two transforms, and both share
// one set of twiddle registers.  Same LDS access patterns as the forward stages (mirrored), hence the same bank behaviour.
template <class P, int S, class TW>
__device__ __forceinline__ void stage_compute_t (float2 (&v)[P::R], const TW& tw, int t)
{
    constexpr int r = P::radix (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        float2 u[r];
#pragma unroll
        for (int a = 0; a < r; ++a)
            u[a] = v[q * r + a];
        fft_reg<r> (u);
        if constexpr (S > 0)
        {
#pragma unroll
            for (int a = 1; a < r; ++a)
                u[a] = cmul (u[a], tw_get<P, S> (tw, q, a, t));
        }
#pragma unroll
        for (int a = 0; a < r; ++a)
            v[q * r + a] = u[a];
    }
}
// transposed stage S, output side: X'[i + a N / r] = v[q r + a]   (mirror of stage_read_lds)
template <class P, int S, class Map = PlainMap>
__device__ __forceinline__ void stage_write_lds_t (const float2 (&v)[P::R], float2* __restrict__ lds, int t)
{
    constexpr int r = P::radix (S);
    constexpr int Q = P::R / r;
    constexpr int m = P::N / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        const int i = Map::template index<P> (t, q);
#pragma unroll
        for (int a = 0; a < r; ++a)
            lds[lds_phys (i + a * m)] = v[q * r + a];
    }
}
// transposed stage S, input side: v[q r + b] = Y'[j + b p]   (mirror of stage_write_lds)
template <class P, int S, class Map = PlainMap>
__device__ __forceinline__ void stage_read_lds_t (float2 (&v)[P::R], const float2* __restrict__ lds, int t)
{
    constexpr int r = P::radix (S);
    constexpr int p = P::prod (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        const int i = Map::template index<P> (t, q);
        const int k = i & (p - 1);
        const int j = (i - k) * r + k;
#pragma unroll
        for (int b = 0; b < r; ++b)
            v[q * r + b] = lds[lds_phys (j + b * p)];
    }
}

// Index (within the frame) of register slot `slot` of work-item t BEFORE stage 0 (input side).
template <class P>
__device__ __forceinline__ constexpr int input_index (int t, int slot)
{
    constexpr int r = P::radix (0);
    return (t + P::T * (slot / r)) + (slot % r) * (P::N / r);
}
// Index (within the frame) of register slot `slot` of work-item t AFTER the last stage (output side).
template <class P>
__device__ __forceinline__ constexpr int output_index (int t, int slot)
{
    constexpr int r = P::radix (P::NS - 1);
    return (t + P::T * (slot / r)) + (slot % r) * (P::N / r);
}

// widest team whose REAL kernels (r2c / c2r) go through the staging area: see Geometry::staged_real (plans.hpp).
// Measured on MI355X (fwd+inv): real N=256 (16 lanes) 4.63 -> 5.32 TB/s, N=512 (32 lanes) +2..5 %, 64 lanes: N=1024 -2 %,
// N=2048 -26 % (the staging area costs occupancy there).
#ifndef CFFT_REAL_STAGED_MAX_T
#define CFFT_REAL_STAGED_MAX_T 32
#endif

// ---- staged global I/O for narrow teams (T < 16): the workgroup's FPB contiguous frames <-> LDS, 16 B per lane ----
template <int THREADS, int ELEMS>
__device__ __forceinline__ void stage_in (const float2* __restrict__ src, float2* __restrict__ stage, unsigned valid, int tid)
{
    const float4* __restrict__ s4 = reinterpret_cast<const float4*> (src);
    float4* __restrict__ d4 = reinterpret_cast<float4*> (stage);
#pragma unroll 4
    for (int e = tid; e < ELEMS / 2; e += THREADS)
        d4[e] = (2u * (unsigned) e < valid) ? gload4 (&s4[e]) : make_float4 (0.f, 0.f, 0.f, 0.f);
}
template <int THREADS, int ELEMS>
__device__ __forceinline__ void stage_out (float2* __restrict__ dst, const float2* __restrict__ stage, unsigned valid, int tid)
{
    float4* __restrict__ d4 = reinterpret_cast<float4*> (dst);
    const float4* __restrict__ s4 = reinterpret_cast<const float4*> (stage);
#pragma unroll 4
    for (int e = tid; e < ELEMS / 2; e += THREADS)
        if (2u * (unsigned) e < valid)
            gstore4 (&d4[e], s4[e]);
}

// Frame strides of a kernel.  Packed batches (STRIDED == false) use the compile-time frame length, so
// their code is exactly the packed-only kernel.  For strided batches the launchers guarantee stride <= kMaxFrameStride
// (2^24 float2, launch.hpp); telling the compiler lets it prove that team * stride + index cannot wrap in 32 bits, so
// that per-access constants fold into the instruction's immediate offset instead of costing one address VGPR each.
#define CFFT_FRAME_STRIDES(STRIDED_, N_)                                               \
    __builtin_assume (in_stride_arg <= (1u << 24));                                    \
    __builtin_assume (out_stride_arg <= (1u << 24));                                   \
    const unsigned in_stride = (STRIDED_) ? in_stride_arg : (unsigned) (N_);           \
    const unsigned out_stride = (STRIDED_) ? out_stride_arg : (unsigned) (N_)

// Strided batches: consecutive frames are `stride` float2 apart in global memory (stride >= N, even)
// while the staging area keeps them packed.  N is a power of two >= 16, so a float4 never straddles two frames.
template <int THREADS, int ELEMS, int N>
__device__ __forceinline__ void stage_in_strided (const float2* __restrict__ src, unsigned stride, float2* __restrict__ stage, unsigned valid, int tid)
{
    if (stride == (unsigned) N)
        return stage_in<THREADS, ELEMS> (src, stage, valid, tid);
    float4* __restrict__ d4 = reinterpret_cast<float4*> (stage);
#pragma unroll 4
    for (int e = tid; e < ELEMS / 2; e += THREADS)
    {
        const unsigned e2 = 2u * (unsigned) e, f = e2 / (unsigned) N, o = e2 % (unsigned) N;
        d4[e] = (e2 < valid) ? gload4 (reinterpret_cast<const float4*> (src + (size_t) f * stride + o)) : make_float4 (0.f, 0.f, 0.f, 0.f);
    }
}
template <int THREADS, int ELEMS, int N>
__device__ __forceinline__ void stage_out_strided (float2* __restrict__ dst, unsigned stride, const float2* __restrict__ stage, unsigned valid, int tid)
{
    if (stride == (unsigned) N)
        return stage_out<THREADS, ELEMS> (dst, stage, valid, tid);
    const float4* __restrict__ s4 = reinterpret_cast<const float4*> (stage);
#pragma unroll 4
    for (int e = tid; e < ELEMS / 2; e += THREADS)
    {
        const unsigned e2 = 2u * (unsigned) e, f = e2 / (unsigned) N, o = e2 % (unsigned) N;
        if (e2 < valid)
            gstore4 (reinterpret_cast<float4*> (dst + (size_t) f * stride + o), s4[e]);
    }
}

// ---- round 5: the staged copy-in one group AHEAD --------------------------------------------------------------------------
// The staged kernels ran copy-in -> barrier -> stages -> barrier -> copy-out per group: while a workgroup computes it has nothing
// in flight, and the narrow plans (complex N <= 64, real N <= 512) sat at a plain copy's rate (0.64-0.66 of 8 TB/s) next to 0.70-0.74
// for their neighbours with direct accesses.  Here the NEXT group's 16-byte loads are issued into registers (K = ELEMS / 2 / THREADS
// float4 per work-item, 2-4 for these plans) right after this group's operands have been put into the staging area, fly during
// the stages and the copy-out, and are written to LDS at the top of the next iteration.  Same accesses, same barriers.
// Per plan, from same-box A/B (profiles/r05_staged_prefetch_ab.txt, 512 MiB batches, two rounds): complex N = 32 +2.5 %, real N = 64 / 128 /
// 256 / 512 +4.8 / +2.7 / +2.8 / +1.6 %; complex N = 16 / 64 and real N = 32 lose 1-2.5 % (their groups are 8-16 KiB: the extra
// registers and the second wait cost more than the latency they hide) and keep the plain copy-in.
// CFFT_STAGED_PREFETCH: 2 (default) = that selection, 1 = every staged plan, 0 = none.
#ifndef CFFT_STAGED_PREFETCH
#define CFFT_STAGED_PREFETCH 2
#endif
template <int N, bool REAL>
static constexpr bool kStagedPrefetch = CFFT_STAGED_PREFETCH == 1 || (CFFT_STAGED_PREFETCH == 2 && (REAL ? N >= 32 : N == 32)); // N: the plan's complex length
template <int THREADS, int ELEMS, int N>
struct StagePrefetch
{
    static constexpr int K = (ELEMS / 2 + THREADS - 1) / THREADS;
    float4 r[K];
    __device__ __forceinline__ void fetch (const float2* __restrict__ src, unsigned stride, unsigned valid, int tid)
    {
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
            const unsigned e = (unsigned) tid + (unsigned) j * THREADS;
            const unsigned e2 = 2u * e;
            if (stride == (unsigned) N)
                r[j] = (e < (unsigned) (ELEMS / 2) && e2 < valid) ? gload4 (reinterpret_cast<const float4*> (src) + e) : make_float4 (0.f, 0.f, 0.f, 0.f);
            else
            {
                const unsigned f = e2 / (unsigned) N, o = e2 % (unsigned) N;
                r[j] = (e < (unsigned) (ELEMS / 2) && e2 < valid) ? gload4 (reinterpret_cast<const float4*> (src + (size_t) f * stride + o)) : make_float4 (0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void put (float2* __restrict__ stage, int tid) const
    {
        float4* __restrict__ d4 = reinterpret_cast<float4*> (stage);
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
            const unsigned e = (unsigned) tid + (unsigned) j * THREADS;
            if (e < (unsigned) (ELEMS / 2))
                d4[e] = r[j];
        }
    }
};

// Stages 0..NS-1 on registers v (already loaded in input_index order); leaves the result in
// output_index order.  `lds` is this team's private LDS image.
// ---- the exchange of the two shortest plans inside the wavefront (round 4 A/B, CFFT_DPP_QUAD) -----------------------------
// N = 16 (4 lanes x 4 points, 4 x 4) and N = 32 (4 lanes x 8 points, 8 x 4): the one exchange between the two stages is a
// 4 x 4 transpose across the four lanes of a team -- lane i, slot a  <-  lane a, slot i -- and a team of four is a DPP quad.
// Two rounds of v_mov_b32 dpp quad_perm (lane ^ 1, lane ^ 2) with selects instead of 4 (8) ds_write_b64 + 4 (8) ds_read_b64
// and a wave-local barrier: the north star's "wavefront shuffle / DPP" for the cross-lane step.  Measured on MI355X
// (profiles/r04_dpp_small_n_ab.txt, fwd + inv): complex N = 16 / 32 +0.9 / +1.0 %, real N = 32 +4.2 %, real N = 64 +0.2 %,
// nothing slower: on (1); 0 = the LDS exchange.
#ifndef CFFT_DPP_QUAD
#define CFFT_DPP_QUAD 1
#endif
__device__ __forceinline__ float dpp_quad_xor1 (float x)
{
    return __int_as_float (__builtin_amdgcn_mov_dpp (__float_as_int (x), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2 (float x)
{
    return __int_as_float (__builtin_amdgcn_mov_dpp (__float_as_int (x), 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true));
}
// w[a] (lane i of the quad)  <-  w[i] (lane a of the quad), a, i < 4
__device__ __forceinline__ void quad_transpose4 (float2& w0, float2& w1, float2& w2, float2& w3, int lane)
{
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    {
        const float2 s01 = b0 ? w0 : w1, s23 = b0 ? w2 : w3;
        const float2 r01 = make_float2 (dpp_quad_xor1 (s01.x), dpp_quad_xor1 (s01.y)), r23 = make_float2 (dpp_quad_xor1 (s23.x), dpp_quad_xor1 (s23.y));
        w0 = b0 ? r01 : w0;
        w1 = b0 ? w1 : r01;
        w2 = b0 ? r23 : w2;
        w3 = b0 ? w3 : r23;
    }
    {
        const float2 s02 = b1 ? w0 : w2, s13 = b1 ? w1 : w3;
        const float2 r02 = make_float2 (dpp_quad_xor2 (s02.x), dpp_quad_xor2 (s02.y)), r13 = make_float2 (dpp_quad_xor2 (s13.x), dpp_quad_xor2 (s13.y));
        w0 = b1 ? r02 : w0;
        w2 = b1 ? w2 : r02;
        w1 = b1 ? r13 : w1;
        w3 = b1 ? w3 : r13;
    }
}
template <class P>
static constexpr bool kQuadExchange = CFFT_DPP_QUAD != 0 && P::T == 4 && P::NS == 2 && P::radix (1) == 4 && (P::N == 16 || P::N == 32);

// MapLast: the butterfly assignment of the LAST stage (its twiddles must have been loaded with the same map); with
// anything but PlainMap the result is in that map's arrangement instead of output_index order.
template <class P, class MapLast = PlainMap, class TW = TwiddleRegs<P>>
__device__ __forceinline__ void fft_core (float2 (&v)[P::R], const TW& tw, float2* __restrict__ lds, int t)
{
    stage_compute_tw<P, 0> (v, tw, t);
    if constexpr (kQuadExchange<P> && std::is_same<MapLast, PlainMap>::value)
    {
        // stage-0 output Y[r0 t' + b] = slot b of lane t'; stage-1 butterfly i = t + 4 q wants X[i + a N / 4] = slot t + 4 q of lane a
#pragma unroll
        for (int q = 0; q < P::R / 4; ++q)
            quad_transpose4 (v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], t);
        stage_compute_tw<P, 1> (v, tw, t);
        return;
    }
    if constexpr (P::NS > 1)
    {
        stage_write_lds<P, 0> (v, lds, t);
        team_sync<P::T>();
        stage_read_lds<P, 1, MapOf<P, 1, MapLast>> (v, lds, t);
        stage_compute_tw<P, 1> (v, tw, t);
    }
    if constexpr (P::NS > 2)
    {
        team_sync<P::T>();
        stage_write_lds<P, 1> (v, lds, t);
        team_sync<P::T>();
        stage_read_lds<P, 2, MapOf<P, 2, MapLast>> (v, lds, t);
        stage_compute_tw<P, 2> (v, tw, t);
    }
    if constexpr (P::NS > 3)
    {
        team_sync<P::T>();
        stage_write_lds<P, 2> (v, lds, t);
        team_sync<P::T>();
        stage_read_lds<P, 3, MapOf<P, 3, MapLast>> (v, lds, t);
        stage_compute_tw<P, 3> (v, tw, t);
    }
}

// The transposed network: v in the arrangement fft_core<P, MapLast> leaves (output_index order for PlainMap), result in
// input_index order.  The caller makes sure that every read of the image by the previous phase has completed (team_sync)
// before calling -- the first thing this does with LDS is write.
template <class P, class MapLast = PlainMap, class TW = TwiddleRegs<P>>
__device__ __forceinline__ void fft_core_transposed (float2 (&v)[P::R], const TW& tw, float2* __restrict__ lds, int t)
{
    stage_compute_t<P, P::NS - 1> (v, tw, t);
    if constexpr (P::NS > 3)
    {
        stage_write_lds_t<P, 3, MapOf<P, 3, MapLast>> (v, lds, t);
        team_sync<P::T>();
        stage_read_lds_t<P, 2> (v, lds, t);
        stage_compute_t<P, 2> (v, tw, t);
        team_sync<P::T>();
    }
    if constexpr (P::NS > 2)
    {
        stage_write_lds_t<P, 2, MapOf<P, 2, MapLast>> (v, lds, t);
        team_sync<P::T>();
        stage_read_lds_t<P, 1> (v, lds, t);
        stage_compute_t<P, 1> (v, tw, t);
        team_sync<P::T>();
    }
    if constexpr (P::NS > 1)
    {
        stage_write_lds_t<P, 1, MapOf<P, 1, MapLast>> (v, lds, t);
        team_sync<P::T>();
        stage_read_lds_t<P, 0> (v, lds, t);
        stage_compute_t<P, 0> (v, tw, t);
    }
}

// ---- the same core with STREAMED twiddles ------------------------------------------------------------------------
// For the N = 16384 plan: 1024 work-items x 16 points leave 128 VGPRs per work-item, not enough to keep every stage's
// twiddles resident.  One stage's twiddles at a time are fetched from the (L2-resident) table while the previous
// stage's outputs travel through LDS.  Two things keep the register count at 120 (tools/kres.py), both needed:
//  * the caller launders the work-item index once per frame (CFFT_LAUNDER_INDEX): otherwise the ~100 loop-invariant
//    LDS / table offsets of the four stages are hoisted out of the persistent loop and held (then spilled) for good;
//  * scheduling fences between the phases, so that the next phase's loads are not hoisted over this phase's registers.
#define CFFT_LAUNDER_INDEX(var) asm volatile ("" : "+v"(var))
#define CFFT_PHASE_FENCE() __builtin_amdgcn_sched_barrier (0)

template <class P, int S>
__device__ __forceinline__ void fetch_stage_twiddles (float2 (&w)[P::R], const float2* __restrict__ table, int t)
{
    constexpr int r = P::radix (S);
    constexpr int p = P::prod (S);
    constexpr int Q = P::R / r;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
        const int k = (t + P::T * q) & (p - 1);
#pragma unroll
        for (int a = 1; a < r; ++a)
            w[q * (r - 1) + (a - 1)] = table[P::tw_offset (S) + (a - 1) * p + k];
    }
}

template <class P, int S>
__device__ __forceinline__ void streamed_stage (float2 (&v)[P::R], float2 (&w)[P::R], const float2* __restrict__ table,
                                                float2* __restrict__ lds, int t)
{
    // v holds stage S-1's outputs: exchange them through LDS while stage S's twiddles arrive, then stage S
    CFFT_PHASE_FENCE();
    stage_write_lds<P, S - 1> (v, lds, t);
    CFFT_PHASE_FENCE();
    fetch_stage_twiddles<P, S> (w, table, t);
    team_sync<P::T>();
    stage_read_lds<P, S> (v, lds, t);
    CFFT_PHASE_FENCE();
    stage_compute_row<P, S> (v, w);
    CFFT_PHASE_FENCE();
}

template <class P>
__device__ __forceinline__ void fft_core_streamed (float2 (&v)[P::R], const float2* __restrict__ table, float2* __restrict__ lds, int t)
{
    float2 w[P::R];
    stage_compute_row<P, 0> (v, w);
    if constexpr (P::NS > 1)
        streamed_stage<P, 1> (v, w, table, lds, t);
    if constexpr (P::NS > 2)
    {
        team_sync<P::T>();
        streamed_stage<P, 2> (v, w, table, lds, t);
    }
    if constexpr (P::NS > 3)
    {
        team_sync<P::T>();
        streamed_stage<P, 3> (v, w, table, lds, t);
    }
}
} // namespace cfft
