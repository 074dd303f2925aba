// Batched real transforms (N real points <-> packed half spectrum of N floats) on top of the
// M = N/2 point complex core.
//
// Replaces, for a batch, the REAL branches of the reference's pffft_transform_internal
// (simd/chowdsp_fft_impl_sse.cpp:1538-1542 forward: rfftf1_ps + pffft_real_finalize + zreorder;
// :1572-1576 backward: zreorder + pffft_real_preprocess + rfftb1_ps).  The reference runs FFTPACK
// real radix-4 passes on V lanes and then stitches the lanes together; here the frame is treated as
// M complex points z[n] = x[2n] + i*x[2n+1], transformed by the complex core, and split / merged with
// the textbook pair formula.  The frequency-domain layout is the reference's canonical packed one
// (confirmed by chowdsp_fft_juce/chowdsp_fft_juce.cpp:58-61,74):
//     [X0.re, X(N/2).re, X1.re, X1.im, ..., X(N/2-1).re, X(N/2-1).im]
//
// Forward:  Z = DFT_M(z);  for 0 < k < M/2, A = Z[k], B = conj(Z[M-k]):
//              E = (A + B)/2,  O = -i/2 * W_N^k * (A - B),   X[k] = E + O,  X[M-k] = conj(E - O)
//           X[0] = Re Z0 + Im Z0,  X[M] = Re Z0 - Im Z0,  X[M/2] = conj(Z[M/2])
// Backward (unscaled, BACKWARD(FORWARD(x)) = N x):  A = X[k], B = conj(X[M-k]):
//              E = A + B,  O = i * conj(W_N^k) * (A - B),   Z[k] = E + O,  Z[M-k] = conj(E - O)
//           Z[0] = (X0 + XM) + i (X0 - XM),  Z[M/2] = 2 conj(X[M/2]);   z = IDFT_M(Z) (unscaled)
#pragma once
#include "stockham.hpp"

namespace cfft
{
// the mirrored bins' stores of the one-pass real forward kernels: default cache policy for the 2048- and 4096-point cores (real N = 4096 / 8192: +5 %, the
// pieces of a line meet in L2), nontemporal like every other streaming store elsewhere (staged / rotated-lane forms do not come here; 8192 points and more:
// +-1 %).  profiles/r05_real_mirror_store.txt; CFFT_REAL_MIRROR_PLAIN = 0: nontemporal everywhere (rounds 1-4).
#ifndef CFFT_REAL_MIRROR_PLAIN
#define CFFT_REAL_MIRROR_PLAIN 1
#endif
template <int M_>
__device__ __forceinline__ void real_mirror_store (float2* p, float2 v)
{
    if (CFFT_REAL_MIRROR_PLAIN && (M_ == 2048 || M_ == 4096))
        *p = v;
    else
        gstore (p, v);
}

// the mirrored bins' LOADS of the power-of-two backward kernels stay nontemporal: the default policy measured 0 +- 0.6 % on them (round 5,
// profiles/r05_real_mirror_store.txt; the mixed-radix kernel with many short frames per workgroup is where it pays: generic_kernels.hpp)
__device__ __forceinline__ float2 real_mirror_load (const float2* p) { return gload (p); }

// natural-order LDS image <-> register order helpers
template <class P>
__device__ __forceinline__ void lds_store_output_order (const float2 (&v)[P::R], float2* __restrict__ lds, int t)
{
#pragma unroll
    for (int s = 0; s < P::R; ++s)
        lds[lds_phys (output_index<P> (t, s))] = v[s];
}
template <class P>
__device__ __forceinline__ void lds_load_input_order (float2 (&v)[P::R], const float2* __restrict__ lds, int t)
{
#pragma unroll
    for (int s = 0; s < P::R; ++s)
        v[s] = lds[lds_phys (input_index<P> (t, s))];
}

// per-work-item W_N^k registers for the pair stage: k = t + T*q, q < R/2
template <class P>
struct PairTwiddles
{
    float2 w[P::R / 2];
};
template <class P>
__device__ __forceinline__ void load_pair_twiddles (PairTwiddles<P>& pw, const float2* __restrict__ wtab, int t)
{
#pragma unroll
    for (int q = 0; q < P::R / 2; ++q)
        pw.w[q] = wtab[t + P::T * q];
}

// forward split of one pair: in A = Z[k], Zmk = Z[M-k]; out Xk, Xmk
__device__ __forceinline__ void real_split_pair (float2 A, float2 Zmk, float2 w, float2& Xk, float2& Xmk)
{
    const float2 B = make_float2 (Zmk.x, -Zmk.y);
    const float2 E = make_float2 (0.5f * (A.x + B.x), 0.5f * (A.y + B.y));
    const float2 D = make_float2 (0.5f * (A.x - B.x), 0.5f * (A.y - B.y));
    const float2 WD = cmul (D, w);           // W^k * (A-B)/2
    const float2 O = make_float2 (WD.y, -WD.x); // * (-i)
    Xk = make_float2 (E.x + O.x, E.y + O.y);
    Xmk = make_float2 (E.x - O.x, -(E.y - O.y));
}
// backward merge of one pair: in A = X[k], Xmk = X[M-k]; out Zk, Zmk
__device__ __forceinline__ void real_merge_pair (float2 A, float2 Xmk, float2 w, float2& Zk, float2& Zmk)
{
    const float2 B = make_float2 (Xmk.x, -Xmk.y);
    const float2 E = make_float2 (A.x + B.x, A.y + B.y);
    const float2 D = make_float2 (A.x - B.x, A.y - B.y);
    const float2 WD = cmulc (D, w);          // conj(W^k) * (A-B)
    const float2 O = make_float2 (-WD.y, WD.x); // * (+i)
    Zk = make_float2 (E.x + O.x, E.y + O.y);
    Zmk = make_float2 (E.x - O.x, -(E.y - O.y));
}

#ifndef CFFT_C2R512_WAVES5
#define CFFT_C2R512_WAVES5 1
#endif
#ifndef CFFT_REAL_ROTATE_LANE0
#define CFFT_REAL_ROTATE_LANE0 1
#endif

// ---- r2c -----------------------------------------------------------------------------------------
// P is the plan of the M = N/2 point complex core.
// LAYOUT 0: the packed spectrum.  LAYOUT 1 / 2: the JUCE engine's interleaved layout written by the pair stage itself
// (chowdsp_fft_juce/chowdsp_fft_juce.cpp:57-66): frames of 2 M complex slots, c[0] = (X0, 0), c[k] = X_k, c[M] = (X_{N/2}, 0),
// and for LAYOUT 2 also c[2 M - k] = conj X_k -- the same values through a different store pattern, in place of a second
// launch over the frames.
template <class P, int FPB, bool STRIDED, int LAYOUT = 0>
__global__ __launch_bounds__ (P::T* FPB) void r2c_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ twtab,
                                                         const float2* __restrict__ wtab,
                                                         unsigned batch,
                                                         unsigned in_stride_arg,
                                                         unsigned out_stride_arg)
{
    CFFT_FRAME_STRIDES (STRIDED, P::N);
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    PairTwiddles<P> pw;
    load_pair_twiddles<P> (pw, wtab, t);

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    const unsigned team_base = (unsigned) team * M; // inside the LDS staging area
    const unsigned team_in = (unsigned) team * in_stride, team_out = (unsigned) team * out_stride; // in global memory
    constexpr bool STAGED = (P::T <= CFFT_REAL_STAGED_MAX_T); // narrow teams: whole-workgroup coalesced copies through LDS (plans.hpp)
    float2* stage = smem + FPB * LdsFrame<M>::floats2;
    StagePrefetch<P::T * FPB, FPB * M, M> ahead; // (STAGED only) the next group's operands, requested a group ahead (stockham.hpp)
    if constexpr (STAGED && kStagedPrefetch<M, true>)
        if (blockIdx.x < ngroups)
            ahead.fetch (in + (size_t) blockIdx.x * FPB * in_stride, in_stride, min ((unsigned) FPB, batch - blockIdx.x * FPB) * M, tid);
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const bool active = g * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) g * FPB * in_stride;
        float2* __restrict__ dst = out + (size_t) g * FPB * out_stride;
        const unsigned valid = min ((unsigned) FPB, batch - g * FPB) * M;

        float2 v[P::R];
        if constexpr (STAGED)
        {
            __syncthreads();
            if constexpr (kStagedPrefetch<M, true>)
            {
                ahead.put (stage, tid);
                const unsigned gn = g + gridDim.x;
                if (gn < ngroups)
                    ahead.fetch (in + (size_t) gn * FPB * in_stride, in_stride, min ((unsigned) FPB, batch - gn * FPB) * M, tid);
            }
            else
                stage_in_strided<P::T * FPB, FPB * M, M> (src, in_stride, stage, valid, tid);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = stage[team_base + (unsigned) input_index<P> (t, s)];
        }
        else
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = active ? gload (&src[team_in + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
        }

        team_sync<P::T>();
        fft_core<P> (v, tw, lds, t);
        if constexpr (P::NS > 1)
            team_sync<P::T>(); // last-stage reads done before the image is overwritten
        lds_store_output_order<P> (v, lds, t);
        team_sync<P::T>();

        float2 hi_k[(STAGED && LAYOUT == 2) ? P::R / 2 : 1], hi_p[(STAGED && LAYOUT == 2) ? P::R / 2 : 1]; // the conjugate half's values
        // One-wavefront teams (T = 64): the mirrored bins M - k of one store instruction are the 64 elements ENDING at a
        // multiple of 64 -- a 512-byte run that starts one element off a 128-byte line, five lines touched, two of them
        // for 8 bytes.  Lane 0's element is the odd one: it belongs to the aligned block that the NEXT pair index covers.
        // So lane 0 stores its mirrored bins one instruction late (same values, same addresses, another store instruction)
        // and every instruction writes one aligned 512-byte block (CFFT_REAL_ROTATE_LANE0 = 0: the plain order).
        constexpr bool ROT = CFFT_REAL_ROTATE_LANE0 && ! STAGED && LAYOUT != 2 && P::T == 64; // LAYOUT 1: the same mirrored family
        float2 xp_keep[ROT ? P::R / 2 : 1];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q; // 0 <= k < M/2
            const float2 A = lds[lds_phys (k)];
            // k == 0 pairs Z[0] with Z[M/2] (both handled specially)
            const float2 Zp = lds[lds_phys (k == 0 ? M / 2 : M - k)];
            float2 Xk, Xp;
            real_split_pair (A, Zp, pw.w[q], Xk, Xp);
            if (k == 0)
            {
                Xk = make_float2 (A.x + A.y, A.x - A.y); // [X0.re, X(N/2).re]
                Xp = make_float2 (Zp.x, -Zp.y);          // X[M/2] = conj(Z[M/2])
            }
            if constexpr (STAGED)
            {
                // LAYOUT 1 / 2: bins 0 .. M - 1 leave through the staged copy like the packed spectrum; bin M on its own
                // (LAYOUT 1) or with the conjugate half in a second staged copy below (LAYOUT 2)
                stage[team_base + (unsigned) k] = (LAYOUT != 0 && k == 0) ? make_float2 (Xk.x, 0.f) : Xk;
                stage[team_base + (unsigned) (k == 0 ? M / 2 : M - k)] = Xp;
                if (LAYOUT == 1 && k == 0 && active)
                    gstore (&dst[team_out + (unsigned) M], make_float2 (Xk.y, 0.f));
                if constexpr (LAYOUT == 2)
                {
                    hi_k[q] = (k == 0) ? make_float2 (Xk.y, 0.f) : make_float2 (Xp.x, -Xp.y);  // slot M + k = conj X_{M-k}
                    hi_p[q] = (k == 0) ? make_float2 (Xp.x, -Xp.y) : make_float2 (Xk.x, -Xk.y); // slot M + (M - k) = conj X_k
                }
            }
            else if (active)
            {
                const unsigned kp = (unsigned) (k == 0 ? M / 2 : M - k);
                if constexpr (ROT)
                {
                    if (LAYOUT == 1 && k == 0)
                    {
                        gstore (&dst[team_out], make_float2 (Xk.x, 0.f));
                        gstore (&dst[team_out + (unsigned) M], make_float2 (Xk.y, 0.f));
                    }
                    else
                        gstore (&dst[team_out + (unsigned) k], Xk);
                    xp_keep[q] = Xp;
                }
                else if constexpr (LAYOUT == 0)
                {
                    gstore (&dst[team_out + (unsigned) k], Xk);
                    real_mirror_store<M> (&dst[team_out + kp], Xp);
                }
                else
                {
                    if (k == 0)
                    {
                        gstore (&dst[team_out], make_float2 (Xk.x, 0.f));
                        gstore (&dst[team_out + (unsigned) M], make_float2 (Xk.y, 0.f));
                    }
                    else
                    {
                        gstore (&dst[team_out + (unsigned) k], Xk);
                        if constexpr (LAYOUT == 2)
                            gstore (&dst[team_out + (unsigned) (2 * M - k)], make_float2 (Xk.x, -Xk.y));
                    }
                    real_mirror_store<M> (&dst[team_out + kp], Xp);
                    if constexpr (LAYOUT == 2)
                        gstore (&dst[team_out + (unsigned) (2 * M) - kp], make_float2 (Xp.x, -Xp.y));
                }
            }
        }
        if constexpr (ROT)
        {
            if (active)
            {
                constexpr int H = P::R / 2;
#pragma unroll
                for (int q = 0; q < H; ++q)
                {
                    const int qn = (q + 1) % H;                                   // lane 0: the pair index after this one
                    const int kn = P::T * qn;                                     // ... its k (a multiple of 64; 0 for the last q)
                    const unsigned kp0 = (unsigned) (kn == 0 ? M / 2 : M - kn);  // ... and that pair's mirrored bin
                    const unsigned kp = (t == 0) ? kp0 : (unsigned) (M - (t + P::T * q));
                    const float2 val = (t == 0) ? xp_keep[qn] : xp_keep[q];
                    gstore (&dst[team_out + kp], val);
                }
            }
        }
        if constexpr (STAGED)
        {
            __syncthreads();
            stage_out_strided<P::T * FPB, FPB * M, M> (dst, out_stride, stage, valid, tid);
            if constexpr (LAYOUT == 2)
            {
                __syncthreads();
#pragma unroll
                for (int q = 0; q < P::R / 2; ++q)
                {
                    const int k = t + P::T * q;
                    stage[team_base + (unsigned) k] = hi_k[q];
                    stage[team_base + (unsigned) (k == 0 ? M / 2 : M - k)] = hi_p[q];
                }
                __syncthreads();
                stage_out_strided<P::T * FPB, FPB * M, M> (dst + M, out_stride, stage, valid, tid);
            }
        }
    }
}

// ---- software-pipelined variants ------------------------------------------------------------------------
// Same scheme as c2c_kernel_pf: the next group's loads are issued before this group's arithmetic and stores, and the
// first group is peeled so that the steady-state loop-top wait is vmcnt(#stores), never vmcnt(0).  To keep the load and
// store counts compile-time constants with several frames per workgroup, the main loop only walks FULL groups
// (batch / FPB of them); a ragged last group is handled once, predicated, by the workgroup whose turn it would be.
template <class P, bool PRED>
__device__ __forceinline__ void r2c_finish (float2 (&v)[P::R], const TwiddleRegs<P>& tw, const PairTwiddles<P>& pw,
                                            float2* __restrict__ lds, float2* __restrict__ dst, unsigned team_out, int t, bool active)
{
    constexpr int M = P::N;
    team_sync<P::T>();
    fft_core<P> (v, tw, lds, t);
    if constexpr (P::NS > 1)
        team_sync<P::T>();
    lds_store_output_order<P> (v, lds, t);
    team_sync<P::T>();
#pragma unroll
    for (int q = 0; q < P::R / 2; ++q)
    {
        const int k = t + P::T * q;
        const float2 A = lds[lds_phys (k)];
        const float2 Zp = lds[lds_phys (k == 0 ? M / 2 : M - k)];
        float2 Xk, Xp;
        real_split_pair (A, Zp, pw.w[q], Xk, Xp);
        if (k == 0)
        {
            Xk = make_float2 (A.x + A.y, A.x - A.y);
            Xp = make_float2 (Zp.x, -Zp.y);
        }
        if (! PRED || active)
        {
            gstore (&dst[team_out + (unsigned) k], Xk);
            real_mirror_store<M> (&dst[team_out + (unsigned) (k == 0 ? M / 2 : M - k)], Xp);
        }
    }
}

template <class P, int FPB, bool STRIDED>
__global__ __launch_bounds__ (P::T* FPB) void r2c_kernel_pf (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ twtab,
                                                            const float2* __restrict__ wtab,
                                                            unsigned batch,
                                                            unsigned in_stride_arg,
                                                            unsigned out_stride_arg)
{
    CFFT_FRAME_STRIDES (STRIDED, P::N);
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;
    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    PairTwiddles<P> pw;
    load_pair_twiddles<P> (pw, wtab, t);

    const unsigned nfull = batch / FPB;
    const unsigned team_in = (unsigned) team * in_stride, team_out = (unsigned) team * out_stride; // in global memory
    float nre[P::R], nim[P::R];
    unsigned g = blockIdx.x;
    if (g < nfull)
    {
        {
            const float2* __restrict__ src = in + (size_t) g * FPB * in_stride;
            float2 v[P::R];
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = gload (&src[team_in + (unsigned) input_index<P> (t, s)]);
            const unsigned gn = g + gridDim.x;
            if (gn < nfull)
            {
                const float2* __restrict__ srcn = in + (size_t) gn * FPB * in_stride;
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_in + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            r2c_finish<P, false> (v, tw, pw, lds, out + (size_t) g * FPB * out_stride, team_out, t, true);
        }
        for (g += gridDim.x; g < nfull; g += gridDim.x)
        {
            float2 v[P::R];
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = make_float2 (nre[s], nim[s]);
            const unsigned gn = g + gridDim.x;
            if (gn < nfull)
            {
                const float2* __restrict__ srcn = in + (size_t) gn * FPB * in_stride;
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_in + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            r2c_finish<P, false> (v, tw, pw, lds, out + (size_t) g * FPB * out_stride, team_out, t, true);
        }
    }
    // ragged last group
    if (FPB > 1 && nfull * FPB < batch && blockIdx.x == nfull % gridDim.x)
    {
        const bool active = nfull * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) nfull * FPB * in_stride;
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            v[s] = active ? gload (&src[team_in + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
        r2c_finish<P, true> (v, tw, pw, lds, out + (size_t) nfull * FPB * out_stride, team_out, t, active);
    }
}

// ---- c2r -----------------------------------------------------------------------------------------
// LAYOUT 1: the spectrum comes in the JUCE engine's interleaved layout (X0 = c[0].re, X_{N/2} = c[M].re,
// chowdsp_fft_juce.cpp:71-73) and the signal is stored times `scale` (the adapter's 1/N, :85) -- one launch instead of a
// repacking pass over the frames before the transform.
// (M = 512: one VGPR over the five-waves-per-SIMD budget without the hint; measured below)
template <class P, int FPB, bool STRIDED, int LAYOUT = 0>
__global__ __launch_bounds__ (P::T* FPB, (P::N == 512 && CFFT_C2R512_WAVES5) ? 5 : 1) void c2r_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ twtab,
                                                         const float2* __restrict__ wtab,
                                                         unsigned batch,
                                                         unsigned in_stride_arg,
                                                         unsigned out_stride_arg,
                                                         float scale)
{
    CFFT_FRAME_STRIDES (STRIDED, P::N);
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    // M = 512 sits one VGPR above the five-waves-per-SIMD budget (96): there the four pair twiddles are re-read per group
    // (a 2 KiB table, requested together with the group's bins) instead of living in registers
    constexpr bool STREAM_PW = (P::N == 512) && CFFT_C2R512_WAVES5;
    PairTwiddles<P> pw;
    if constexpr (! STREAM_PW)
        load_pair_twiddles<P> (pw, wtab, t);

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    const unsigned team_base = (unsigned) team * M; // inside the LDS staging area
    const unsigned team_in = (unsigned) team * in_stride, team_out = (unsigned) team * out_stride; // in global memory
    constexpr bool STAGED = (P::T <= CFFT_REAL_STAGED_MAX_T); // narrow teams: whole-workgroup coalesced copies through LDS (plans.hpp)
    float2* stage = smem + FPB * LdsFrame<M>::floats2;
    StagePrefetch<P::T * FPB, FPB * M, M> ahead; // (STAGED only) the next group's operands, requested a group ahead (stockham.hpp)
    if constexpr (STAGED && kStagedPrefetch<M, true>)
        if (blockIdx.x < ngroups)
            ahead.fetch (in + (size_t) blockIdx.x * FPB * in_stride, in_stride, min ((unsigned) FPB, batch - blockIdx.x * FPB) * M, tid);
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const bool active = g * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) g * FPB * in_stride;
        float2* __restrict__ dst = out + (size_t) g * FPB * out_stride;
        const unsigned valid = min ((unsigned) FPB, batch - g * FPB) * M;
        if constexpr (STAGED)
        {
            __syncthreads();
            if constexpr (kStagedPrefetch<M, true>)
            {
                ahead.put (stage, tid);
                const unsigned gn = g + gridDim.x;
                if (gn < ngroups)
                    ahead.fetch (in + (size_t) gn * FPB * in_stride, in_stride, min ((unsigned) FPB, batch - gn * FPB) * M, tid);
            }
            else
                stage_in_strided<P::T * FPB, FPB * M, M> (src, in_stride, stage, valid, tid);
            __syncthreads();
        }

        if constexpr (STREAM_PW)
        {
            const float2* wp = wtab;
            asm volatile ("" : "+s"(wp)); // not hoisted out of the group loop
            load_pair_twiddles<P> (pw, wp, t);
        }
        // pair stage straight from global memory (ascending k and descending M-k, both lane-contiguous)
        float2 zk[P::R / 2], zp[P::R / 2];
        // one-wavefront teams: lane 0 loads its mirrored bins one instruction late, so that every load instruction reads one
        // aligned 512-byte block (see r2c_kernel)
        constexpr bool ROT = CFFT_REAL_ROTATE_LANE0 && ! STAGED && P::T == 64; // the JUCE layout reads the same mirrored family
        float2 xp_l[ROT ? P::R / 2 : 1];
        if constexpr (ROT)
        {
            constexpr int H = P::R / 2;
#pragma unroll
            for (int q = 0; q < H; ++q)
            {
                const int qn = (q + 1) % H;
                const int kn = P::T * qn;
                const unsigned kp0 = (unsigned) (kn == 0 ? M / 2 : M - kn);
                const unsigned kp = (t == 0) ? kp0 : (unsigned) (M - (t + P::T * q));
                xp_l[q] = active ? real_mirror_load (&src[team_in + kp]) : make_float2 (0.f, 0.f);
            }
        }
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            const int kq = (k == 0 ? M / 2 : M - k);
            float2 A = STAGED ? stage[team_base + (unsigned) k]
                              : (active ? gload (&src[team_in + (unsigned) k]) : make_float2 (0.f, 0.f));
            float2 Xp;
            if constexpr (ROT)
                Xp = (t == 0) ? xp_l[(q + P::R / 2 - 1) % (P::R / 2)] : xp_l[q]; // lane 0's bin of pair q came with the instruction before
            else
                Xp = STAGED ? stage[team_base + (unsigned) kq]
                            : (active ? gload (&src[team_in + (unsigned) kq]) : make_float2 (0.f, 0.f));
            if constexpr (LAYOUT == 1)
            {
                if (k == 0)
                    A.y = active ? gload (&src[team_in + (unsigned) M]).x : 0.f; // packed slot 0 = (X0.re, X_{N/2}.re)
            }
            real_merge_pair (A, Xp, pw.w[q], zk[q], zp[q]);
            if (k == 0)
            {
                zk[q] = make_float2 (A.x + A.y, A.x - A.y);   // Z[0] = (X0 + XM) + i (X0 - XM)
                zp[q] = make_float2 (2.f * Xp.x, -2.f * Xp.y); // Z[M/2] = 2 conj(X[M/2])
            }
        }
        team_sync<P::T>(); // previous frame's LDS reads are finished
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            // stored re<->im swapped: the inverse DFT is swap(DFT(swap(.)))
            lds[lds_phys (k)] = make_float2 (zk[q].y, zk[q].x);
            lds[lds_phys (k == 0 ? M / 2 : M - k)] = make_float2 (zp[q].y, zp[q].x);
        }
        team_sync<P::T>();
        float2 v[P::R];
        lds_load_input_order<P> (v, lds, t);
        team_sync<P::T>();
        fft_core<P> (v, tw, lds, t);
        if constexpr (LAYOUT != 0)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = make_float2 (v[s].x * scale, v[s].y * scale);
        }
        if constexpr (STAGED)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                stage[team_base + (unsigned) output_index<P> (t, s)] = make_float2 (v[s].y, v[s].x);
            __syncthreads();
            stage_out_strided<P::T * FPB, FPB * M, M> (dst, out_stride, stage, valid, tid);
        }
        else if (active)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                gstore (&dst[team_out + (unsigned) output_index<P> (t, s)], make_float2 (v[s].y, v[s].x));
        }
    }
}

// Software-pipelined c2r for one-frame-per-workgroup plans (M >= 4096): the next frame's pair loads are issued
// before this frame's merge stage, core and stores (same reasoning as c2c_kernel_pf).  Packed batches only.
template <class P>
__device__ __forceinline__ void c2r_finish (const float2 (&a)[P::R / 2], const float2 (&xp)[P::R / 2], const TwiddleRegs<P>& tw,
                                            const PairTwiddles<P>& pw, float2* __restrict__ lds, float2* __restrict__ dst, int t)
{
    constexpr int M = P::N;
    float2 zk[P::R / 2], zp[P::R / 2];
#pragma unroll
    for (int q = 0; q < P::R / 2; ++q)
    {
        const int k = t + P::T * q;
        real_merge_pair (a[q], xp[q], pw.w[q], zk[q], zp[q]);
        if (k == 0)
        {
            zk[q] = make_float2 (a[q].x + a[q].y, a[q].x - a[q].y);
            zp[q] = make_float2 (2.f * xp[q].x, -2.f * xp[q].y);
        }
    }
    team_sync<P::T>(); // previous frame's LDS reads are finished
#pragma unroll
    for (int q = 0; q < P::R / 2; ++q)
    {
        const int k = t + P::T * q;
        lds[lds_phys (k)] = make_float2 (zk[q].y, zk[q].x);
        lds[lds_phys (k == 0 ? M / 2 : M - k)] = make_float2 (zp[q].y, zp[q].x);
    }
    team_sync<P::T>();
    float2 v[P::R];
    lds_load_input_order<P> (v, lds, t);
    team_sync<P::T>();
    fft_core<P> (v, tw, lds, t);
#pragma unroll
    for (int s = 0; s < P::R; ++s)
        gstore (&dst[(unsigned) output_index<P> (t, s)], make_float2 (v[s].y, v[s].x));
}

template <class P>
__global__ __launch_bounds__ (P::T) void c2r_kernel_pf (const float2* __restrict__ in,
                                                      float2* __restrict__ out,
                                                      const float2* __restrict__ twtab,
                                                      const float2* __restrict__ wtab,
                                                      unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int t = threadIdx.x;
    float2* lds = smem;
    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    PairTwiddles<P> pw;
    load_pair_twiddles<P> (pw, wtab, t);

    unsigned g = blockIdx.x;
    if (g >= batch)
        return;
    float nax[P::R / 2], nay[P::R / 2], npx[P::R / 2], npy[P::R / 2]; // prefetch registers
#define CFFT_C2R_LOAD(G, AX, AY, PX, PY)                                              \
    {                                                                                 \
        const float2* __restrict__ src_ = in + (size_t) (G) * M;                      \
        _Pragma ("unroll") for (int q = 0; q < P::R / 2; ++q)                         \
        {                                                                             \
            const int k_ = t + P::T * q;                                              \
            const float2 a_ = gload (&src_[(unsigned) k_]);                           \
            const float2 p_ = real_mirror_load (&src_[(unsigned) (k_ == 0 ? M / 2 : M - k_)]); \
            AX[q] = a_.x;                                                             \
            AY[q] = a_.y;                                                             \
            PX[q] = p_.x;                                                             \
            PY[q] = p_.y;                                                             \
        }                                                                             \
    }
    {
        float ax[P::R / 2], ay[P::R / 2], px[P::R / 2], py[P::R / 2];
        CFFT_C2R_LOAD (g, ax, ay, px, py)
        float2 a[P::R / 2], xp[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            a[q] = make_float2 (ax[q], ay[q]);
            xp[q] = make_float2 (px[q], py[q]);
        }
        const unsigned gn = g + gridDim.x;
        if (gn < batch)
            CFFT_C2R_LOAD (gn, nax, nay, npx, npy)
        c2r_finish<P> (a, xp, tw, pw, lds, out + (size_t) g * M, t);
    }
    for (g += gridDim.x; g < batch; g += gridDim.x)
    {
        float2 a[P::R / 2], xp[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            a[q] = make_float2 (nax[q], nay[q]);
            xp[q] = make_float2 (npx[q], npy[q]);
        }
        const unsigned gn = g + gridDim.x;
        if (gn < batch)
            CFFT_C2R_LOAD (gn, nax, nay, npx, npy)
        c2r_finish<P> (a, xp, tw, pw, lds, out + (size_t) g * M, t);
    }
#undef CFFT_C2R_LOAD
}

// ---- streamed-twiddle variants for the N/2 = 16384 plan (see c2c_kernel_streamed) ---------------------------------
template <class P>
__global__ __launch_bounds__ (P::T) void r2c_kernel_streamed (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ twtab,
                                                            const float2* __restrict__ wtab,
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int t0 = threadIdx.x;
    float2* lds = smem;
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            v[s] = gload (&src[(unsigned) input_index<P> (t, s)]);
        team_sync<P::T>();
        fft_core_streamed<P> (v, twtab, lds, t);
        team_sync<P::T>(); // last-stage reads done before the image is overwritten
        CFFT_PHASE_FENCE();
        lds_store_output_order<P> (v, lds, t);
        CFFT_PHASE_FENCE();
        float2 pw[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
            pw[q] = wtab[t + P::T * q];
        team_sync<P::T>();
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            const float2 A = lds[lds_phys (k)];
            const float2 Zp = lds[lds_phys (k == 0 ? M / 2 : M - k)];
            float2 Xk, Xp;
            real_split_pair (A, Zp, pw[q], Xk, Xp);
            if (k == 0)
            {
                Xk = make_float2 (A.x + A.y, A.x - A.y);
                Xp = make_float2 (Zp.x, -Zp.y);
            }
            gstore (&dst[(unsigned) k], Xk);
            real_mirror_store<M> (&dst[(unsigned) (k == 0 ? M / 2 : M - k)], Xp);
        }
    }
}

template <class P>
__global__ __launch_bounds__ (P::T) void c2r_kernel_streamed (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ twtab,
                                                            const float2* __restrict__ wtab,
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int t0 = threadIdx.x;
    float2* lds = smem;
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        float2 zk[P::R / 2], zp[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            const float2 A = gload (&src[(unsigned) k]);
            const float2 Xp = real_mirror_load (&src[(unsigned) (k == 0 ? M / 2 : M - k)]);
            real_merge_pair (A, Xp, wtab[k], zk[q], zp[q]);
            if (k == 0)
            {
                zk[q] = make_float2 (A.x + A.y, A.x - A.y);
                zp[q] = make_float2 (2.f * Xp.x, -2.f * Xp.y);
            }
        }
        team_sync<P::T>(); // previous frame's LDS reads are finished
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            lds[lds_phys (k)] = make_float2 (zk[q].y, zk[q].x);
            lds[lds_phys (k == 0 ? M / 2 : M - k)] = make_float2 (zp[q].y, zp[q].x);
        }
        team_sync<P::T>();
        CFFT_PHASE_FENCE();
        float2 v[P::R];
        lds_load_input_order<P> (v, lds, t);
        team_sync<P::T>();
        fft_core_streamed<P> (v, twtab, lds, t);
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            gstore (&dst[(unsigned) output_index<P> (t, s)], make_float2 (v[s].y, v[s].x));
    }
}
} // namespace cfft
