// Frequency-domain multiply-accumulate, plain accumulate, and the fused convolution kernel.
//
// conv_mac_kernel   <-> pffft_convolve_internal   (simd/chowdsp_fft_impl_sse.cpp:1603-1645)
// accumulate_kernel <-> fft_accumulate_internal   (simd/chowdsp_fft_impl_sse.cpp:1647-1660)
// The reference multiplies in its SIMD-lane ("unordered") layout; this backend's unordered layout is
// the canonical one, so the same arithmetic runs on interleaved (re, im) pairs.  Rounding order of the
// reference SSE path is kept: ab + ((ar*br - ai*bi) * scaling)  (no FMA contraction of the product sum
// is required for parity; the compiler may fuse, the difference is below 1 ulp of the result).
#pragma once
#include "real_kernels.hpp"

namespace cfft
{
// ab[f][k] += a[f][k] * b[f or 0][k] * scaling ; frame = `cplx_per_frame` float2 values.
// ACCUMULATE == false: ab = a * b * scaling (a may alias ab) -- the middle step of the unfused convolution sequence.
// REAL: element 0 of every frame packs two purely real bins (DC, Nyquist) and is multiplied per
// component (reference :1614-1619, :1640-1644).
template <bool REAL, bool ACCUMULATE>
__global__ __launch_bounds__ (256) void conv_mac_kernel (const float4* __restrict__ a,
                                                        const float4* __restrict__ b,
                                                        float4* __restrict__ ab,
                                                        float scaling,
                                                        unsigned pairs_per_frame, // float4 = 2 complex per element
                                                        size_t total_pairs,
                                                        int b_is_shared)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total_pairs; i += stride)
    {
        const unsigned in_frame = (unsigned) (i % pairs_per_frame);
        const float4 va = gload4 (&a[i]);
        const float4 vb = b_is_shared ? b[in_frame] : gload4 (&b[i]);
        float4 acc = ACCUMULATE ? gload4 (&ab[i]) : make_float4 (0.f, 0.f, 0.f, 0.f);
        float r0 = va.x * vb.x - va.y * vb.y;
        float i0 = va.x * vb.y + va.y * vb.x;
        const float r1 = va.z * vb.z - va.w * vb.w;
        const float i1 = va.z * vb.w + va.w * vb.z;
        if (REAL && in_frame == 0)
        {
            r0 = va.x * vb.x;
            i0 = va.y * vb.y;
        }
        acc.x += r0 * scaling;
        acc.y += i0 * scaling;
        acc.z += r1 * scaling;
        acc.w += i1 * scaling;
        gstore4 (&ab[i], acc);
    }
}

__global__ __launch_bounds__ (256) void accumulate_kernel (const float* __restrict__ a,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ ab,
                                                          size_t count)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    const size_t n4 = count / 4;
    const float4* a4 = (const float4*) a;
    const float4* b4 = (const float4*) b;
    float4* o4 = (float4*) ab;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        const float4 x = gload4 (&a4[i]), y = gload4 (&b4[i]);
        gstore4 (&o4[i], make_float4 (x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
    }
    // tail (count not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        ab[i] = a[i] + b[i];
}

// ---- streaming partitioned convolution engine: frequency-domain delay line kernels -------------
// Y[c][k] = sum_{p<P} FDL[(head - p) mod P][c][k] * H[p][c or 0][k] * scale, accumulated in partition order
// (= P successive fft_convolve_unordered calls into a zeroed ab; reference sse.cpp:1621-1644 for one of them).
__device__ __forceinline__ void fdl_mac_term (float4& acc, const float4 a, const float4 b, float scale, bool k0)
{
    float r0 = a.x * b.x - a.y * b.y;
    float i0 = a.x * b.y + a.y * b.x;
    const float r1 = a.z * b.z - a.w * b.w;
    const float i1 = a.z * b.w + a.w * b.z;
    if (k0)
    {
        r0 = a.x * b.x; // packed DC and Nyquist bins are purely real
        i0 = a.y * b.y;
    }
    acc.x += r0 * scale;
    acc.y += i0 * scale;
    acc.z += r1 * scale;
    acc.w += i1 * scale;
}

// The delay line is streamed once per block (nontemporal), the filter partitions stay in L2.  The partition loop
// is unrolled by UNROLL with all loads of a group issued before the arithmetic: P is a run-time value, and without
// this each 16-byte load waits for the previous term's multiply-adds.
template <int UNROLL>
__global__ __launch_bounds__ (256) void fdl_mac_kernel (const float4* __restrict__ fdl,
                                                       const float4* __restrict__ H,
                                                       float4* __restrict__ Y,
                                                       float scale,
                                                       unsigned pairs_per_frame,
                                                       unsigned channels,
                                                       unsigned P,
                                                       unsigned head,
                                                       int h_shared)
{
    const size_t total = (size_t) channels * pairs_per_frame;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    const size_t slot_quads = (size_t) channels * pairs_per_frame;
    const size_t h_part_quads = (size_t) (h_shared ? 1u : channels) * pairs_per_frame;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        const unsigned c = (unsigned) (i / pairs_per_frame);
        const unsigned k = (unsigned) (i % pairs_per_frame);
        const float4* __restrict__ hk = H + (h_shared ? (size_t) k : i); // + p * h_part_quads
        const bool k0 = (k == 0);
        (void) c;
        float4 acc = make_float4 (0.f, 0.f, 0.f, 0.f);
        unsigned p = 0;
        unsigned slot = head; // (head - p) mod P
        for (; p + UNROLL <= P; p += UNROLL)
        {
            float4 a[UNROLL], b[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                a[u] = gload4 (&fdl[(size_t) slot * slot_quads + i]);
                b[u] = hk[(size_t) (p + u) * h_part_quads];
                slot = (slot == 0 ? P : slot) - 1;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                fdl_mac_term (acc, a[u], b[u], scale, k0);
        }
        for (; p < P; ++p)
        {
            const float4 a = gload4 (&fdl[(size_t) slot * slot_quads + i]);
            const float4 b = hk[(size_t) p * h_part_quads];
            slot = (slot == 0 ? P : slot) - 1;
            fdl_mac_term (acc, a, b, scale, k0);
        }
        gstore4 (&Y[i], acc);
    }
}

// frame[c] = [prev[c] | in[c]] ; prev[c] = in[c]      (L = block floats, multiples of 4)
__global__ __launch_bounds__ (256) void ols_assemble_kernel (const float4* __restrict__ in, float4* __restrict__ prev,
                                                            float4* __restrict__ frame, unsigned quads_per_block, unsigned channels)
{
    const size_t total = (size_t) channels * quads_per_block;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        const size_t c = i / quads_per_block, q = i % quads_per_block;
        const float4 x = in[i];
        const float4 old = prev[i];
        frame[c * 2 * quads_per_block + q] = old;
        frame[c * 2 * quads_per_block + quads_per_block + q] = x;
        prev[i] = x;
    }
}

// out[c] = second half of y[c]
__global__ __launch_bounds__ (256) void ols_extract_kernel (const float4* __restrict__ y, float4* __restrict__ out,
                                                           unsigned quads_per_block, unsigned channels)
{
    const size_t total = (size_t) channels * quads_per_block;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        const size_t c = i / quads_per_block, q = i % quads_per_block;
        out[i] = y[c * 2 * quads_per_block + quads_per_block + q];
    }
}

// The cores of the fused kernels.  CFFT_CONV_STREAM_TW = 1 (an A/B switch of the translation unit): plans of 1024 points and
// more fetch each stage's twiddles from the L2-resident table while the previous stage's outputs travel through LDS
// (fft_core_streamed, stockham.hpp) instead of keeping them in ~60 VGPRs -- the registers that the packed-arithmetic form
// (CFFT_PACKED, butterflies.hpp) needs to stay at two waves per SIMD.  These kernels are VALU-issue bound: 1427 vector
// instructions per work-item and frame at N = 4096 real, the two resident waves of a SIMD each 50 % VALU-active
// (profiles/r03_sq_counters_conv.txt).
#ifndef CFFT_CONV_STREAM_TW
#define CFFT_CONV_STREAM_TW 0
#endif
// minimum waves per SIMD asked of the compiler for the software-pipelined fused kernels (1 = no constraint beyond the workgroup)
#ifndef CFFT_CONV_PF_MIN_WAVES
#define CFFT_CONV_PF_MIN_WAVES 1
#endif
// round 4: backward transforms of the fused kernels as the transposed network (1, default) or through a natural-order image (0)
#ifndef CFFT_CONV_TRANSPOSED
#define CFFT_CONV_TRANSPOSED 1
#endif
// fused convolution of narrow-team plans through the LDS staging area (1, round 4) or with direct global accesses (0)
#ifndef CFFT_CONV_STAGED
#define CFFT_CONV_STAGED 1
#endif
template <class P>
__device__ __forceinline__ void conv_core (float2 (&v)[P::R], const TwiddleRegs<P>& tw, const float2* __restrict__ twtab, float2* __restrict__ lds, int t)
{
#if CFFT_CONV_STREAM_TW
    if constexpr (P::N >= 1024)
    {
        fft_core_streamed<P> (v, twtab, lds, t);
        return;
    }
#endif
    fft_core<P> (v, tw, lds, t);
}

// ---- fused: y = BACKWARD( FORWARD(x) * H ) * scaling, one kernel, spectrum never leaves the CU -----
// COMPLEX plan: P is the N point plan, H has N complex bins in canonical order.
// The bins a work-item multiplies (its output_index positions) are the same for every frame: kept in VGPRs.
// STAGED (narrow teams, round 4): `dst` is the workgroup's LDS staging area -- the result leaves through whole-workgroup 16-byte
// copies like the plain transforms' (c2c_kernels.hpp), instead of 32-byte pieces per team and instruction
template <class P, bool PREDICATED, bool STAGED = false>
__device__ __forceinline__ void conv_c2c_finish (float2 (&v)[P::R],
                                                 const TwiddleRegs<P>& tw,
                                                 const float2 (&h)[P::R],
                                                 float2* lds,
                                                 float2* __restrict__ dst,
                                                 unsigned team_base,
                                                 int t,
                                                 float scaling,
                                                 bool active,
                                                 const float2* __restrict__ twtab)
{
    team_sync<P::T>();
    conv_core<P> (v, tw, twtab, lds, t);
    // multiply by the filter spectrum, scale, swap re/im for the inverse
    float2 w[P::R];
#pragma unroll
    for (int s = 0; s < P::R; ++s)
    {
        const float2 y = cmul (v[s], h[s]);
        w[s] = make_float2 (y.y * scaling, y.x * scaling);
    }
    if constexpr (P::NS > 1)
        team_sync<P::T>();
    // round 4: the backward transform as the TRANSPOSED network (stockham.hpp): it takes the spectrum where the forward
    // transform left it -- no natural-order image pass, one LDS round trip and two barriers less per frame.  By A/B per
    // plan (profiles/r04_conv_transposed_ab.txt; these kernels are bound by latency at two waves per SIMD, not by LDS
    // traffic): N = 4096 +2.7 %, 1024 +1.6 %, 512 +0.7 %, 256 +1.6 %, 32 +9 %, 16 +0.4 %; N = 2048 -2.5 / -3.3 %, 128 -2.1 %, 64 0
    // (these three keep the image pass).  CFFT_CONV_TRANSPOSED = 2: every plan, 0: none.
    if constexpr (CFFT_CONV_TRANSPOSED && ! CFFT_CONV_STREAM_TW && (CFFT_CONV_TRANSPOSED == 2 || (P::N != 2048 && P::N != 128 && P::N != 64)))
    {
        fft_core_transposed<P> (w, tw, lds, t);
        if (STAGED || ! PREDICATED || active)
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                if constexpr (STAGED)
                    dst[team_base + (unsigned) input_index<P> (t, s)] = make_float2 (w[s].y, w[s].x);
                else
                    gstore (&dst[team_base + (unsigned) input_index<P> (t, s)], make_float2 (w[s].y, w[s].x));
            }
        }
        return;
    }
    lds_store_output_order<P> (w, lds, t);
    team_sync<P::T>();
    float2 u[P::R]; // fresh arrays: reusing the caller's `v` through the reference leaves it in scratch memory
    lds_load_input_order<P> (u, lds, t);
    team_sync<P::T>();
    conv_core<P> (u, tw, twtab, lds, t);
    if (STAGED || ! PREDICATED || active)
    {
#pragma unroll
        for (int s = 0; s < P::R; ++s)
        {
            if constexpr (STAGED)
                dst[team_base + (unsigned) output_index<P> (t, s)] = make_float2 (u[s].y, u[s].x);
            else
                gstore (&dst[team_base + (unsigned) output_index<P> (t, s)], make_float2 (u[s].y, u[s].x));
        }
    }
}

template <class P, int FPB>
__global__ __launch_bounds__ (P::T* FPB) void conv_frames_c2c_kernel (const float2* __restrict__ in,
                                                                     const float2* __restrict__ H,
                                                                     float2* __restrict__ out,
                                                                     const float2* __restrict__ twtab,
                                                                     float scaling,
                                                                     unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<P::N>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    float2 h[P::R];
#pragma unroll
    for (int s = 0; s < P::R; ++s)
        h[s] = H[output_index<P> (t, s)];

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    const unsigned team_base = (unsigned) team * P::N;
    // Narrowest teams (T = 4: N = 16, 32): a team's instruction touches 32 bytes of its frame.  The workgroup's FPB frames are
    // one contiguous block: they travel between HBM and an LDS staging area by whole-workgroup 16-byte copies, as in
    // c2c_kernel (round 4, profiles/r04_conv_staged_small_ab.txt: complex N = 16 / 32 0.37 / 0.485 -> 0.62 / 0.66 of 8 TB/s, real N = 32 / 64 0.37 / 0.45 -> 0.62 / 0.61).
    constexpr bool STAGED = (P::T < 8) && CFFT_CONV_STAGED; // (8-lane teams, N = 64: measured equal / 3 % slower staged)
    float2* stage = smem + FPB * LdsFrame<P::N>::floats2;
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const bool active = g * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) g * (FPB * P::N);
        float2 v[P::R];
        if constexpr (STAGED)
        {
            const unsigned valid = min ((unsigned) FPB, batch - g * FPB) * P::N;
            __syncthreads(); // the previous group's copy-out has read the staging area
            stage_in<P::T * FPB, FPB * P::N> (src, stage, valid, tid);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = stage[team_base + (unsigned) input_index<P> (t, s)];
            conv_c2c_finish<P, true, true> (v, tw, h, lds, stage, team_base, t, scaling, active, twtab);
            __syncthreads();
            stage_out<P::T * FPB, FPB * P::N> (out + (size_t) g * (FPB * P::N), stage, valid, tid);
        }
        else
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = active ? gload (&src[team_base + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
            conv_c2c_finish<P, true> (v, tw, h, lds, out + (size_t) g * (FPB * P::N), team_base, t, scaling, active, twtab);
        }
    }
}

// software-pipelined variant: the next group's loads are in flight while this group's two FFT cores run; full
// groups store unconditionally, a ragged last group is handled once by the workgroup whose turn it would be
template <class P, int FPB>
__global__ __launch_bounds__ (P::T* FPB, CFFT_CONV_PF_MIN_WAVES) void conv_frames_c2c_kernel_pf (const float2* __restrict__ in,
                                                                        const float2* __restrict__ H,
                                                                        float2* __restrict__ out,
                                                                        const float2* __restrict__ twtab,
                                                                        float scaling,
                                                                        unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<P::N>::floats2;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    float2 h[P::R];
#pragma unroll
    for (int s = 0; s < P::R; ++s)
        h[s] = H[output_index<P> (t, s)];

    const unsigned nfull = batch / FPB;
    const unsigned team_base = (unsigned) team * P::N;
    float nre[P::R], nim[P::R];
    unsigned g = blockIdx.x;
    if (g < nfull)
    {
        {
            const float2* __restrict__ src = in + (size_t) g * (FPB * P::N);
            float2 v[P::R];
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = gload (&src[team_base + (unsigned) input_index<P> (t, s)]);
            const unsigned gn = g + gridDim.x;
            if (gn < nfull)
            {
                const float2* __restrict__ srcn = in + (size_t) gn * (FPB * P::N);
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_base + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            conv_c2c_finish<P, false> (v, tw, h, lds, out + (size_t) g * (FPB * P::N), team_base, t, scaling, true, twtab);
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
                const float2* __restrict__ srcn = in + (size_t) gn * (FPB * P::N);
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_base + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            conv_c2c_finish<P, false> (v, tw, h, lds, out + (size_t) g * (FPB * P::N), team_base, t, scaling, true, twtab);
        }
    }
    if (FPB > 1 && nfull * FPB < batch && blockIdx.x == nfull % gridDim.x)
    {
        const bool active = nfull * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) nfull * (FPB * P::N);
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            v[s] = active ? gload (&src[team_base + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
        conv_c2c_finish<P, true> (v, tw, h, lds, out + (size_t) nfull * (FPB * P::N), team_base, t, scaling, active, twtab);
    }
}

// REAL plan: P is the M = N/2 point plan, H is a packed real spectrum (M float2, element 0 = DC/Nyquist).
// The filter bins a work-item multiplies are the same for every frame, so the persistent workgroup keeps them in
// VGPRs (R float2) instead of re-reading them from L2 for each frame.
template <class P, bool REGS>
struct PairFilter
{
    float2 k[REGS ? P::R / 2 : 1], p[REGS ? P::R / 2 : 1];
    const float2* __restrict__ H; // REGS == false: read per frame from L2 (kernels at the VGPR limit)
    __device__ __forceinline__ float2 bin_k (int q, int k) const { return REGS ? this->k[REGS ? q : 0] : H[k]; }
    __device__ __forceinline__ float2 bin_p (int q, int kp) const { return REGS ? this->p[REGS ? q : 0] : H[kp]; }
};

template <class P, bool REGS>
__device__ __forceinline__ void load_pair_filter (PairFilter<P, REGS>& h, const float2* __restrict__ H, int t)
{
    constexpr int M = P::N;
    h.H = H;
    if constexpr (REGS)
    {
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            h.k[q] = H[k];
            h.p[q] = H[k == 0 ? M / 2 : M - k];
        }
    }
}

// everything after the frame is in registers: forward core, pair split * H merge, inverse core, store
template <class P, bool PREDICATED, bool HREGS, bool STAGED = false>
__device__ __forceinline__ void conv_real_finish (float2 (&v)[P::R],
                                                  const TwiddleRegs<P>& tw,
                                                  const PairTwiddles<P>& pw,
                                                  const PairFilter<P, HREGS>& h,
                                                  float2* lds,
                                                  float2* __restrict__ dst,
                                                  unsigned team_base,
                                                  int t,
                                                  float scaling,
                                                  bool active,
                                                  const float2* __restrict__ twtab)
{
    constexpr int M = P::N;
    team_sync<P::T>();
    conv_core<P> (v, tw, twtab, lds, t);
    if constexpr (P::NS > 1)
        team_sync<P::T>();
    lds_store_output_order<P> (v, lds, t);
    team_sync<P::T>();

    // split -> multiply -> merge on each (k, M-k) pair, all in registers
    float2 zk[P::R / 2], zp[P::R / 2];
#pragma unroll
    for (int q = 0; q < P::R / 2; ++q)
    {
        const int k = t + P::T * q;
        const int kp = (k == 0 ? M / 2 : M - k);
        const float2 A = lds[lds_phys (k)];
        const float2 Zp = lds[lds_phys (kp)];
        const float2 hk = h.bin_k (q, k), hp = h.bin_p (q, kp);
        float2 Xk, Xp;
        real_split_pair (A, Zp, pw.w[q], Xk, Xp);
        float2 Yk = cmul (Xk, hk), Yp = cmul (Xp, hp);
        if (k == 0)
        {
            // packed DC / Nyquist: real products (reference :1640-1644); X[M/2] = conj(Z[M/2])
            Yk = make_float2 ((A.x + A.y) * hk.x, (A.x - A.y) * hk.y);
            Yp = cmul (make_float2 (Zp.x, -Zp.y), hp);
        }
        Yk = make_float2 (Yk.x * scaling, Yk.y * scaling);
        Yp = make_float2 (Yp.x * scaling, Yp.y * scaling);
        real_merge_pair (Yk, Yp, pw.w[q], zk[q], zp[q]);
        if (k == 0)
        {
            zk[q] = make_float2 (Yk.x + Yk.y, Yk.x - Yk.y);
            zp[q] = make_float2 (2.f * Yp.x, -2.f * Yp.y);
        }
    }
    team_sync<P::T>(); // every pair has been read before the image is rewritten
#pragma unroll
    for (int q = 0; q < P::R / 2; ++q)
    {
        const int k = t + P::T * q;
        lds[lds_phys (k)] = make_float2 (zk[q].y, zk[q].x);
        lds[lds_phys (k == 0 ? M / 2 : M - k)] = make_float2 (zp[q].y, zp[q].x);
    }
    team_sync<P::T>();
    float2 u[P::R]; // a fresh array: reusing the caller's `v` through the reference leaves it in scratch memory
    lds_load_input_order<P> (u, lds, t);
    team_sync<P::T>();
    conv_core<P> (u, tw, twtab, lds, t);
    if (STAGED || ! PREDICATED || active)
    {
#pragma unroll
        for (int s = 0; s < P::R; ++s)
        {
            if constexpr (STAGED) // (conv_c2c_finish: the workgroup's LDS staging area)
                dst[team_base + (unsigned) output_index<P> (t, s)] = make_float2 (u[s].y, u[s].x);
            else
                gstore (&dst[team_base + (unsigned) output_index<P> (t, s)], make_float2 (u[s].y, u[s].x));
        }
    }
}

// ---- round 4: the fused real convolution with the pair stage IN REGISTERS ---------------------------------------------
// What the kernel above does between its two transforms -- spectrum to a natural-order image, pairs (k, M - k) read back,
// split * H merge, pairs written, image read in input order: two LDS round trips and four barriers -- exists only because
// a bin and its mirror bin sit in different work-items.  They need not: the last forward stage reads its operands from
// LDS, so WHICH butterflies a work-item computes there is free.  With NB = M / r butterflies i (outputs k = i + NB b) the
// mirror of (i, b) is (NB - i, r - 1 - b): give a work-item the butterflies i and NB - i (RealPairMap) and all its pairs
// are its own registers, slot (q0, b) with slot (q1, r - 1 - b).  Two butterflies mirror into themselves (i = 0: b <-> r - b,
// with the packed DC / Nyquist bin and bin M / 2 left over; i = NB / 2: b <-> r - 1 - b): they go to work-item 0, which
// re-arranges its registers into the regular pattern with selects (first wave only) -- its pair twiddles and filter bins
// are loaded for ITS bins, so the arithmetic needs no special case beyond the DC / Nyquist product (sse.cpp:1640-1644).
// The backward transform is the transposed network (stockham.hpp), which starts from exactly this arrangement and shares
// the forward twiddle registers.  Per frame: 4 LDS round trips and 8 barriers instead of 6 and 12.
template <class P>
struct RealPairMap
{
    template <class PP>
    static __device__ __forceinline__ constexpr int index (int t, int q)
    {
        constexpr int NB = P::N / P::radix (P::NS - 1);
        const int i = t + (q >> 1) * P::T;
        if ((q & 1) == 0)
            return i;
        return i == 0 ? NB / 2 : NB - i;
    }
};
template <class P>
static constexpr bool kRealPairInRegs = P::NS >= 2 && ((P::R / P::radix (P::NS - 1)) % 2 == 0) && P::T >= 64;

// bin index of operand A of pair (h, b) of work-item t (its partner is M - k; k == 0: the packed DC / Nyquist bin with M / 2)
template <class P>
__device__ __forceinline__ int real_pair_bin (int t, int h, int b)
{
    constexpr int r = P::radix (P::NS - 1), NB = P::N / r;
    const int i = t + h * P::T;
    if (i != 0)
        return i + NB * b;
    return b < r / 2 ? NB * b : NB / 2 + NB * (b - r / 2);
}
// The whole pair stage -- split, * H_k, * H_(M-k), * scaling, merge -- is ONE real-linear map of (Z_k, Z_(M-k)) whose
// coefficients depend only on the filter, the pair twiddle and `scaling`, i.e. are the same for every frame:
//     Z'_k     = c0.xy * A       + c0.zw * conj (X)        X = B (regular pair), X = A (the packed DC / Nyquist bin: real-linear in A alone)
//     Z'_(M-k) = c1.xy * conj (A) + c1.zw * B
// 16 multiply-adds per pair instead of about 40 (two half-butterflies, three complex products, four scalings), no filter
// loads per frame.  The persistent work-item derives its 8 coefficient sets once per launch by pushing unit vectors
// through the very functions the three-call sequence uses (real_split_pair, cmul, real_merge_pair, the real DC / Nyquist
// products of sse.cpp:1640-1644) and keeps them in 64 VGPRs (the pair twiddles and the per-frame filter loads they
// replace took 16 + the loads' staging registers).
// CFFT_CONV_TP_MAP_LDS pairs of the P::R / 2 live in a private LDS area of the work-item instead ([j][t], conflict-free
// 16-byte reads), for builds that need the registers (packed arithmetic)
#ifndef CFFT_CONV_TP_MAP_LDS
#define CFFT_CONV_TP_MAP_LDS 0
#endif
template <class P>
struct RealPairMapRegs
{
    static constexpr int in_lds = CFFT_CONV_TP_MAP_LDS, in_regs = P::R / 2 - in_lds;
    float4 c0[in_regs > 0 ? in_regs : 1], c1[in_regs > 0 ? in_regs : 1];
    float4* spill; // this work-item's column of the LDS area: entry j at spill[j * stride]
    static constexpr int stride = P::T; // (FPB == 1)
    __device__ __forceinline__ float4 get0 (int p) const { return p < in_regs ? c0[p < in_regs ? p : 0] : spill[(2 * (p - in_regs)) * stride]; }
    __device__ __forceinline__ float4 get1 (int p) const { return p < in_regs ? c1[p < in_regs ? p : 0] : spill[(2 * (p - in_regs) + 1) * stride]; }
    __device__ __forceinline__ void set (int p, float4 a, float4 b)
    {
        if (p < in_regs)
        {
            c0[p < in_regs ? p : 0] = a;
            c1[p < in_regs ? p : 0] = b;
        }
        else
        {
            spill[(2 * (p - in_regs)) * stride] = a;
            spill[(2 * (p - in_regs) + 1) * stride] = b;
        }
    }
};
template <class P>
__device__ __forceinline__ void real_pair_apply (float2 A, float2 Zp, float2 w, float2 fk, float2 fp, float scaling, bool dc, float2& zk, float2& zp)
{
    float2 Xk, Xp;
    real_split_pair (A, Zp, w, Xk, Xp);
    float2 Yk = cmul (Xk, fk), Yp = cmul (Xp, fp);
    if (dc)
    {
        // packed DC / Nyquist: real products (reference :1640-1644); X[M/2] = conj(Z[M/2])
        Yk = make_float2 ((A.x + A.y) * fk.x, (A.x - A.y) * fk.y);
        Yp = cmul (make_float2 (Zp.x, -Zp.y), fp);
    }
    Yk = make_float2 (Yk.x * scaling, Yk.y * scaling);
    Yp = make_float2 (Yp.x * scaling, Yp.y * scaling);
    real_merge_pair (Yk, Yp, w, zk, zp);
    if (dc)
    {
        zk = make_float2 (Yk.x + Yk.y, Yk.x - Yk.y);
        zp = make_float2 (2.f * Yp.x, -2.f * Yp.y);
    }
}
template <class P>
__device__ __forceinline__ void load_real_pair_map (RealPairMapRegs<P>& m, const float2* __restrict__ H, const float2* __restrict__ wtab, float scaling, int t)
{
    constexpr int M = P::N, r = P::radix (P::NS - 1);
#pragma unroll
    for (int h = 0; h < P::R / r / 2; ++h)
#pragma unroll
        for (int b = 0; b < r; ++b)
        {
            // the table holds W_N^k for k <= M / 2 (N = 2 M); beyond: W_N^k = - conj (W_N^(M - k))
            const int k = real_pair_bin<P> (t, h, b);
            const float2 wt = wtab[k <= M / 2 ? k : M - k];
            const float2 w = k <= M / 2 ? wt : make_float2 (-wt.x, wt.y);
            const float2 fk = H[k], fp = H[k == 0 ? M / 2 : M - k];
            const bool dc = k == 0;
            float2 k10, p10, k01, p01, ki0, pi0;
            real_pair_apply<P> (make_float2 (1.f, 0.f), make_float2 (0.f, 0.f), w, fk, fp, scaling, dc, k10, p10); // A = 1
            real_pair_apply<P> (make_float2 (0.f, 0.f), make_float2 (1.f, 0.f), w, fk, fp, scaling, dc, k01, p01); // B = 1
            real_pair_apply<P> (make_float2 (0.f, 1.f), make_float2 (0.f, 0.f), w, fk, fp, scaling, dc, ki0, pi0); // A = i (DC bin only)
            // regular: Z'_k = M11 A + M12 conj B with M11 = k10, M12 = conj-free: B = 1 -> conj B = 1 -> M12 = k01;
            //          Z'_p = conj (M21 A + M22 conj B): A = 1 -> conj M21 = p10; B = 1 -> conj M22 = p01
            // DC bin:  Z'_0 = a A + b conj A: A = 1 -> a + b = k10; A = i -> i (a - b) = ki0  =>  a - b = -i ki0 = (ki0.y, -ki0.x)
            const float2 amb = make_float2 (ki0.y, -ki0.x);
            const float2 c_a = dc ? make_float2 (0.5f * (k10.x + amb.x), 0.5f * (k10.y + amb.y)) : k10;
            const float2 c_b = dc ? make_float2 (0.5f * (k10.x - amb.x), 0.5f * (k10.y - amb.y)) : k01;
            m.set (h * r + b, make_float4 (c_a.x, c_a.y, c_b.x, c_b.y), make_float4 (p10.x, p10.y, p01.x, p01.y));
        }
}

template <class P, bool PREDICATED>
__device__ __forceinline__ void conv_real_finish_tp (float2 (&v)[P::R],
                                                     const TwiddleRegsL1<P>& tw, // last stage loaded with RealPairMap, stage 1 from its LDS table
                                                     const RealPairMapRegs<P>& pm,
                                                     float2* lds,
                                                     float2* __restrict__ dst,
                                                     unsigned team_base,
                                                     int t,
                                                     bool active,
                                                     bool first_wave)
{
    constexpr int r = P::radix (P::NS - 1), HP = P::R / r / 2;
    team_sync<P::T>();
    fft_core<P, RealPairMap<P>> (v, tw, lds, t);
    const bool t0 = t == 0;
    float2 u[P::R]; // operands of the backward transform (a fresh array: reusing `v` through the reference leaves it in scratch memory)
#pragma unroll
    for (int h = 0; h < HP; ++h)
    {
        // butterfly i: slot b = bin i + NB b;   butterfly NB - i: slot r - 1 - b = bin M - (i + NB b)
#define va(B) v[(2 * h) * r + (B)]
#define vb(B) v[(2 * h + 1) * r + (B)]
#define ua(B) u[(2 * h) * r + (B)]
#define ub(B) u[(2 * h + 1) * r + (B)]
        float2 A[r], Zp[r];
#pragma unroll
        for (int b = 0; b < r; ++b)
        {
            A[b] = va (b);
            Zp[b] = vb (r - 1 - b);
        }
        if (h == 0 && first_wave)
        {
            // work-item 0: butterflies 0 and NB / 2, which mirror into themselves
#pragma unroll
            for (int b = r / 2; b < r; ++b)
                A[b] = t0 ? vb (b - r / 2) : A[b];
            Zp[0] = t0 ? va (r / 2) : Zp[0];
#pragma unroll
            for (int b = 1; b < r / 2; ++b)
                Zp[b] = t0 ? va (r - b) : Zp[b];
#pragma unroll
            for (int b = r / 2; b < r; ++b)
                Zp[b] = t0 ? vb (r - 1 - (b - r / 2)) : Zp[b];
        }
        float2 zk[r], zp[r];
#pragma unroll
        for (int b = 0; b < r; ++b)
        {
            const float4 c0 = pm.get0 (h * r + b), c1 = pm.get1 (h * r + b);
            const float2 a = A[b], bb = Zp[b];
            float2 x = bb;
            if (h == 0 && b == 0 && first_wave)
                x = t0 ? a : bb; // the packed DC / Nyquist bin maps onto itself
            // Z'_k = c0.xy * a + c0.zw * conj (x)
            zk[b].x = fmaf (c0.w, x.y, fmaf (c0.z, x.x, fmaf (-c0.y, a.y, c0.x * a.x)));
            zk[b].y = fmaf (c0.w, x.x, fmaf (-c0.z, x.y, fmaf (c0.y, a.x, c0.x * a.y)));
            // Z'_(M-k) = c1.xy * conj (a) + c1.zw * b
            zp[b].x = fmaf (-c1.w, bb.y, fmaf (c1.z, bb.x, fmaf (c1.y, a.y, c1.x * a.x)));
            zp[b].y = fmaf (c1.w, bb.x, fmaf (c1.z, bb.y, fmaf (c1.y, a.x, -c1.x * a.y)));
        }
        // back into the butterflies' slots, re <-> im swapped: operands of the backward transform
#pragma unroll
        for (int b = 0; b < r; ++b)
        {
            ua (b) = make_float2 (zk[b].y, zk[b].x);
            ub (r - 1 - b) = make_float2 (zp[b].y, zp[b].x);
        }
        if (h == 0 && first_wave)
        {
            ua (r / 2) = t0 ? make_float2 (zp[0].y, zp[0].x) : ua (r / 2);
#pragma unroll
            for (int j = r / 2 + 1; j < r; ++j)
                ua (j) = t0 ? make_float2 (zp[r - j].y, zp[r - j].x) : ua (j);
#pragma unroll
            for (int j = 0; j < r / 2; ++j)
                ub (j) = t0 ? make_float2 (zk[j + r / 2].y, zk[j + r / 2].x) : ub (j);
#pragma unroll
            for (int j = r / 2; j < r; ++j)
                ub (j) = t0 ? make_float2 (zp[(r - 1 - j) + r / 2].y, zp[(r - 1 - j) + r / 2].x) : ub (j);
        }
#undef va
#undef vb
#undef ua
#undef ub
    }
    team_sync<P::T>(); // every work-item's last-stage reads of the image have completed
    fft_core_transposed<P, RealPairMap<P>> (u, tw, lds, t);
    if (! PREDICATED || active)
    {
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            gstore (&dst[team_base + (unsigned) input_index<P> (t, s)], make_float2 (u[s].y, u[s].x));
    }
}

// software-pipelined kernel around conv_real_finish_tp (same frame loop as conv_frames_real_kernel_pf)
template <class P, int FPB>
__global__ __launch_bounds__ (P::T* FPB, CFFT_CONV_PF_MIN_WAVES) void conv_frames_real_tp_kernel_pf (const float2* __restrict__ in,
                                                                                                    const float2* __restrict__ H,
                                                                                                    float2* __restrict__ out,
                                                                                                    const float2* __restrict__ twtab,
                                                                                                    const float2* __restrict__ wtab,
                                                                                                    float scaling,
                                                                                                    unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;
    float2* l1 = smem + FPB * LdsFrame<M>::floats2; // stage-1 twiddle table, shared by the workgroup's teams
    const bool first_wave = __builtin_amdgcn_readfirstlane (t) < 64; // wave-uniform: the wave that holds work-item 0 of its team

    fill_l1_table<P, P::T * FPB> (l1, twtab, tid);
    TwiddleRegsL1<P> tw;
    load_twiddles<P, RealPairMap<P>> (tw, twtab, t, l1);
    RealPairMapRegs<P> pm;
    static_assert (FPB == 1 || RealPairMapRegs<P>::in_lds == 0, "the LDS part of the pair map is laid out for one team per workgroup");
    pm.spill = reinterpret_cast<float4*> (l1 + (P::radix (1) - 1) * P::prod (1)) + t; // behind the stage-1 table (16-byte aligned: 240 float2)
    load_real_pair_map<P> (pm, H, wtab, scaling, t);

    const unsigned nfull = batch / FPB;
    const unsigned team_base = (unsigned) team * M;
    float nre[P::R], nim[P::R];
    unsigned g = blockIdx.x;
    if (g < nfull)
    {
        {
            const float2* __restrict__ src = in + (size_t) g * (FPB * M);
            float2 v[P::R];
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = gload (&src[team_base + (unsigned) input_index<P> (t, s)]);
            const unsigned gn = g + gridDim.x;
            if (gn < nfull)
            {
                const float2* __restrict__ srcn = in + (size_t) gn * (FPB * M);
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_base + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            conv_real_finish_tp<P, false> (v, tw, pm, lds, out + (size_t) g * (FPB * M), team_base, t, true, first_wave);
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
                const float2* __restrict__ srcn = in + (size_t) gn * (FPB * M);
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_base + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            conv_real_finish_tp<P, false> (v, tw, pm, lds, out + (size_t) g * (FPB * M), team_base, t, true, first_wave);
        }
    }
    if (FPB > 1 && nfull * FPB < batch && blockIdx.x == nfull % gridDim.x)
    {
        const bool active = nfull * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) nfull * (FPB * M);
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            v[s] = active ? gload (&src[team_base + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
        conv_real_finish_tp<P, true> (v, tw, pm, lds, out + (size_t) nfull * (FPB * M), team_base, t, active, first_wave);
    }
}

template <class P, int FPB, bool HREGS>
__global__ __launch_bounds__ (P::T* FPB) void conv_frames_real_kernel (const float2* __restrict__ in,
                                                                      const float2* __restrict__ H,
                                                                      float2* __restrict__ out,
                                                                      const float2* __restrict__ twtab,
                                                                      const float2* __restrict__ wtab,
                                                                      float scaling,
                                                                      unsigned batch)
{
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
    PairFilter<P, HREGS> h;
    load_pair_filter<P> (h, H, t);

    const unsigned ngroups = (batch + FPB - 1) / FPB;
    const unsigned team_base = (unsigned) team * M;
    constexpr bool STAGED = (P::T < 8) && CFFT_CONV_STAGED; // narrow teams: through the LDS staging area (conv_frames_c2c_kernel)
    float2* stage = smem + FPB * LdsFrame<M>::floats2;
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const bool active = g * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) g * (FPB * M);
        float2 v[P::R];
        if constexpr (STAGED)
        {
            const unsigned valid = min ((unsigned) FPB, batch - g * FPB) * M;
            __syncthreads();
            stage_in<P::T * FPB, FPB * M> (src, stage, valid, tid);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = stage[team_base + (unsigned) input_index<P> (t, s)];
            conv_real_finish<P, true, HREGS, true> (v, tw, pw, h, lds, stage, team_base, t, scaling, active, twtab);
            __syncthreads();
            stage_out<P::T * FPB, FPB * M> (out + (size_t) g * (FPB * M), stage, valid, tid);
        }
        else
        {
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = active ? gload (&src[team_base + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
            conv_real_finish<P, true, HREGS> (v, tw, pw, h, lds, out + (size_t) g * (FPB * M), team_base, t, scaling, active, twtab);
        }
    }
}

// software-pipelined variant (same scheme as r2c_kernel_pf): the next group's loads are in flight while this
// group's two FFT cores run; full groups store unconditionally, the ragged last group is handled separately
template <class P, int FPB, bool HREGS>
__global__ __launch_bounds__ (P::T* FPB, CFFT_CONV_PF_MIN_WAVES) void conv_frames_real_kernel_pf (const float2* __restrict__ in,
                                                                         const float2* __restrict__ H,
                                                                         float2* __restrict__ out,
                                                                         const float2* __restrict__ twtab,
                                                                         const float2* __restrict__ wtab,
                                                                         float scaling,
                                                                         unsigned batch)
{
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
    PairFilter<P, HREGS> h;
    load_pair_filter<P> (h, H, t);

    const unsigned nfull = batch / FPB;
    const unsigned team_base = (unsigned) team * M;
    float nre[P::R], nim[P::R];
    unsigned g = blockIdx.x;
    if (g < nfull)
    {
        {
            const float2* __restrict__ src = in + (size_t) g * (FPB * M);
            float2 v[P::R];
#pragma unroll
            for (int s = 0; s < P::R; ++s)
                v[s] = gload (&src[team_base + (unsigned) input_index<P> (t, s)]);
            const unsigned gn = g + gridDim.x;
            if (gn < nfull)
            {
                const float2* __restrict__ srcn = in + (size_t) gn * (FPB * M);
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_base + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            conv_real_finish<P, false, HREGS> (v, tw, pw, h, lds, out + (size_t) g * (FPB * M), team_base, t, scaling, true, twtab);
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
                const float2* __restrict__ srcn = in + (size_t) gn * (FPB * M);
#pragma unroll
                for (int s = 0; s < P::R; ++s)
                {
                    const float2 x = gload (&srcn[team_base + (unsigned) input_index<P> (t, s)]);
                    nre[s] = x.x;
                    nim[s] = x.y;
                }
            }
            conv_real_finish<P, false, HREGS> (v, tw, pw, h, lds, out + (size_t) g * (FPB * M), team_base, t, scaling, true, twtab);
        }
    }
    if (FPB > 1 && nfull * FPB < batch && blockIdx.x == nfull % gridDim.x)
    {
        const bool active = nfull * FPB + team < batch;
        const float2* __restrict__ src = in + (size_t) nfull * (FPB * M);
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R; ++s)
            v[s] = active ? gload (&src[team_base + (unsigned) input_index<P> (t, s)]) : make_float2 (0.f, 0.f);
        conv_real_finish<P, true, HREGS> (v, tw, pw, h, lds, out + (size_t) nfull * (FPB * M), team_base, t, scaling, active, twtab);
    }
}

// ---- streaming partitioned convolution engine, fused per-block kernels ----------------------------
// One block of the engine is two kernels instead of five (assemble, r2c, delay-line MAC, c2r, extract):
//   ols_r2c_kernel      frame = [previous block | new block] read from the two half-frame buffers directly (no
//                       assembled copy), the new block also stored for the next call, spectrum written into the
//                       delay-line slot;
//   ols_mac_c2r_kernel  Y = sum_p FDL[(head - p) mod P] * H[p] * scale accumulated in registers in partition order
//                       while streaming the delay line (the sum never goes to memory), inverse transform, and only
//                       the valid second half of the result is written.
// HBM traffic per channel and block (N = 2L floats per frame): (P + 3) N instead of (P + 8.5) N.
// P is the M = N/2 = L point plan (any LDS-resident length up to kMaxFusedN; narrow teams load directly as well).
// Engine position, kept in DEVICE memory so that one captured process() call (a hipGraph) can be replayed for
// successive blocks: head = delay-line slot that receives this block's spectrum, cur = which of the two half-frame
// buffers holds the previous block.  The last workgroup of ols_mac_c2r_kernel to finish moves both on (every workgroup
// has read them by then, so no extra launch is needed); ols_advance_kernel does the same for other callers.
struct OlsState
{
    unsigned head;
    unsigned cur;
    unsigned done; // workgroups of the block's last kernel that have finished (the last one moves head / cur on)
};

__global__ void ols_advance_kernel (OlsState* state, unsigned partitions)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        state->head = (state->head + 1) % partitions;
        state->cur ^= 1u;
    }
}

template <class P, int FPB>
__global__ __launch_bounds__ (P::T* FPB) void ols_r2c_kernel (float2* buf0,                    // [C][M/2] the two half-frame buffers:
                                                             float2* buf1,                    //   state->cur says which holds the previous block
                                                             const float2* __restrict__ cur,  // [C][M/2] new block
                                                             float2* __restrict__ fdl,        // [Pn][C][M] delay line
                                                             const float2* __restrict__ twtab,
                                                             const float2* __restrict__ wtab,
                                                             unsigned channels,
                                                             const OlsState* __restrict__ state)
{
    const unsigned st_cur = state->cur, st_head = state->head; // workgroup-uniform (scalar loads)
    const float2* __restrict__ prev = st_cur == 0 ? buf0 : buf1;
    float2* __restrict__ keep = st_cur == 0 ? buf1 : buf0;
    float2* __restrict__ slot = fdl + (size_t) st_head * channels * P::N;
    static_assert (P::R == P::radix (0), "input_index (t, s) == t + s * T is assumed below");
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N, HALF = M / 2;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;
    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, twtab, t);
    PairTwiddles<P> pw;
    load_pair_twiddles<P> (pw, wtab, t);

    const unsigned ngroups = (channels + FPB - 1) / FPB;
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const unsigned c = g * FPB + team;
        const bool active = c < channels;
        const float2* __restrict__ lo = prev + (size_t) g * (FPB * HALF);
        const float2* __restrict__ hi = cur + (size_t) g * (FPB * HALF);
        float2* __restrict__ kp = keep + (size_t) g * (FPB * HALF);
        const unsigned hbase = (unsigned) team * HALF + (unsigned) t;
        float2 v[P::R];
#pragma unroll
        for (int s = 0; s < P::R / 2; ++s)
            v[s] = active ? gload (&lo[hbase + (unsigned) (s * P::T)]) : make_float2 (0.f, 0.f);
#pragma unroll
        for (int s = P::R / 2; s < P::R; ++s)
            v[s] = active ? gload (&hi[hbase + (unsigned) (s * P::T - HALF)]) : make_float2 (0.f, 0.f);
        // keep the new block for the next call; after ALL loads were issued, so that no store sits between two loads
        // in the in-order memory queue (a store needs its load's data, i.e. a wait for everything before it)
        if (active)
        {
#pragma unroll
            for (int s = P::R / 2; s < P::R; ++s)
                gstore (&kp[hbase + (unsigned) (s * P::T - HALF)], v[s]);
        }
        r2c_finish<P, true> (v, tw, pw, lds, slot + (size_t) g * (FPB * M), (unsigned) team * M, t, active);
    }
}

template <class P, int FPB>
__global__ __launch_bounds__ (P::T* FPB) void ols_mac_c2r_kernel (const float2* __restrict__ fdl, // [Pn][C][M]
                                                                 const float2* __restrict__ H,   // [Pn][C or 1][M]
                                                                 float2* __restrict__ out,       // [C][M/2]
                                                                 const float2* __restrict__ twtab,
                                                                 const float2* __restrict__ wtab,
                                                                 float scale,
                                                                 unsigned channels,
                                                                 unsigned Pn,
                                                                 OlsState* state,
                                                                 int h_shared)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N, HALF = M / 2;
    const unsigned head = state->head;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;

    const size_t slot_elems = (size_t) channels * M;
    const size_t hpart_elems = (size_t) (h_shared ? 1u : channels) * M;
    const unsigned ngroups = (channels + FPB - 1) / FPB;
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const unsigned c = g * FPB + team;
        const bool active = c < channels;
        // delay-line multiply-accumulate, partition order (= P successive fft_convolve_unordered calls into a zeroed
        // ab; reference sse.cpp:1621-1644 for one of them)
        float2 yk[P::R / 2], yp[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
            yk[q] = yp[q] = make_float2 (0.f, 0.f);
        if (active)
        {
            // workgroup-uniform bases (SGPR pairs) + 32-bit per-work-item offsets: no 64-bit address VGPRs per load
            const float2* __restrict__ fg = fdl + (size_t) g * (FPB * M);
            const float2* __restrict__ hg = H + (h_shared ? (size_t) 0 : (size_t) g * (FPB * M));
            const unsigned fo = (unsigned) team * M, ho = h_shared ? 0u : (unsigned) team * M;
            unsigned sl = head;
#pragma unroll 1
            for (unsigned p = 0; p < Pn; ++p)
            {
                const float2* __restrict__ fs = fg + (size_t) sl * slot_elems;
                const float2* __restrict__ hs = hg + (size_t) p * hpart_elems;
                float2 ak[P::R / 2], ap[P::R / 2], bk[P::R / 2], bp[P::R / 2];
#pragma unroll
                for (int q = 0; q < P::R / 2; ++q)
                {
                    const int k = t + P::T * q;
                    const int kq = (k == 0 ? M / 2 : M - k);
                    ak[q] = gload (&fs[fo + (unsigned) k]);
                    ap[q] = gload (&fs[fo + (unsigned) kq]);
                    bk[q] = hs[ho + (unsigned) k];
                    bp[q] = hs[ho + (unsigned) kq];
                }
#pragma unroll
                for (int q = 0; q < P::R / 2; ++q)
                {
                    const int k = t + P::T * q;
                    float2 mk = cmul (ak[q], bk[q]);
                    if (k == 0)
                        mk = make_float2 (ak[q].x * bk[q].x, ak[q].y * bk[q].y); // packed DC / Nyquist: real products
                    const float2 mp = cmul (ap[q], bp[q]);
                    yk[q] = make_float2 (yk[q].x + mk.x * scale, yk[q].y + mk.y * scale);
                    yp[q] = make_float2 (yp[q].x + mp.x * scale, yp[q].y + mp.y * scale);
                }
                sl = (sl == 0 ? Pn : sl) - 1;
            }
        }
        // from here on: c2r_kernel with the pair inputs already in registers.
        // The twiddles are (re)loaded from L2 HERE, per frame, not kept across the loop as in the other kernels: during
        // the accumulation above the registers hold 2 x R accumulators and 4 x R loads in flight, and the 2 x ~40
        // twiddle registers on top would halve the occupancy.  (The pointer laundering keeps the compiler from
        // hoisting the loads back out of the loop.)  ~40 L2 loads per frame next to 32 x P streamed ones.
        const float2* twp = twtab;
        const float2* wp = wtab;
        asm volatile ("" : "+s"(twp), "+s"(wp));
        TwiddleRegs<P> tw;
        load_twiddles<P> (tw, twp, t);
        PairTwiddles<P> pw;
        load_pair_twiddles<P> (pw, wp, t);
        float2 zk[P::R / 2], zp[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = t + P::T * q;
            real_merge_pair (yk[q], yp[q], pw.w[q], zk[q], zp[q]);
            if (k == 0)
            {
                zk[q] = make_float2 (yk[q].x + yk[q].y, yk[q].x - yk[q].y);
                zp[q] = make_float2 (2.f * yp[q].x, -2.f * yp[q].y);
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
        if (active)
        {
            // overlap-save: the first half of the circular convolution is aliased, only the second half leaves the chip
            float2* __restrict__ dst = out + (size_t) c * HALF;
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const int o = output_index<P> (t, s);
                if (o >= HALF)
                    gstore (&dst[o - HALF], make_float2 (v[s].y, v[s].x));
            }
        }
    }
    // end of the block: the last workgroup to get here moves the engine position on.  Every workgroup has read
    // head (above) before it arrives, so nobody can see the new value within this launch.
    if (threadIdx.x == 0)
    {
        const unsigned arrived = atomicAdd (&state->done, 1u);
        if (arrived == gridDim.x - 1)
        {
            state->done = 0;
            state->head = (head + 1) % Pn;
            state->cur ^= 1u;
        }
    }
}

// ---- the whole block in ONE kernel ------------------------------------------------------------------------------
// Channels are independent, so the team that transforms a channel's new frame can go straight on to that channel's
// delay-line sum and inverse transform: ols_r2c_kernel + ols_mac_c2r_kernel back to back per channel, with the new
// spectrum used from registers for partition 0 (it is still written to its delay-line slot for the following blocks).
// One launch per block instead of two (what counts for a handful of channels), (P + 2) N of traffic instead of (P + 3) N.
template <class P, int FPB>
__global__ __launch_bounds__ (P::T* FPB) void ols_block_kernel (float2* buf0, // [C][M/2] the two half-frame buffers
                                                               float2* buf1,
                                                               const float2* __restrict__ cur,  // [C][M/2] new block
                                                               float2* fdl,                     // [Pn][C][M] delay line
                                                               const float2* __restrict__ H,    // [Pn][C or 1][M]
                                                               float2* __restrict__ out,        // [C][M/2]
                                                               const float2* __restrict__ twtab,
                                                               const float2* __restrict__ wtab,
                                                               float scale,
                                                               unsigned channels,
                                                               unsigned Pn,
                                                               OlsState* state,
                                                               int h_shared)
{
    static_assert (P::R == P::radix (0), "input_index (t, s) == t + s * T is assumed below");
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int M = P::N, HALF = M / 2;
    const unsigned st_cur = state->cur, head = state->head; // workgroup-uniform (scalar loads)
    const float2* __restrict__ prev = st_cur == 0 ? buf0 : buf1;
    float2* __restrict__ keep = st_cur == 0 ? buf1 : buf0;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    float2* lds = smem + team * LdsFrame<M>::floats2;
    const size_t slot_elems = (size_t) channels * M;
    const size_t hpart_elems = (size_t) (h_shared ? 1u : channels) * M;
    const unsigned ngroups = (channels + FPB - 1) / FPB;
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const unsigned c = g * FPB + team;
        const bool active = c < channels;
        // twiddles are (re)loaded per frame and per half: between the two transforms the registers hold the accumulators
        // and the loads in flight (see ols_mac_c2r_kernel); the laundering keeps the compiler from hoisting them
        const float2* twp = twtab;
        const float2* wp = wtab;
        asm volatile ("" : "+s"(twp), "+s"(wp));
        // Round 4: a work-item's pairs are ADJACENT bins -- k = 2 j + e, j = t + T q', e = 0 / 1 -- so that the delay line, the
        // filter partitions and the new spectrum move 16 bytes per instruction (bins 2 j, 2 j + 1 and their mirrors M - 2 j - 1,
        // M - 2 j): half the vector-memory instructions of the k = t + T q assignment, whose 8-byte loads were what this kernel
        // was issue-bound on (profiles/r03_sq_counters_conv.txt: 4624 reads per wave, 55 % issue stalls).  The pair stage reads
        // the forward spectrum from the LDS image, so the assignment is free.  j = 0 (work-item 0, q' = 0) is the odd one:
        // bin 0 pairs with M / 2 (packed DC / Nyquist) and bin 1 with M - 1.
        PairTwiddles<P> pw;
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
            pw.w[q] = wp[2 * (t + P::T * (q >> 1)) + (q & 1)];
        const bool j0 = t == 0; // (with q' == 0)
        float2 xk[P::R / 2], xp[P::R / 2]; // this block's spectrum, pairs (k, M - k)
        {
            TwiddleRegs<P> tw;
            load_twiddles<P> (tw, twp, t);
            const float2* __restrict__ lo = prev + (size_t) g * (FPB * HALF);
            const float2* __restrict__ hi = cur + (size_t) g * (FPB * HALF);
            float2* __restrict__ kp = keep + (size_t) g * (FPB * HALF);
            const unsigned hbase = (unsigned) team * HALF + (unsigned) t;
            float2 v[P::R];
#pragma unroll
            for (int s = 0; s < P::R / 2; ++s)
                v[s] = active ? gload (&lo[hbase + (unsigned) (s * P::T)]) : make_float2 (0.f, 0.f);
#pragma unroll
            for (int s = P::R / 2; s < P::R; ++s)
                v[s] = active ? gload (&hi[hbase + (unsigned) (s * P::T - HALF)]) : make_float2 (0.f, 0.f);
            if (active)
            {
#pragma unroll
                for (int s = P::R / 2; s < P::R; ++s)
                    gstore (&kp[hbase + (unsigned) (s * P::T - HALF)], v[s]);
            }
            team_sync<P::T>();
            fft_core<P> (v, tw, lds, t);
            if constexpr (P::NS > 1)
                team_sync<P::T>();
            lds_store_output_order<P> (v, lds, t);
            team_sync<P::T>();
        }
        float2* __restrict__ slot = fdl + (size_t) head * slot_elems + (size_t) g * (FPB * M);
        // idle teams of the last group (channels % FPB != 0) read the group's first channel's filter bins instead of
        // running past the end of a per-channel H (Pn == 1: there is no next partition behind it)
        const unsigned fo = (unsigned) team * M, ho = (h_shared || ! active) ? 0u : (unsigned) team * M;
        const float2* __restrict__ hg = H + (h_shared ? (size_t) 0 : (size_t) g * (FPB * M));
        float2 yk[P::R / 2], yp[P::R / 2];
        // the filter bins of a partition / the delay-line bins of a slot, this work-item's pairs: [q] <-> bin k, its mirror
#define CFFT_OLS_LOAD_PAIRS(BK, BP, BASE, OFF, LOAD4, LOAD4U, LOAD2)                                              \
    {                                                                                                             \
        _Pragma ("unroll") for (int qp = 0; qp < P::R / 4; ++qp)                                                  \
        {                                                                                                         \
            const unsigned j_ = (unsigned) (t + P::T * qp);                                                       \
            const float4 a_ = LOAD4 (reinterpret_cast<const float4*> (&(BASE)[(OFF) + 2u * j_]));                 \
            const unsigned mi_ = (qp == 0 && j0) ? (unsigned) (M - 2) : (unsigned) M - 2u * j_ - 1u;              \
            const float4 m_ = LOAD4U (&(BASE)[(OFF) + mi_]);                                                      \
            BK[2 * qp] = make_float2 (a_.x, a_.y);                                                                \
            BK[2 * qp + 1] = make_float2 (a_.z, a_.w);                                                            \
            BP[2 * qp] = make_float2 (m_.z, m_.w);     /* mirror of 2 j: M - 2 j (j = 0: replaced below) */       \
            BP[2 * qp + 1] = make_float2 (m_.x, m_.y); /* mirror of 2 j + 1: M - 2 j - 1 */                       \
            if (qp == 0)                                                                                          \
            {                                                                                                     \
                const float2 s_ = LOAD2 (&(BASE)[(OFF) + (unsigned) (M / 2)]); /* every work-item, one line */    \
                BP[1] = j0 ? make_float2 (m_.z, m_.w) : BP[1];                 /* bin M - 1 */                    \
                BP[0] = j0 ? s_ : BP[0];                                       /* bin M / 2 */                    \
            }                                                                                                     \
        }                                                                                                         \
    }
#define CFFT_OLS_PLAIN2(p) (*(p))
#define CFFT_OLS_PLAIN4(p) (*(p))
        {
            float2 bk0[P::R / 2], bp0[P::R / 2];
            CFFT_OLS_LOAD_PAIRS (bk0, bp0, hg, ho, CFFT_OLS_PLAIN4, load4_a8, CFFT_OLS_PLAIN2)
#pragma unroll
            for (int q = 0; q < P::R / 2; ++q)
            {
                const int k = 2 * (t + P::T * (q >> 1)) + (q & 1);
                const int kq = (k == 0 ? M / 2 : M - k);
                const float2 A = lds[lds_phys (k)];
                const float2 Zp = lds[lds_phys (kq)];
                real_split_pair (A, Zp, pw.w[q], xk[q], xp[q]);
                if (k == 0)
                {
                    xk[q] = make_float2 (A.x + A.y, A.x - A.y);
                    xp[q] = make_float2 (Zp.x, -Zp.y);
                }
                // partition 0 straight from the registers
                float2 mk = cmul (xk[q], bk0[q]);
                if (k == 0)
                    mk = make_float2 (xk[q].x * bk0[q].x, xk[q].y * bk0[q].y); // packed DC / Nyquist: real products
                const float2 mp = cmul (xp[q], bp0[q]);
                yk[q] = make_float2 (mk.x * scale, mk.y * scale);
                yp[q] = make_float2 (mp.x * scale, mp.y * scale);
            }
            if (active)
            {
                // the new spectrum into its delay-line slot, 16 bytes per store (work-item 0's first group: bin M / 2 and bin M - 1 on their own)
#pragma unroll
                for (int qp = 0; qp < P::R / 4; ++qp)
                {
                    const unsigned j = (unsigned) (t + P::T * qp);
                    gstore4 (reinterpret_cast<float4*> (&slot[fo + 2u * j]), make_float4 (xk[2 * qp].x, xk[2 * qp].y, xk[2 * qp + 1].x, xk[2 * qp + 1].y));
                    if (qp == 0 && j0)
                    {
                        gstore (&slot[fo + (unsigned) (M / 2)], xp[0]);
                        gstore (&slot[fo + (unsigned) (M - 1)], xp[1]);
                    }
                    else
                        gstore4_a8 (&slot[fo + (unsigned) M - 2u * j - 1u], make_float4 (xp[2 * qp + 1].x, xp[2 * qp + 1].y, xp[2 * qp].x, xp[2 * qp].y));
                }
            }
        }
        if (active)
        {
            const float2* __restrict__ fg = fdl + (size_t) g * (FPB * M);
            unsigned sl = (head == 0 ? Pn : head) - 1;
#pragma unroll 1
            for (unsigned p = 1; p < Pn; ++p)
            {
                const float2* __restrict__ fs = fg + (size_t) sl * slot_elems;
                const float2* __restrict__ hs = hg + (size_t) p * hpart_elems;
                float2 ak[P::R / 2], ap[P::R / 2], bk[P::R / 2], bp[P::R / 2];
                CFFT_OLS_LOAD_PAIRS (ak, ap, fs, fo, gload4, gload4_a8, gload)
                CFFT_OLS_LOAD_PAIRS (bk, bp, hs, ho, CFFT_OLS_PLAIN4, load4_a8, CFFT_OLS_PLAIN2)
#pragma unroll
                for (int q = 0; q < P::R / 2; ++q)
                {
                    const int k = 2 * (t + P::T * (q >> 1)) + (q & 1);
                    float2 mk = cmul (ak[q], bk[q]);
                    if (k == 0)
                        mk = make_float2 (ak[q].x * bk[q].x, ak[q].y * bk[q].y);
                    const float2 mp = cmul (ap[q], bp[q]);
                    yk[q] = make_float2 (yk[q].x + mk.x * scale, yk[q].y + mk.y * scale);
                    yp[q] = make_float2 (yp[q].x + mp.x * scale, yp[q].y + mp.y * scale);
                }
                sl = (sl == 0 ? Pn : sl) - 1;
            }
        }
        // inverse half: as ols_mac_c2r_kernel
        const float2* twp2 = twtab;
        asm volatile ("" : "+s"(twp2));
        TwiddleRegs<P> tw2;
        load_twiddles<P> (tw2, twp2, t);
        float2 zk[P::R / 2], zp[P::R / 2];
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = 2 * (t + P::T * (q >> 1)) + (q & 1);
            real_merge_pair (yk[q], yp[q], pw.w[q], zk[q], zp[q]);
            if (k == 0)
            {
                zk[q] = make_float2 (yk[q].x + yk[q].y, yk[q].x - yk[q].y);
                zp[q] = make_float2 (2.f * yp[q].x, -2.f * yp[q].y);
            }
        }
        team_sync<P::T>(); // every pair of the forward spectrum has been read from the image
#pragma unroll
        for (int q = 0; q < P::R / 2; ++q)
        {
            const int k = 2 * (t + P::T * (q >> 1)) + (q & 1);
            lds[lds_phys (k)] = make_float2 (zk[q].y, zk[q].x);
            lds[lds_phys (k == 0 ? M / 2 : M - k)] = make_float2 (zp[q].y, zp[q].x);
        }
        team_sync<P::T>();
        float2 u[P::R];
        lds_load_input_order<P> (u, lds, t);
        team_sync<P::T>();
        fft_core<P> (u, tw2, lds, t);
        if (active)
        {
            float2* __restrict__ dst = out + (size_t) c * HALF;
#pragma unroll
            for (int s = 0; s < P::R; ++s)
            {
                const int o = output_index<P> (t, s);
                if (o >= HALF)
                    gstore (&dst[o - HALF], make_float2 (u[s].y, u[s].x));
            }
        }
        team_sync<P::T>(); // the image is free for the next channel's forward transform
    }
#undef CFFT_OLS_LOAD_PAIRS
#undef CFFT_OLS_PLAIN2
#undef CFFT_OLS_PLAIN4
    if (threadIdx.x == 0)
    {
        const unsigned arrived = atomicAdd (&state->done, 1u);
        if (arrived == gridDim.x - 1)
        {
            state->done = 0;
            state->head = (head + 1) % Pn;
            state->cur ^= 1u;
        }
    }
}
} // namespace cfft
