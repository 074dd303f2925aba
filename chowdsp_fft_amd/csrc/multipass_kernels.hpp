// Frames too long for one workgroup's LDS (N > 4096 complex points): two passes through device
// memory ("four-step"), N = n1 * n2, with the length-N inter-pass twiddle table resident in L2/HBM.
//
//   n = a + n1*b   (a < n1, b < n2)        k = n2*ka + kb   (ka < n1, kb < n2)
//   X[n2*ka + kb] = sum_a W_n1^{a ka} [ W_N^{a kb} sum_b x[a + n1 b] W_n2^{b kb} ]
//
//   pass A ("columns"): for every a, the n2-point DFT over b (stride n1), times W_N^{a kb},
//                       written back to the same positions  I[kb*n1 + a]
//   pass B ("rows")   : for every kb, the n1-point DFT over a on the contiguous row I[kb*n1 + .],
//                       written transposed to X[n2*ka + kb]
//
// The reference handles any N with the same in-cache pass loop (cfftf1_ps,
// simd/chowdsp_fft_impl_sse.cpp:372-432) and only changes the scratch strategy for large N
// (chowdsp_fft.h:131-134); on the GPU the frame no longer fits on chip, hence the split.
//
// Both passes work on tiles of C adjacent columns / rows so that every global access touches C*8
// contiguous bytes (128 B for C = 16).  A tile is staged through LDS: coalesced load -> per-frame
// LDS-resident FFT by teams -> coalesced (twiddled) store.
#pragma once
#include "real_kernels.hpp"

namespace cfft
{
// frame stride inside the tile image: padded frame + 1 so that lanes walking across frames do not
// collide on LDS banks
template <int N>
struct TileFrame
{
    static constexpr int stride = LdsFrame<N>::floats2 + 1;
};

// CONTIG_IN_C : global input is contiguous along the frame index c (pass A), else along the element index (pass B)
template <class P, int C, bool CONTIG_IN_C, bool TWIDDLE, bool SWAP_IN, bool SWAP_OUT>
__global__ __launch_bounds__ (256) void tile_fft_kernel (const float2* __restrict__ in,
                                                        float2* __restrict__ out,
                                                        const float2* __restrict__ stage_tw,
                                                        const float2* __restrict__ big_tw,
                                                        unsigned tiles_per_frame,
                                                        unsigned in_cstride, unsigned in_jstride,
                                                        unsigned out_kstride,
                                                        unsigned frame_len, // N (complex points per frame)
                                                        unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int n = P::N;
    constexpr int FS = TileFrame<n>::stride;
    constexpr int FPB = 256 / P::T;
    static_assert (C % FPB == 0 || FPB % C == 0, "tile width vs teams per workgroup");
    constexpr int ROUNDS = (C + FPB - 1) / FPB;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;

    TwiddleRegs<P> tw;
    load_twiddles<P> (tw, stage_tw, t);

    const size_t nwork = (size_t) batch * tiles_per_frame;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles_per_frame);
        const unsigned tile = (unsigned) (w % tiles_per_frame);
        const float2* __restrict__ src = in + (size_t) f * frame_len + (size_t) tile * C * in_cstride;
        float2* __restrict__ dst = out + (size_t) f * frame_len + (size_t) tile * C;
        const float2* __restrict__ twp = big_tw + (size_t) tile * C;

        // ---- phase 1: coalesced tile load into the LDS image --------------------------------
        __syncthreads(); // previous tile fully stored
#pragma unroll 4
        for (int e = tid; e < C * n; e += 256)
        {
            const int c = CONTIG_IN_C ? (e % C) : (e / n);
            const int j = CONTIG_IN_C ? (e / C) : (e % n);
            const float2 x = src[(size_t) c * in_cstride + (size_t) j * in_jstride];
            smem[c * FS + lds_phys (j)] = SWAP_IN ? make_float2 (x.y, x.x) : x;
        }
        __syncthreads();

        // ---- phase 2: LDS-resident FFT of every frame of the tile ---------------------------
#pragma unroll 1
        for (int r = 0; r < ROUNDS; ++r)
        {
            const int c = r * FPB + team;
            const bool have = (c < C); // only false when FPB > C
            float2* lds = smem + (have ? c : 0) * FS;
            float2 v[P::R];
            if (have)
                lds_load_input_order<P> (v, lds, t);
            __syncthreads();
            if (have)
            {
                // fft_core synchronises with team_sync<T>(): whole-workgroup barriers when T > 64, wave-local otherwise.
                // With T > 64 every team must reach the barriers, which holds because have is workgroup-uniform then.
                fft_core<P> (v, tw, lds, t);
            }
            __syncthreads();
            if (have)
                lds_store_output_order<P> (v, lds, t);
        }
        __syncthreads();

        // ---- phase 3: coalesced (twiddled) store, contiguous along c -------------------------
#pragma unroll 4
        for (int e = tid; e < C * n; e += 256)
        {
            const int c = e % C;
            const int k = e / C;
            float2 y = smem[c * FS + lds_phys (k)];
            const size_t o = (size_t) k * out_kstride + c;
            if (TWIDDLE)
                y = cmul (y, twp[o]);
            dst[o] = SWAP_OUT ? make_float2 (y.y, y.x) : y;
        }
    }
}

// Pair stage of the real transform as a stand-alone streaming kernel (N/2 > 4096 only).
// FORWARD: Z (M complex, canonical) -> packed X, in place allowed.  BACKWARD: packed X -> Z.
template <bool FORWARD>
__global__ __launch_bounds__ (256) void real_pair_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ wtab,
                                                         unsigned M,
                                                         unsigned batch)
{
    const unsigned half = M / 2; // pairs per frame, k = 0 .. M/2-1
    const size_t total = (size_t) batch * half;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        const unsigned f = (unsigned) (i / half);
        const unsigned k = (unsigned) (i % half);
        const unsigned kp = (k == 0 ? half : M - k);
        const float2* __restrict__ src = in + (size_t) f * M;
        float2* __restrict__ dst = out + (size_t) f * M;
        const float2 A = gload (&src[k]), Bp = gload (&src[kp]);
        const float2 w = wtab[k];
        float2 rk, rp;
        if (FORWARD)
        {
            real_split_pair (A, Bp, w, rk, rp);
            if (k == 0)
            {
                rk = make_float2 (A.x + A.y, A.x - A.y);
                rp = make_float2 (Bp.x, -Bp.y);
            }
        }
        else
        {
            real_merge_pair (A, Bp, w, rk, rp);
            if (k == 0)
            {
                rk = make_float2 (A.x + A.y, A.x - A.y);
                rp = make_float2 (2.f * Bp.x, -2.f * Bp.y);
            }
        }
        gstore (&dst[k], rk);
        gstore (&dst[kp], rp);
    }
}
} // namespace cfft
