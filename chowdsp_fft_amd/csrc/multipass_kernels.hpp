// Frames too long for one workgroup's LDS (N > 4096 complex points): two passes through device
// memory ("four-step"), N = n1 * n2, with the length-N inter-pass twiddle table resident in L2/HBM.
//
//   n = a + n1*b   (a < n1, b < n2)        k = n2*ka + kb   (ka < n1, kb < n2)
//   X[n2*ka + kb] = sum_a W_n1^{a ka} [ W_N^{a kb} sum_b x[a + n1 b] W_n2^{b kb} ]
//
//   pass A ("columns"): for every a, the n2-point DFT over b (stride n1), times W_N^{a kb} -> I[kb][a]
//   pass B ("rows")   : for every kb, the n1-point DFT over a of I[kb][.], written transposed to X[n2*ka + kb]
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

// phase 2 of both passes: LDS-resident FFT of each of the C frames of the tile image (FS floats2 apart)
// 1024-thread workgroups have 128 VGPRs per work-item: the radix-16 plans then stream their twiddles (stockham.hpp)
template <class P, int BLOCK>
struct TileStreamsTwiddles
{
    static constexpr bool value = (BLOCK == 1024 && P::R >= 16);
};

template <class P, int C, int BLOCK>
__device__ __forceinline__ void tile_fft_frames (float2* __restrict__ smem, const TwiddleRegs<P>& tw, const float2* __restrict__ stage_tw, int team, int t)
{
    constexpr int FS = TileFrame<P::N>::stride;
    constexpr int FPB = BLOCK / P::T;
    static_assert (C % FPB == 0 || FPB % C == 0, "tile width vs teams per workgroup");
    constexpr int ROUNDS = (C + FPB - 1) / FPB;
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r)
    {
        const int c = r * FPB + team;
        const bool have = (c < C); // only false when FPB > C (then T <= 64 and fft_core uses wave-local sync only)
        float2* lds = smem + (have ? c : 0) * FS;
        float2 v[P::R];
        if (have)
            lds_load_input_order<P> (v, lds, t);
        __syncthreads();
        if (have)
        {
            // T > 64: fft_core's workgroup barriers are reached by every team (have is uniform)
            if constexpr (TileStreamsTwiddles<P, BLOCK>::value)
            {
                int tt = t;
                CFFT_LAUNDER_INDEX (tt);
                fft_core_streamed<P> (v, stage_tw, lds, tt);
            }
            else
                fft_core<P> (v, tw, lds, t);
        }
        __syncthreads();
        if (have)
            lds_store_output_order<P> (v, lds, t);
    }
}

// Workspace layout between the passes ("tile-major"): frame f, pass-A tile tA (CA adjacent columns),
// output row kb, column c inside the tile:   W[f*N + (tA*n2 + kb)*CA + c]
// -> pass A stores one fully contiguous n2*CA block per tile, pass B gathers CB*CA-element (2 KiB) runs.
// The inter-pass twiddle table uses the same order, so its reads are contiguous as well.

// loads in flight per work-item in the tile loops of pass_a_kernel / pass_b_kernel.  A/B on MI355X (tools/gpu_r3_pow2_tile_unroll_ab.sh,
// 4 / 8 / 16, fraction of 8 TB/s): c61440 0.319 / 0.327 / 0.310, c245760 0.289 / 0.287 / 0.164, c24576 0.320 / 0.171 / 0.180,
// c655360 0.246 / 0.240 / 0.149 -- more than four gains nothing and spills in the 1024-work-item variants
#ifndef CFFT_POW2_TILE_UNROLL
#define CFFT_POW2_TILE_UNROLL 4
#endif

// ---- pass A: n2-point column FFTs of CA adjacent columns, times W_N^{a*kb} ----------------------------
template <class P, int CA, bool SWAP_IN, int BLOCK>
__global__ __launch_bounds__ (BLOCK) void pass_a_kernel (const float2* __restrict__ in,
                                                      float2* __restrict__ work,
                                                      const float2* __restrict__ stage_tw,
                                                      const float2* __restrict__ big_tw,
                                                      unsigned n1,
                                                      unsigned batch,
                                                      size_t in_stride) // float2 between consecutive input frames
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int n2 = P::N;
    constexpr int FS = TileFrame<n2>::stride;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    TwiddleRegs<P> tw;
    if constexpr (! TileStreamsTwiddles<P, BLOCK>::value)
        load_twiddles<P> (tw, stage_tw, t);

    const unsigned tiles = n1 / CA;
    const size_t frame_len = (size_t) n1 * n2;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        float2* __restrict__ dst = work + (size_t) f * frame_len + (size_t) tile * (n2 * CA);
        const float2* __restrict__ twp = big_tw + (size_t) tile * (n2 * CA);

        __syncthreads(); // previous tile fully stored
#pragma unroll CFFT_POW2_TILE_UNROLL
        for (int e = tid; e < CA * n2; e += BLOCK)
        {
            const int c = e % CA, j = e / CA;
            const float2 x = gload (&src[(size_t) j * n1 + c]); // caller's input: streamed once
            smem[c * FS + lds_phys (j)] = SWAP_IN ? make_float2 (x.y, x.x) : x;
        }
        __syncthreads();
        tile_fft_frames<P, CA, BLOCK> (smem, tw, stage_tw, team, t);
        __syncthreads();
#pragma unroll CFFT_POW2_TILE_UNROLL
        for (int e = tid; e < CA * n2; e += BLOCK)
        {
            const int c = e % CA, k = e / CA;
            dst[e] = cmul (smem[c * FS + lds_phys (k)], twp[e]); // plain store: the workspace should stay cached
        }
    }
}

// ---- pass B: n1-point row FFTs of CB adjacent rows, transposed store X[n2*ka + kb] -------------------
template <class P, int CB, bool SWAP_OUT, int BLOCK>
__global__ __launch_bounds__ (BLOCK) void pass_b_kernel (const float2* __restrict__ work,
                                                      float2* __restrict__ out,
                                                      const float2* __restrict__ stage_tw,
                                                      unsigned n2,
                                                      unsigned ca_log2, // log2 of pass A's tile width
                                                      unsigned batch,
                                                      size_t out_stride) // float2 between consecutive output frames
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int n1 = P::N;
    constexpr int FS = TileFrame<n1>::stride;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    TwiddleRegs<P> tw;
    if constexpr (! TileStreamsTwiddles<P, BLOCK>::value)
        load_twiddles<P> (tw, stage_tw, t);

    const unsigned tiles = n2 / CB;
    const size_t frame_len = (size_t) n1 * n2;
    const size_t nwork = (size_t) batch * tiles;
    const unsigned ca = 1u << ca_log2;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const float2* __restrict__ src = work + (size_t) f * frame_len;
        float2* __restrict__ dst = out + (size_t) f * out_stride + kb0;

        __syncthreads();
#pragma unroll CFFT_POW2_TILE_UNROLL
        for (int e = tid; e < CB * n1; e += BLOCK)
        {
            // c fastest (ca), then the tile's row kbl (CB), then pass A's tile index tA
            const unsigned c = (unsigned) e & (ca - 1);
            const unsigned kbl = ((unsigned) e >> ca_log2) % CB;
            const unsigned tA = ((unsigned) e >> ca_log2) / CB;
            const float2 x = src[((size_t) tA * n2 + kb0 + kbl) * ca + c];
            smem[kbl * FS + lds_phys ((int) (tA * ca + c))] = x;
        }
        __syncthreads();
        tile_fft_frames<P, CB, BLOCK> (smem, tw, stage_tw, team, t);
        __syncthreads();
#pragma unroll CFFT_POW2_TILE_UNROLL
        for (int e = tid; e < CB * n1; e += BLOCK)
        {
            const int c = e % CB, k = e / CB;
            const float2 y = smem[c * FS + lds_phys (k)];
            gstore (&dst[(size_t) k * n2 + c], SWAP_OUT ? make_float2 (y.y, y.x) : y); // result: streamed once
        }
    }
}

// =====================================================================================================
// REAL frames too long for one pass (N = 2^17 .. 2^20 samples): the four-step algorithm on the real data itself,
// N = R1 x R2, instead of a complex two-pass transform of N/2 points followed by a third sweep for the pair stage.
//
//   n = a + R1 b,  k = R2 ka + kb:   X[R2 ka + kb] = sum_a W_R1^{a ka} [ W_N^{a kb} C_a[kb] ],   C_a = DFT_R2 of column a
//
// The columns are real, so two adjacent ones are transformed as ONE complex column -- z_c[b] = x[2c + R1 b] + i x[2c+1 + R1 b]
// is exactly the float2 at index c + (R1/2) b of the frame -- and separated afterwards,
//     C_2c[kb] = (Z_c[kb] + conj Z_c[R2 - kb]) / 2,     C_2c+1[kb] = (Z_c[kb] - conj Z_c[R2 - kb]) / (2 i),
// which needs Z_c[kb] and Z_c[R2 - kb] of the SAME column: both are in the tile's LDS image.  Being spectra of real data,
// C_a[R2 - kb] = conj C_a[kb]: only the rows kb = 0 .. R2/2 are kept (the intermediate is N/2 + R1 complex values, the size of
// the real frame), and each of them yields, after its R1-point row transform, R1 bins of the packed spectrum: the bins
// k = R2 ka + kb < N/2 directly, the others as the conjugates of X[N - k].  Two sweeps over the data instead of three.
// The reference does this with FFTPACK real passes + pffft_real_finalize on 4 lanes (simd/chowdsp_fft_impl_sse.cpp:552-966).
//
// Workspace per frame: [column tile tA][row kb <= R2/2][32]: the 32 real columns of a tile, a = 32 tA + al.
// =====================================================================================================
template <class P, int CA, int BLOCK>
__global__ __launch_bounds__ (BLOCK) void pass_a_real_kernel (const float2* __restrict__ in, // the real frame as N/2 float2
                                                           float2* __restrict__ work,
                                                           const float2* __restrict__ stage_tw,
                                                           const float2* __restrict__ big_tw, // [tA][kb][c][2]: W_N^{a kb}, a = 2 (CA tA + c) + {0, 1}
                                                           unsigned n1c, // complex columns = R1 / 2
                                                           unsigned batch,
                                                           size_t in_stride,
                                                           size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int n2 = P::N, NR = n2 / 2 + 1;
    constexpr int FS = TileFrame<n2>::stride;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    TwiddleRegs<P> tw;
    if constexpr (! TileStreamsTwiddles<P, BLOCK>::value)
        load_twiddles<P> (tw, stage_tw, t);
    const unsigned tiles = n1c / CA;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        float4* __restrict__ dst = reinterpret_cast<float4*> (work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) NR * CA * 2));
        const float4* __restrict__ twp = reinterpret_cast<const float4*> (big_tw + (size_t) tile * ((size_t) NR * CA * 2));
        __syncthreads(); // previous tile fully stored
#pragma unroll 4
        for (int e = tid; e < CA * n2; e += BLOCK)
        {
            const int c = e % CA, j = e / CA;
            smem[c * FS + lds_phys (j)] = gload (&src[(size_t) j * n1c + c]);
        }
        __syncthreads();
        tile_fft_frames<P, CA, BLOCK> (smem, tw, stage_tw, team, t);
        __syncthreads();
#pragma unroll 2
        for (int e = tid; e < CA * NR; e += BLOCK)
        {
            const int c = e % CA, kb = e / CA;
            const float2 zk = smem[c * FS + lds_phys (kb)];
            const float2 zm = smem[c * FS + lds_phys ((n2 - kb) & (n2 - 1))];
            // even column: (zk + conj zm) / 2;  odd column: (zk - conj zm) / (2 i)
            const float2 ce = make_float2 (0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            const float2 co = make_float2 (0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
            const float4 w4 = twp[e];
            const float2 re = cmul (ce, make_float2 (w4.x, w4.y)), ro = cmul (co, make_float2 (w4.z, w4.w));
            dst[e] = make_float4 (re.x, re.y, ro.x, ro.y); // plain store: the workspace should stay cached
        }
    }
}

// Row pass: R1-point transforms of CB adjacent rows kb (of the R2/2 + 1 kept), stored into the packed spectrum
// P[0] = (X0, X_{N/2}), P[k] = X[k] (0 < k < N/2): bins R2 ka + kb with ka < R1/2 directly, the upper half of every inner
// row as conjugates at N - k = R2 (R1 - 1 - ka) + (R2 - kb).
template <class P, int CB, int BLOCK>
__global__ __launch_bounds__ (BLOCK) void pass_b_real_kernel (const float2* __restrict__ work,
                                                           float2* __restrict__ out,
                                                           const float2* __restrict__ stage_tw,
                                                           unsigned n2, // R2
                                                           unsigned batch,
                                                           size_t out_stride,
                                                           size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int R1 = P::N;
    constexpr int FS = TileFrame<R1>::stride;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    TwiddleRegs<P> tw;
    if constexpr (! TileStreamsTwiddles<P, BLOCK>::value)
        load_twiddles<P> (tw, stage_tw, t);
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
        float2* __restrict__ dst = out + (size_t) f * out_stride;
        __syncthreads();
#pragma unroll 2
        for (int e = tid; e < CB * R1; e += BLOCK)
        {
            // al fastest (32 contiguous values), then the tile's row, then pass A's tile
            const unsigned al = (unsigned) e & 31u;
            const unsigned kbl = ((unsigned) e >> 5) % CB;
            const unsigned tA = ((unsigned) e >> 5) / CB;
            const float2 x = (int) kbl < valid ? src[((size_t) tA * NR + kb0 + kbl) * 32 + al] : make_float2 (0.f, 0.f);
            smem[kbl * FS + lds_phys ((int) (tA * 32 + al))] = x;
        }
        __syncthreads();
        tile_fft_frames<P, CB, BLOCK> (smem, tw, stage_tw, team, t);
        __syncthreads();
#pragma unroll 2
        for (int e = tid; e < CB * R1; e += BLOCK)
        {
            const int c = e % CB, k = e / CB;
            if (c >= valid)
                continue;
            const unsigned kb = kb0 + (unsigned) c;
            const float2 y = smem[c * FS + lds_phys (k)];
            if (k < R1 / 2)
            {
                if (kb == 0 && k == 0)
                    gstore (&dst[0], make_float2 (y.x, smem[c * FS + lds_phys (R1 / 2)].x)); // (X0, X_{N/2}): both real
                else
                    gstore (&dst[(size_t) k * n2 + kb], y);
            }
            else if (kb != 0 && 2 * kb != n2)
                gstore (&dst[(size_t) (R1 - 1 - k) * n2 + (n2 - kb)], make_float2 (y.x, -y.y));
        }
    }
}

// Backward, first sweep: the rows.  Row kb (0 .. R2/2) of the spectrum is gathered from the packed input -- bins
// R2 ka + kb directly for ka < R1/2, conj P[R2 (R1 - 1 - ka) + R2 - kb] for the rest (rows 0 and R2/2 are their own
// mirrors: conj P[R2 (R1 - ka)] resp. the same formula) --, transformed backward (re <-> im swap on both sides), times
// conj W_N^{a kb}, into the workspace.
template <class P, int CB, int BLOCK>
__global__ __launch_bounds__ (BLOCK) void pass_b_real_inv_kernel (const float2* __restrict__ in, // packed spectrum
                                                               float2* __restrict__ work,
                                                               const float2* __restrict__ stage_tw,
                                                               const float2* __restrict__ big_tw, // as pass_a_real_kernel's
                                                               unsigned n2,
                                                               unsigned batch,
                                                               size_t in_stride,
                                                               size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int R1 = P::N;
    constexpr int FS = TileFrame<R1>::stride;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    TwiddleRegs<P> tw;
    if constexpr (! TileStreamsTwiddles<P, BLOCK>::value)
        load_twiddles<P> (tw, stage_tw, t);
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        const float2* __restrict__ src = in + (size_t) f * in_stride;
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len;
        __syncthreads();
#pragma unroll 1
        for (int e = tid; e < CB * R1; e += BLOCK)
        {
            const int c = e % CB, k = e / CB; // row fastest: CB contiguous bins per access
            float2 x = make_float2 (0.f, 0.f);
            if (c < valid)
            {
                const unsigned kb = kb0 + (unsigned) c;
                if (k < R1 / 2)
                {
                    x = gload (&src[(size_t) k * n2 + kb]);
                    if (kb == 0 && k == 0)
                        x = make_float2 (x.x, 0.f); // X0 (its slot's second float is X_{N/2})
                }
                else if (kb == 0 && k == R1 / 2)
                    x = make_float2 (gload (&src[0]).y, 0.f); // X_{N/2}
                else
                {
                    // X[R2 k + kb] = conj X[N - R2 k - kb];  rows 0 and R2/2 mirror onto themselves
                    const size_t m = (kb == 0) ? (size_t) (R1 - k) * n2 : (size_t) (R1 - 1 - k) * n2 + (n2 - kb);
                    const float2 p = gload (&src[m]);
                    x = make_float2 (p.x, -p.y);
                }
            }
            smem[c * FS + lds_phys (k)] = make_float2 (x.y, x.x); // swapped: backward = swap . forward . swap
        }
        __syncthreads();
        tile_fft_frames<P, CB, BLOCK> (smem, tw, stage_tw, team, t);
        __syncthreads();
#pragma unroll 1
        for (int e = tid; e < CB * R1; e += BLOCK)
        {
            const unsigned al = (unsigned) e & 31u;
            const unsigned kbl = ((unsigned) e >> 5) % CB;
            const unsigned tA = ((unsigned) e >> 5) / CB;
            if ((int) kbl < valid)
            {
                const float2 ys = smem[kbl * FS + lds_phys ((int) (tA * 32 + al))];
                const size_t idx = ((size_t) tA * NR + kb0 + kbl) * 32 + al;
                dst[idx] = cmulc (make_float2 (ys.y, ys.x), big_tw[idx]); // un-swap, times conj W_N^{a kb}
            }
        }
    }
}

// Backward, second sweep: the columns.  For the complex column c = the real columns 2c, 2c + 1:
// Z_c[kb] = C_2c[kb] + i C_2c+1[kb] for kb <= R2/2 and conj C_2c[R2 - kb] + i conj C_2c+1[R2 - kb] above; backward
// transform over kb, stored as the float2 column c of the real frame.
template <class P, int CA, int BLOCK>
__global__ __launch_bounds__ (BLOCK) void pass_a_real_inv_kernel (const float2* __restrict__ work,
                                                               float2* __restrict__ out, // the real frame as N/2 float2
                                                               const float2* __restrict__ stage_tw,
                                                               unsigned n1c,
                                                               unsigned batch,
                                                               size_t out_stride,
                                                               size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int n2 = P::N, NR = n2 / 2 + 1;
    constexpr int FS = TileFrame<n2>::stride;
    const int tid = threadIdx.x;
    const int team = tid / P::T;
    const int t = tid % P::T;
    TwiddleRegs<P> tw;
    if constexpr (! TileStreamsTwiddles<P, BLOCK>::value)
        load_twiddles<P> (tw, stage_tw, t);
    const unsigned tiles = n1c / CA;
    const size_t nwork = (size_t) batch * tiles;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles);
        const unsigned tile = (unsigned) (w % tiles);
        const float4* __restrict__ src = reinterpret_cast<const float4*> (work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) NR * CA * 2));
        float2* __restrict__ dst = out + (size_t) f * out_stride + (size_t) tile * CA;
        __syncthreads();
#pragma unroll 2
        for (int e = tid; e < CA * NR; e += BLOCK)
        {
            const int c = e % CA, kb = e / CA;
            const float4 p = src[e]; // (C_2c[kb], C_2c+1[kb])
            // Z[kb] = Ce + i Co, Z[R2 - kb] = conj Ce + i conj Co; stored swapped for the backward transform
            const float2 zk = make_float2 (p.x - p.w, p.y + p.z);
            const float2 zm = make_float2 (p.x + p.w, p.z - p.y);
            smem[c * FS + lds_phys (kb)] = make_float2 (zk.y, zk.x);
            if (kb != 0 && 2 * kb != n2)
                smem[c * FS + lds_phys (n2 - kb)] = make_float2 (zm.y, zm.x);
        }
        __syncthreads();
        tile_fft_frames<P, CA, BLOCK> (smem, tw, stage_tw, team, t);
        __syncthreads();
#pragma unroll 4
        for (int e = tid; e < CA * n2; e += BLOCK)
        {
            const int c = e % CA, j = e / CA;
            const float2 y = smem[c * FS + lds_phys (j)];
            gstore (&dst[(size_t) j * n1c + c], make_float2 (y.y, y.x));
        }
    }
}

// Pair stage of the real transform as a stand-alone streaming kernel (long frames only; any even M).
// FORWARD: Z (M complex, canonical) -> packed X, in place allowed.  BACKWARD: packed X -> Z.
template <bool FORWARD>
__global__ __launch_bounds__ (256) void real_pair_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ wtab,
                                                         unsigned M,
                                                         unsigned batch,
                                                         size_t in_stride,
                                                         size_t out_stride)
{
    // Round 4: a work-item handles TWO adjacent pairs, k = 2 j and 2 j + 1 with their (adjacent) mirrors M - 2 j - 1, M - 2 j, so that
    // every access moves 16 bytes (the one-pair form issued two 8-byte loads, waited, stored: 3.1 TB/s, a third of the long real
    // mixed-radix plans' kernel time).  work item = (frame, chunk of 256 pair-pairs); j = 0 is the odd one: bin 0 pairs with M / 2
    // (packed DC / Nyquist) and bin 1 with M - 1.
    const unsigned half = M / 2; // pairs per frame, k = 0 .. M/2-1 (M is a multiple of 16)
    const unsigned quads = half / 2;
    const unsigned chunks = (quads + 255u) / 256u;
    const size_t nwork = (size_t) batch * chunks;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / chunks);
        const unsigned j = (unsigned) (w % chunks) * 256u + threadIdx.x;
        if (j >= quads)
            continue;
        const float2* __restrict__ src = in + (size_t) f * in_stride;
        float2* __restrict__ dst = out + (size_t) f * out_stride;
        const bool j0 = j == 0;
        const unsigned mi = j0 ? M - 2u : M - 2u * j - 1u; // first bin of the mirrored 16 bytes
        const float4 a4 = gload4 (reinterpret_cast<const float4*> (&src[2u * j]));
        const float4 m4 = gload4_a8 (&src[mi]);
        const float4 w4 = *reinterpret_cast<const float4*> (&wtab[2u * j]);
        const float2 s0 = j0 ? gload (&src[half]) : make_float2 (0.f, 0.f);
        const float2 A0 = make_float2 (a4.x, a4.y), A1 = make_float2 (a4.z, a4.w);
        const float2 B1 = j0 ? make_float2 (m4.z, m4.w) : make_float2 (m4.x, m4.y); // bin M - 2 j - 1
        const float2 B0 = j0 ? s0 : make_float2 (m4.z, m4.w);                       // bin M - 2 j  (j = 0: M / 2)
        float2 r0, p0, r1, p1;
        if (FORWARD)
        {
            real_split_pair (A0, B0, make_float2 (w4.x, w4.y), r0, p0);
            real_split_pair (A1, B1, make_float2 (w4.z, w4.w), r1, p1);
            if (j0)
            {
                r0 = make_float2 (A0.x + A0.y, A0.x - A0.y);
                p0 = make_float2 (B0.x, -B0.y);
            }
        }
        else
        {
            real_merge_pair (A0, B0, make_float2 (w4.x, w4.y), r0, p0);
            real_merge_pair (A1, B1, make_float2 (w4.z, w4.w), r1, p1);
            if (j0)
            {
                r0 = make_float2 (A0.x + A0.y, A0.x - A0.y);
                p0 = make_float2 (2.f * B0.x, -2.f * B0.y);
            }
        }
        gstore4 (reinterpret_cast<float4*> (&dst[2u * j]), make_float4 (r0.x, r0.y, r1.x, r1.y));
        if (j0)
        {
            gstore (&dst[half], p0);
            gstore (&dst[M - 1u], p1);
        }
        else
            gstore4_a8 (&dst[mi], make_float4 (p1.x, p1.y, p0.x, p0.y));
    }
}


// =====================================================================================================
// Fast path for N = 65536 = 256 x 256 (BASELINE configuration 5; also the core of real N = 131072).
//
// Both passes are built like the single-pass kernel: 16 global loads per work-item straight into VGPRs, radix-16,
// ONE LDS exchange, radix-16, 16 global stores -- no staging copies, two workgroup barriers per tile, and the next
// tile's loads are issued before this tile's arithmetic (software pipeline, loop-top wait vmcnt(16)).
// A tile is 16 adjacent columns (pass A) / rows (pass B); the 256 work-items are 16 teams x 16 butterflies with the
// teams INTERLEAVED across lanes (lane = column + 16 * butterfly) so that one wave-instruction touches 4 x 128
// contiguous bytes of the strided side, and 512 contiguous bytes of the workspace side.
//
// Workspace ("block") layout between the passes: 16 x 16 blocks,
//     W[f*N + ((a >> 4) * 16 + (kb >> 4)) * 256 + (a & 15) * 16 + (kb & 15)]
// (a = column / pass-B element index, kb = pass-A output index / pass-B row).  Pass A writes it with lanes
// (kb&15 fastest, a&15 next) -- it re-maps its work-items across the LDS exchange to get there -- and pass B reads
// it with lane = (kb&15) + 16*(a&15): both sides move 2 KiB contiguous per workgroup instruction.  The inter-pass
// twiddle table is stored in the same order.
// =====================================================================================================
// Plan256 (= Plan<256, 16, 16, 16>, fixed for this path whatever CorePlan<256> is) lives in plans.hpp
static constexpr int kFS256 = TileFrame<256>::stride;

__device__ __forceinline__ void load_stage1_twiddles_256 (float2 (&w)[15], const float2* __restrict__ stage_tw, int k)
{
#pragma unroll
    for (int a = 1; a < 16; ++a)
        w[a - 1] = stage_tw[Plan256::tw_offset (1) + (a - 1) * 16 + k];
}

// radix-16 stage 0 on v (no twiddle), exchange through the tile image, radix-16 stage 1 with twiddles w.
// Stage 0 role: team c0, butterfly t0.  Stage 1 role: team c1, butterfly i1.  Result: v[b] = Y_{c1}[i1 + 16 b].
__device__ __forceinline__ void fft256_two_stage (float2 (&v)[16], const float2 (&w)[15], float2* __restrict__ smem,
                                                  int c0, int t0, int c1, int i1)
{
    fft_reg<16> (v);
    __syncthreads(); // the previous tile's stage-1 reads are finished
#pragma unroll
    for (int b = 0; b < 16; ++b)
        smem[c0 * kFS256 + 17 * t0 + b] = v[b]; // lds_phys (16*t0 + b) = 17*t0 + b
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
        v[a] = smem[c1 * kFS256 + i1 + 17 * a]; // lds_phys (i1 + 16*a) = i1 + 17*a
#pragma unroll
    for (int a = 1; a < 16; ++a)
        v[a] = cmul (v[a], w[a - 1]);
    fft_reg<16> (v);
}

template <bool SWAP_IN>
__global__ __launch_bounds__ (256) void pass_a256_kernel (const float2* __restrict__ in,
                                                         float2* __restrict__ work,
                                                         const float2* __restrict__ stage_tw,
                                                         const float2* __restrict__ big_tw,
                                                         unsigned batch,
                                                         size_t in_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr unsigned n1 = 256, N = 65536, tiles = 16;
    const int tid = threadIdx.x;
    const int c0 = tid & 15, t0 = tid >> 4; // stage 0: lanes run along the 16 columns of the tile
    const int i1 = tid & 15, c1 = tid >> 4; // stage 1: lanes run along the butterfly index
    float2 w[15];
    load_stage1_twiddles_256 (w, stage_tw, i1);

    const unsigned nwork = batch * tiles;
    unsigned wk = blockIdx.x;
    if (wk >= nwork)
        return;
    float nre[16], nim[16];
#define CFFT_A256_LOAD(W, RE, IM)                                                                  \
    {                                                                                              \
        const float2* __restrict__ src_ = in + (size_t) ((W) / tiles) * in_stride + ((W) % tiles) * 16; \
        _Pragma ("unroll") for (int s = 0; s < 16; ++s)                                            \
        {                                                                                          \
            const float2 x_ = gload (&src_[(unsigned) (t0 + 16 * s) * n1 + c0]);                   \
            RE[s] = SWAP_IN ? x_.y : x_.x;                                                         \
            IM[s] = SWAP_IN ? x_.x : x_.y;                                                         \
        }                                                                                          \
    }
#define CFFT_A256_BODY(W, V)                                                                       \
    {                                                                                              \
        const unsigned wn_ = (W) + gridDim.x;                                                      \
        if (wn_ < nwork)                                                                           \
            CFFT_A256_LOAD (wn_, nre, nim)                                                         \
        fft256_two_stage (V, w, smem, c0, t0, c1, i1);                                             \
        const size_t base_ = (size_t) (((W) % tiles) * 16) * 256 + c1 * 16 + i1;                   \
        float2* __restrict__ dst_ = work + (size_t) ((W) / tiles) * N + base_;                     \
        const float2* __restrict__ twp_ = big_tw + base_;                                          \
        /* twiddles four at a time, one group ahead of the stores: gfx9 counts loads and stores in ONE in-order counter, */ \
        /* so "load W; wait; store" per element makes every wait drain the previous store (tools/isa_vmem_scan.py); with  */ \
        /* group g + 1 requested before group g is stored a wait only covers older loads                                  */ \
        float2 bw_[2][4];                                                                          \
        CFFT_PHASE_FENCE();                                                                        \
        _Pragma ("unroll") for (int b = 0; b < 4; ++b)                                             \
            bw_[0][b] = twp_[b * 256];                                                             \
        _Pragma ("unroll") for (int g_ = 0; g_ < 4; ++g_)                                          \
        {                                                                                          \
            if (g_ < 3)                                                                            \
            {                                                                                      \
                _Pragma ("unroll") for (int b = 0; b < 4; ++b)                                     \
                    bw_[(g_ + 1) & 1][b] = twp_[(b + 4 * (g_ + 1)) * 256];                         \
            }                                                                                      \
            CFFT_PHASE_FENCE();                                                                    \
            _Pragma ("unroll") for (int b = 0; b < 4; ++b)                                         \
                dst_[(b + 4 * g_) * 256] = cmul (V[b + 4 * g_], bw_[g_ & 1][b]);                   \
            CFFT_PHASE_FENCE();                                                                    \
        }                                                                                          \
    }
    {
        float2 v[16];
        float re[16], im[16];
        CFFT_A256_LOAD (wk, re, im)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            v[s] = make_float2 (re[s], im[s]);
        CFFT_A256_BODY (wk, v)
    }
    for (wk += gridDim.x; wk < nwork; wk += gridDim.x)
    {
        float2 v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s)
            v[s] = make_float2 (nre[s], nim[s]);
        CFFT_A256_BODY (wk, v)
    }
#undef CFFT_A256_BODY
#undef CFFT_A256_LOAD
}

template <bool SWAP_OUT>
__global__ __launch_bounds__ (256) void pass_b256_kernel (const float2* __restrict__ work,
                                                         float2* __restrict__ out,
                                                         const float2* __restrict__ stage_tw,
                                                         unsigned batch,
                                                         size_t out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr unsigned n2 = 256, N = 65536, tiles = 16;
    const int tid = threadIdx.x;
    const int kbl = tid & 15, t = tid >> 4; // both stages: lanes run along the 16 rows of the tile
    float2 w[15];
    load_stage1_twiddles_256 (w, stage_tw, t);

    const unsigned nwork = batch * tiles;
    unsigned wk = blockIdx.x;
    if (wk >= nwork)
        return;
    float nre[16], nim[16];
#define CFFT_B256_LOAD(W, RE, IM)                                                                  \
    {                                                                                              \
        const float2* __restrict__ src_ = work + (size_t) ((W) / tiles) * N + ((W) % tiles) * 256 + tid; \
        _Pragma ("unroll") for (int s = 0; s < 16; ++s)                                            \
        {                                                                                          \
            const float2 x_ = src_[s * 16 * 256]; /* block (a_hi = s, kb_hi = tile), element tid */ \
            RE[s] = x_.x;                                                                          \
            IM[s] = x_.y;                                                                          \
        }                                                                                          \
    }
#define CFFT_B256_BODY(W, V)                                                                       \
    {                                                                                              \
        const unsigned wn_ = (W) + gridDim.x;                                                      \
        if (wn_ < nwork)                                                                           \
            CFFT_B256_LOAD (wn_, nre, nim)                                                         \
        CFFT_PHASE_FENCE(); /* the next tile's loads stay ahead of this tile's transform */       \
        fft256_two_stage (V, w, smem, kbl, t, kbl, t);                                             \
        float2* __restrict__ dst_ = out + (size_t) ((W) / tiles) * out_stride + ((W) % tiles) * 16 + kbl; \
        _Pragma ("unroll") for (int b = 0; b < 16; ++b)                                            \
            gstore (&dst_[(unsigned) (t + 16 * b) * n2], SWAP_OUT ? make_float2 (V[b].y, V[b].x) : V[b]); \
    }
    {
        float2 v[16];
        float re[16], im[16];
        CFFT_B256_LOAD (wk, re, im)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            v[s] = make_float2 (re[s], im[s]);
        CFFT_B256_BODY (wk, v)
    }
    for (wk += gridDim.x; wk < nwork; wk += gridDim.x)
    {
        float2 v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s)
            v[s] = make_float2 (nre[s], nim[s]);
        CFFT_B256_BODY (wk, v)
    }
#undef CFFT_B256_BODY
#undef CFFT_B256_LOAD
}
} // namespace cfft
