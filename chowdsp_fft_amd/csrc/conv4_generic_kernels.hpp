// Fused convolution of long mixed-radix COMPLEX frames (general two-pass plans, N = n1 x n2 with both factors at most 639 points) in
// THREE sweeps -- round 5, the complex counterpart of real4g_rows_conv_kernel (real4_generic_kernels.hpp).
//
// The transform of such a frame is the four-step algorithm: n2-point columns, times W_N^(a kb), n1-point rows.  Run the backward
// transform the other way round (rows first, then columns) and the two row passes meet on the same rows: row kb of the forward
// transform holds X[n2 ka + kb] for every ka, which is exactly the row the backward transform starts from.  So
//     columns forward (x -> workspace)  ->  rows: forward, * H * scaling, backward, in place in the workspace  ->  columns backward (workspace -> y)
// moves the frame three times where the call sequence the reference's users write (forward transform, fft_convolve_unordered,
// backward transform: test/test.cpp:161-166) moves it five times through the two-pass plans.
// Same tile shape, workspace layout ([column tile tA][kb][16]) and inter-pass table as pass_a_generic_kernel / pass_b_generic_kernel
// (generic_kernels.hpp); same pipeline as the real kernels (next tile in registers across the stages, stage twiddles in LDS, a tile's
// twiddles and filter bins loaded before its stores); compile-time stage lists through kernels_real4_static_*.hip.
#pragma once
#include "real4_generic_kernels.hpp"

namespace cfft
{
template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void c4g_cols_fwd_kernel (const float2* __restrict__ in,
                                                              float2* __restrict__ work,
                                                              const float2* __restrict__ twtab,
                                                              const float2* __restrict__ big_tw, // [tA][kb][c]: W_N^(a kb), a = 16 tA + c
                                                              GenericPlan plan,                  // n = n2, frames = 16
                                                              unsigned n1,
                                                              unsigned batch,
                                                              size_t in_stride,
                                                              unsigned lfs,
                                                              size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CA = kReal4GTile;
    const int tid = threadIdx.x;
    const int n2 = tile_frame_len<SP> (plan);
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs;
    float2* twl = smem + (size_t) 2 * CA * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n2 - 1, tid);
    const unsigned tiles = n1 / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2;
    float2 pre[SLOTS];
    const auto request = [&] (size_t w) {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) { return &src[(unsigned) (e / CA) * n1 + (unsigned) (e % CA)]; });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) n2 * CA);
        const float2* __restrict__ twp = big_tw + (size_t) tile * ((size_t) n2 * CA);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
                bufA[(e % CA) * lfs + e / CA] = pre[u];
        }
        __syncthreads();
        float2 tw[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
            tw[u] = twp[min (tid + u * BLOCK, total - 1)];
        if (w + gridDim.x < nwork)
            request (w + gridDim.x);
        const float2* res = r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CA, tid, lfs);
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
                dst[e] = cmul (res[(e % CA) * lfs + e / CA], tw[u]); // plain store: the workspace should stay cached
        }
    }
}

template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void c4g_rows_conv_kernel (float2* __restrict__ work,
                                                               const float2* __restrict__ twtab,
                                                               const float2* __restrict__ big_tw,
                                                               const float2* __restrict__ H, // N complex bins, canonical order
                                                               float scaling,
                                                               GenericPlan plan, // n = n1, frames = 16
                                                               unsigned n2,
                                                               unsigned batch,
                                                               unsigned lfs,
                                                               size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CB = kReal4GTile;
    const int tid = threadIdx.x;
    const int n1 = tile_frame_len<SP> (plan);
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs;
    float2* twl = smem + (size_t) 2 * CB * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n1 - 1, tid);
    const unsigned tiles = n2 / CB;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * n1;
    float2 pre[SLOTS];
    // element e of a tile: column c = e & 15 fastest (16 contiguous values), then the tile's row, then the column tile
    const auto ws_index = [n2] (unsigned e, unsigned kb0) {
        const unsigned c = e & 15u, kbl = (e >> 4) % CB, tA = (e >> 4) / CB;
        return (tA * n2 + kb0 + kbl) * 16u + c;
    };
    const auto request = [&] (size_t w) {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) { return &src[ws_index ((unsigned) e, tile * CB)]; });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
            {
                const unsigned c = (unsigned) e & 15u, kbl = ((unsigned) e >> 4) % CB, tA = ((unsigned) e >> 4) / CB;
                bufA[kbl * lfs + tA * 16 + c] = pre[u];
            }
        }
        __syncthreads();
        // this tile's filter bins (row fastest: 16 contiguous bins per access) and inter-pass twiddles first, the next tile's operands behind them
        float2 hv[SLOTS], tw[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = min (tid + u * BLOCK, total - 1);
            hv[u] = H[(unsigned) (e / CB) * n2 + kb0 + (unsigned) (e % CB)];
        }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
            tw[u] = big_tw[ws_index ((unsigned) min (tid + u * BLOCK, total - 1), kb0)];
        if (w + gridDim.x < nwork)
            request (w + gridDim.x);
        float2* res = const_cast<float2*> (r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CB, tid, lfs));
        // X[n2 ka + kb] * H * scaling, stored swapped (operand of the backward row transform), in place
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= total)
                continue;
            const int c = e % CB, k = e / CB;
            const float2 z = cmul (res[c * lfs + k], make_float2 (hv[u].x * scaling, hv[u].y * scaling));
            res[c * lfs + k] = make_float2 (z.y, z.x);
        }
        __syncthreads();
        const float2* back = r4g_core<BLOCK, SP> (plan, twl, res, res == bufA ? bufB : bufA, CB, tid, lfs);
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= total)
                continue;
            const unsigned c = (unsigned) e & 15u, kbl = ((unsigned) e >> 4) % CB, tA = ((unsigned) e >> 4) / CB;
            const float2 ys = back[kbl * lfs + tA * 16 + c];
            dst[ws_index ((unsigned) e, kb0)] = cmulc (make_float2 (ys.y, ys.x), tw[u]); // un-swap, times conj W_N^(a kb)
        }
    }
}

template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void c4g_cols_inv_kernel (const float2* __restrict__ work,
                                                              float2* __restrict__ out,
                                                              const float2* __restrict__ twtab,
                                                              GenericPlan plan, // n = n2, frames = 16
                                                              unsigned n1,
                                                              unsigned batch,
                                                              size_t out_stride,
                                                              unsigned lfs,
                                                              size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CA = kReal4GTile;
    const int tid = threadIdx.x;
    const int n2 = tile_frame_len<SP> (plan);
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs;
    float2* twl = smem + (size_t) 2 * CA * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n2 - 1, tid);
    const unsigned tiles = n1 / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2;
    float2 pre[SLOTS];
    const auto request = [&] (size_t w) {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) n2 * CA);
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) { return &src[e]; });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        float2* __restrict__ dst = out + (size_t) f * out_stride + (size_t) tile * CA;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
                bufA[(e % CA) * lfs + e / CA] = make_float2 (pre[u].y, pre[u].x); // swapped: backward = swap . forward . swap
        }
        __syncthreads();
        if (w + gridDim.x < nwork)
            request (w + gridDim.x);
        const float2* res = r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CA, tid, lfs);
        for (int e = tid; e < total; e += BLOCK)
        {
            const int c = e % CA, j = e / CA;
            const float2 y = res[c * lfs + j];
            gstore (&dst[(unsigned) j * n1 + (unsigned) c], make_float2 (y.y, y.x));
        }
    }
}
} // namespace cfft
