// Long MIXED-RADIX real frames (N = 2^a 3^b 5^c real samples, complex length beyond the one-pass kernels): the four-step algorithm on
// the real data itself, N = R1 x R2, with the run-time stage lists of generic_kernels.hpp for both cores -- round 5.
//
// Until round 4 these frames ran as a complex two-pass transform of N / 2 points followed by the stand-alone pair stage: three sweeps
// over the data (0.21-0.24 of 8 TB/s).  The power-of-two lengths have had the two-sweep form since round 2 (multipass_kernels.hpp,
// pass_a_real_kernel ... pass_a_real_inv_kernel); these four kernels are that construction with generic cores:
//
//   n = a + R1 b,  k = R2 ka + kb:   X[R2 ka + kb] = sum_a W_R1^{a ka} [ W_N^{a kb} C_a[kb] ],   C_a = DFT_R2 of (real) column a
//
//   pass A   two adjacent real columns as ONE complex column (the float2 at c + (R1 / 2) b of the frame), R2-point transform, separated
//            through Z[kb] and Z[R2 - kb] of the same column (both in the tile's image), times W_N^{a kb}; rows kb <= R2 / 2 only
//            (the others are conjugates): workspace [column tile tA][kb][32 real columns], N / 2 + R1 complex values per frame
//   pass B   R1-point transforms of 16 adjacent rows kb, stored into the packed spectrum: bins k = R2 ka + kb < N / 2 directly, the rest
//            as conj X[N - k]
// and the backward transform the other way round (gather rows from the packed spectrum, inverse rows, conj W, inverse columns).
// Layouts, index algebra and the (X0, X_{N/2}) slot are exactly those of the power-of-two kernels; R1 is a multiple of 32 and R2 even
// (tables.cpp split_real4_generic).  The reference does this with FFTPACK real passes of radix 2-5 + pffft_real_finalize on 4 lanes
// (simd/chowdsp_fft_impl_sse.cpp:552-786 radf2..radf5, :1081-1356 radb2..radb5, :849-1078).
#pragma once
#include "generic_kernels.hpp"

namespace cfft
{
constexpr int kReal4GTile = 16; // complex columns (pass A) / rows (pass B) per tile
// Operands of a tile per work-item: 16 frames of at most 639 points on 1024 work-items (or of at most 159 on 256: generic_tile_block).
// The whole NEXT tile is requested into these registers as soon as this tile's operands have been committed to the image; they fly
// during the stages and the stores (the tile kernels of round 3 were parked on memory for 65-80 % of their wave cycles without that).
// vmcnt is one in-order counter, so nothing between the request and the end of the stages may wait on a younger load: the stage
// twiddles are read from an LDS copy behind the images, and a tile's inter-pass twiddles are all loaded BEFORE its stores (a load
// issued behind stores waits for them too).  Slots beyond the tile re-read its last element (a test around a load makes the compiler
// wait for every load before the next).
constexpr int kReal4GSlots = 10; // (the most: SLOTS below is the launcher's pick of 4, 6, 8 or 10 -- the fewest that hold a tile of this plan)

// the stages of one tile: the run-time list, or (SP) the compile-time list of this length (radix switch, stage loop and divisions fold away)
template <int BLOCK, class SP>
__device__ __forceinline__ const float2* r4g_core (const GenericPlan& plan, const float2* twl, float2* a, float2* b, int frames, int tid, unsigned lfs)
{
    if constexpr (std::is_void<SP>::value)
        return generic_core<BLOCK, false, false, false> (plan, twl, a, b, nullptr, nullptr, frames, tid, lfs, plan.tw_offset[0]);
    else
        return static_core<SP, BLOCK, false, false, false, 0, false, SP::n + 1> (twl, a, b, nullptr, nullptr, frames, tid, 0u, 0u);
}

template <int BLOCK, int SLOTS, class T, class ADDR>
__device__ __forceinline__ void r4g_request (T (&pre)[SLOTS], int total, int tid, ADDR addr)
{
#pragma unroll
    for (int u = 0; u < SLOTS; ++u)
        pre[u] = *addr (min (tid + u * BLOCK, total - 1));
}

// Order of the ROW tiles (speed only: a bijection of [0, nwork)).  A row tile's mirrored half (bins N - k) is a 16-bin run that starts one
// bin off a 128-byte line, so the tiles kb0 and kb0 + 16 of a frame share lines on the packed side: measured, the backward row pass fetched
// 1.68 x and the forward one wrote 1.17 x their bytes over the fabric with the plain order (consecutive items on eight different XCDs).
// Workgroups are dealt round-robin over the XCDs; item v is mapped so that ONE XCD works through consecutive tiles, whose shared lines then
// meet in that XCD's L2 (CFFT_REAL4G_XCD_ORDER = 0: plain order, A/B).
#ifndef CFFT_REAL4G_XCD_ORDER
#define CFFT_REAL4G_XCD_ORDER 1
#endif
__device__ __forceinline__ size_t r4g_row_order (size_t v, size_t nwork)
{
    if (! CFFT_REAL4G_XCD_ORDER)
        return v;
    const size_t q = nwork / 8, r = nwork % 8, x = v % 8;
    return x * q + (x < r ? x : r) + v / 8; // XCD x owns q (+ 1 while x < r) consecutive tiles
}

template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void real4g_pass_a_kernel (const float2* __restrict__ in, // the real frame as N/2 float2
                                                               float2* __restrict__ work,
                                                               const float2* __restrict__ twtab,
                                                               const float2* __restrict__ big_tw, // [tA][kb][c][2]: W_N^{a kb}, a = 2 (16 tA + c) + {0, 1}
                                                               GenericPlan plan,                  // n = R2, frames = 16
                                                               unsigned n1c,                      // complex columns = R1 / 2
                                                               unsigned batch,
                                                               size_t in_stride,
                                                               unsigned lfs,
                                                               size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CA = kReal4GTile;
    const int tid = threadIdx.x;
    const int n2 = tile_frame_len<SP> (plan), NR = n2 / 2 + 1; // (SP: compile-time stage list of this length, void = the run-time list)
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs;
    float2* twl = smem + (size_t) 2 * CA * lfs; // the stage twiddles' LDS copy (n2 - 1 entries)
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n2 - 1, tid);
    const unsigned tiles = n1c / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2, half = CA * NR;
    float2 pre[SLOTS];
    const auto request = [&] (size_t w) {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) { return &src[(unsigned) (e / CA) * n1c + (unsigned) (e % CA)]; });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        float4* __restrict__ dst = reinterpret_cast<float4*> (work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) NR * CA * 2));
        const float4* __restrict__ twp = reinterpret_cast<const float4*> (big_tw + (size_t) tile * ((size_t) NR * CA * 2));
        __syncthreads(); // previous tile fully read out of the images
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
                bufA[(e % CA) * lfs + e / CA] = pre[u];
        }
        __syncthreads();
        if (w + gridDim.x < nwork)
            request (w + gridDim.x);
        const float2* res = r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CA, tid, lfs);
        // the tile's inter-pass twiddles first (half = 16 (R2 / 2 + 1) float4, at most SLOTS / 2 per work-item), then the stores
        float4 w4[SLOTS / 2];
#pragma unroll
        for (int u = 0; u < SLOTS / 2; ++u)
            w4[u] = twp[min (tid + u * BLOCK, half - 1)];
#pragma unroll
        for (int u = 0; u < SLOTS / 2; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= half)
                continue;
            const int c = e % CA, kb = e / CA;
            const float2 zk = res[c * lfs + kb];
            const float2 zm = res[c * lfs + (kb == 0 ? 0 : n2 - kb)];
            // even column: (zk + conj zm) / 2;  odd column: (zk - conj zm) / (2 i)
            const float2 ce = make_float2 (0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            const float2 co = make_float2 (0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
            const float2 re = cmul (ce, make_float2 (w4[u].x, w4[u].y)), ro = cmul (co, make_float2 (w4[u].z, w4[u].w));
            dst[e] = make_float4 (re.x, re.y, ro.x, ro.y); // plain store: the workspace should stay cached
        }
    }
}

template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void real4g_pass_b_kernel (const float2* __restrict__ work,
                                                               float2* __restrict__ out, // packed spectrum
                                                               const float2* __restrict__ twtab,
                                                               GenericPlan plan, // n = R1, frames = 16
                                                               unsigned n2,      // R2
                                                               unsigned batch,
                                                               size_t out_stride,
                                                               unsigned lfs,
                                                               size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CB = kReal4GTile;
    const int tid = threadIdx.x;
    const int R1 = tile_frame_len<SP> (plan);
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs;
    float2* twl = smem + (size_t) 2 * CB * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], R1 - 1, tid);
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * R1;
    float2 pre[SLOTS];
    const auto request = [&] (size_t v) {
        const size_t w = r4g_row_order (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const unsigned last = min ((unsigned) CB, NR - kb0) - 1u;
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
        // al fastest (32 contiguous values), then the tile's row, then pass A's tile
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) {
            const unsigned al = (unsigned) e & 31u, kbl = min (((unsigned) e >> 5) % CB, last), tA = ((unsigned) e >> 5) / CB;
            return &src[(tA * NR + kb0 + kbl) * 32u + al];
        });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t v = blockIdx.x; v < nwork; v += gridDim.x)
    {
        const size_t w = r4g_row_order (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        float2* __restrict__ dst = out + (size_t) f * out_stride;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
            {
                const unsigned al = (unsigned) e & 31u, kbl = ((unsigned) e >> 5) % CB, tA = ((unsigned) e >> 5) / CB;
                bufA[kbl * lfs + tA * 32 + al] = (int) kbl < valid ? pre[u] : make_float2 (0.f, 0.f);
            }
        }
        __syncthreads();
        if (v + gridDim.x < nwork)
            request (v + gridDim.x);
        const float2* res = r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CB, tid, lfs);
        for (int e = tid; e < total; e += BLOCK)
        {
            const int c = e % CB, k = e / CB;
            if (c >= valid)
                continue;
            const unsigned kb = kb0 + (unsigned) c;
            const float2 y = res[c * lfs + k];
            if (k < R1 / 2)
            {
                if (kb == 0 && k == 0)
                    gstore (&dst[0], make_float2 (y.x, res[c * lfs + R1 / 2].x)); // (X0, X_{N/2}): both real
                else
                    gstore (&dst[(unsigned) k * n2 + kb], y);
            }
            else if (kb != 0 && 2 * kb != n2)
                dst[(unsigned) (R1 - 1 - k) * n2 + (n2 - kb)] = make_float2 (y.x, -y.y); // (default cache policy: the run shares its first / last line with the neighbouring tile's -- they meet in L2)
        }
    }
}

// Backward, first sweep: row kb (0 .. R2/2) of the spectrum gathered from the packed input, transformed backward (re <-> im swap on
// both sides), times conj W_N^{a kb}, into the workspace
template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void real4g_pass_b_inv_kernel (const float2* __restrict__ in, // packed spectrum
                                                                   float2* __restrict__ work,
                                                                   const float2* __restrict__ twtab,
                                                                   const float2* __restrict__ big_tw,
                                                                   GenericPlan plan, // n = R1, frames = 16
                                                                   unsigned n2,
                                                                   unsigned batch,
                                                                   size_t in_stride,
                                                                   unsigned lfs,
                                                                   size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CB = kReal4GTile;
    const int tid = threadIdx.x;
    const int R1 = tile_frame_len<SP> (plan);
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs;
    float2* twl = smem + (size_t) 2 * CB * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], R1 - 1, tid);
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * R1;
    float2 pre[SLOTS];
    const auto request = [&] (size_t v) {
        const size_t w = r4g_row_order (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int last = (int) min ((unsigned) CB, NR - kb0) - 1;
        const float2* __restrict__ src = in + (size_t) f * in_stride;
        // bins R2 k + kb directly for k < R1/2; X_{N/2} from slot 0; the rest as conjugates of X[N - R2 k - kb] (rows 0 and R2/2 mirror
        // onto themselves).  Row fastest: 16 contiguous bins per access
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) {
            const int c = min (e % CB, last), k = e / CB;
            const unsigned kb = kb0 + (unsigned) c;
            unsigned idx; // (frame-local offsets stay far below 2^32: one address register per request)
            if (k < R1 / 2)
                idx = (unsigned) k * n2 + kb;
            else if (kb == 0 && k == R1 / 2)
                idx = 0;
            else
                idx = (kb == 0) ? (unsigned) (R1 - k) * n2 : (unsigned) (R1 - 1 - k) * n2 + (n2 - kb);
            return &src[idx];
        });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t v = blockIdx.x; v < nwork; v += gridDim.x)
    {
        const size_t w = r4g_row_order (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= total)
                continue;
            const int c = e % CB, k = e / CB;
            const unsigned kb = kb0 + (unsigned) c;
            float2 v = pre[u];
            if (k < R1 / 2)
            {
                if (kb == 0 && k == 0)
                    v = make_float2 (v.x, 0.f); // X0 (its slot's second float is X_{N/2})
            }
            else if (kb == 0 && k == R1 / 2)
                v = make_float2 (v.y, 0.f); // X_{N/2}
            else
                v = make_float2 (v.x, -v.y);
            if (c >= valid)
                v = make_float2 (0.f, 0.f);
            bufA[c * lfs + k] = make_float2 (v.y, v.x); // swapped: backward = swap . forward . swap
        }
        __syncthreads();
        if (v + gridDim.x < nwork)
            request (v + gridDim.x);
        const float2* res = r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CB, tid, lfs);
        // the inter-pass twiddles of the whole tile first, then the stores
        float2 tw[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const unsigned e = (unsigned) min (tid + u * BLOCK, total - 1);
            const unsigned al = e & 31u, kbl = min ((e >> 5) % CB, (unsigned) (valid - 1)), tA = (e >> 5) / CB;
            tw[u] = big_tw[(tA * NR + kb0 + kbl) * 32u + al];
        }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= total)
                continue;
            const unsigned al = (unsigned) e & 31u, kbl = ((unsigned) e >> 5) % CB, tA = ((unsigned) e >> 5) / CB;
            if ((int) kbl < valid)
            {
                const float2 ys = res[kbl * lfs + tA * 32 + al];
                dst[(tA * NR + kb0 + kbl) * 32u + al] = cmulc (make_float2 (ys.y, ys.x), tw[u]); // un-swap, times conj W_N^{a kb}
            }
        }
    }
}

// Fused convolution (fft_convolve_frames_batch, round 5): the forward row pass, the product with the filter spectrum and the backward
// row pass in ONE kernel, in place in the workspace.  Row kb of the forward transform yields X[R2 k + kb] for EVERY k -- exactly the
// row the backward transform's first sweep gathers (its upper half from the conjugates it would read back) -- so between the two
// column passes the spectrum never has to exist in memory: pass A -> this kernel -> pass A backward = three sweeps over the data
// instead of five (forward two, product one, backward two).  H is the packed spectrum of a REAL filter (what the reference's call
// sequence multiplies by, test/test.cpp:212-218): bins k < N/2 directly, the others as conj H[N - k]; bin 0 and bin N/2 are real
// products with the two halves of the packed slot 0 (simd/chowdsp_fft_impl_sse.cpp:1640-1644).
template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void real4g_rows_conv_kernel (float2* __restrict__ work,
                                                                  const float2* __restrict__ twtab,
                                                                  const float2* __restrict__ big_tw,
                                                                  const float2* __restrict__ H, // packed spectrum, N / 2 float2
                                                                  float scaling,
                                                                  GenericPlan plan, // n = R1, frames = 16
                                                                  unsigned n2,
                                                                  unsigned batch,
                                                                  unsigned lfs,
                                                                  size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CB = kReal4GTile;
    const int tid = threadIdx.x;
    const int R1 = tile_frame_len<SP> (plan);
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs;
    float2* twl = smem + (size_t) 2 * CB * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], R1 - 1, tid);
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * R1;
    float2 pre[SLOTS];
    const auto request = [&] (size_t v) {
        const size_t w = r4g_row_order (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const unsigned last = min ((unsigned) CB, NR - kb0) - 1u;
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
        r4g_request<BLOCK> (pre, total, tid, [&] (int e) {
            const unsigned al = (unsigned) e & 31u, kbl = min (((unsigned) e >> 5) % CB, last), tA = ((unsigned) e >> 5) / CB;
            return &src[(tA * NR + kb0 + kbl) * 32u + al];
        });
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t v = blockIdx.x; v < nwork; v += gridDim.x)
    {
        const size_t w = r4g_row_order (v, nwork);
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e < total)
            {
                const unsigned al = (unsigned) e & 31u, kbl = ((unsigned) e >> 5) % CB, tA = ((unsigned) e >> 5) / CB;
                bufA[kbl * lfs + tA * 32 + al] = (int) kbl < valid ? pre[u] : make_float2 (0.f, 0.f);
            }
        }
        __syncthreads();
        // this tile's filter bins (element (row c, index k), row fastest: 16 contiguous bins per access) and its inter-pass twiddles FIRST,
        // the next tile's operands behind them: a wait for the former must not wait for the latter (one in-order counter)
        float2 hv[SLOTS], tw[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = min (tid + u * BLOCK, total - 1);
            const int c = min (e % CB, valid - 1), k = e / CB;
            const unsigned kb = kb0 + (unsigned) c;
            unsigned idx;
            if (k < R1 / 2)
                idx = (unsigned) k * n2 + kb;
            else if (kb == 0 && k == R1 / 2)
                idx = 0;
            else
                idx = (kb == 0) ? (unsigned) (R1 - k) * n2 : (unsigned) (R1 - 1 - k) * n2 + (n2 - kb);
            hv[u] = H[idx];
        }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const unsigned e = (unsigned) min (tid + u * BLOCK, total - 1);
            const unsigned al = e & 31u, kbl = min ((e >> 5) % CB, (unsigned) (valid - 1)), tA = (e >> 5) / CB;
            tw[u] = big_tw[(tA * NR + kb0 + kbl) * 32u + al];
        }
        if (v + gridDim.x < nwork)
            request (v + gridDim.x);
        float2* res = const_cast<float2*> (r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CB, tid, lfs));
        // X[R2 k + kb] * H * scaling, stored swapped (operand of the backward row transform) in place
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= total)
                continue;
            const int c = e % CB, k = e / CB;
            const unsigned kb = kb0 + (unsigned) c;
            const float2 y = res[c * lfs + k];
            float2 h = hv[u];
            if (kb == 0 && k == 0)
                h = make_float2 (h.x, 0.f); // X0 * H0 (both real)
            else if (kb == 0 && k == R1 / 2)
                h = make_float2 (h.y, 0.f); // X_{N/2} * H_{N/2}
            else if (k >= R1 / 2)
                h = make_float2 (h.x, -h.y); // H[k'] = conj H[N - k'] beyond N/2
            const float2 z = cmul (y, make_float2 (h.x * scaling, h.y * scaling));
            res[c * lfs + k] = c < valid ? make_float2 (z.y, z.x) : make_float2 (0.f, 0.f);
        }
        __syncthreads();
        const float2* back = r4g_core<BLOCK, SP> (plan, twl, res, res == bufA ? bufB : bufA, CB, tid, lfs);
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= total)
                continue;
            const unsigned al = (unsigned) e & 31u, kbl = ((unsigned) e >> 5) % CB, tA = ((unsigned) e >> 5) / CB;
            if ((int) kbl < valid)
            {
                const float2 ys = back[kbl * lfs + tA * 32 + al];
                dst[(tA * NR + kb0 + kbl) * 32u + al] = cmulc (make_float2 (ys.y, ys.x), tw[u]); // un-swap, times conj W_N^{a kb}
            }
        }
    }
}

// Backward, second sweep: the columns.  Z_c[kb] = C_2c[kb] + i C_2c+1[kb] for kb <= R2/2 and conj C_2c[R2 - kb] + i conj C_2c+1[R2 - kb]
// above; backward transform over kb, stored as the float2 column c of the real frame.
template <int BLOCK, int SLOTS, class SP = void>
__global__ __launch_bounds__ (BLOCK) void real4g_pass_a_inv_kernel (const float2* __restrict__ work,
                                                                   float2* __restrict__ out, // the real frame as N/2 float2
                                                                   const float2* __restrict__ twtab,
                                                                   GenericPlan plan, // n = R2, frames = 16
                                                                   unsigned n1c,
                                                                   unsigned batch,
                                                                   size_t out_stride,
                                                                   unsigned lfs,
                                                                   size_t ws_frame_len)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    constexpr int CA = kReal4GTile;
    const int tid = threadIdx.x;
    const int n2 = tile_frame_len<SP> (plan), NR = n2 / 2 + 1; // (SP: compile-time stage list of this length, void = the run-time list)
    if constexpr (! std::is_void<SP>::value)
        lfs = SP::n + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs;
    float2* twl = smem + (size_t) 2 * CA * lfs;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n2 - 1, tid);
    const unsigned tiles = n1c / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2, half = CA * NR;
    float4 pre[SLOTS / 2];
    const auto request = [&] (size_t w) {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float4* __restrict__ src = reinterpret_cast<const float4*> (work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) NR * CA * 2));
        r4g_request<BLOCK> (pre, half, tid, [&] (int e) { return &src[e]; }); // (C_2c[kb], C_2c+1[kb])
    };
    if (blockIdx.x < nwork)
        request (blockIdx.x);
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        float2* __restrict__ dst = out + (size_t) f * out_stride + (size_t) tile * CA;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLOTS / 2; ++u)
        {
            const int e = tid + u * BLOCK;
            if (e >= half)
                continue;
            const int c = e % CA, kb = e / CA;
            const float4 p = pre[u];
            // Z[kb] = Ce + i Co, Z[R2 - kb] = conj Ce + i conj Co; stored swapped for the backward transform
            const float2 zk = make_float2 (p.x - p.w, p.y + p.z);
            const float2 zm = make_float2 (p.x + p.w, p.z - p.y);
            bufA[c * lfs + kb] = make_float2 (zk.y, zk.x);
            if (kb != 0 && 2 * kb != n2)
                bufA[c * lfs + (n2 - kb)] = make_float2 (zm.y, zm.x);
        }
        __syncthreads();
        if (w + gridDim.x < nwork)
            request (w + gridDim.x);
        const float2* res = r4g_core<BLOCK, SP> (plan, twl, bufA, bufB, CA, tid, lfs);
        for (int e = tid; e < total; e += BLOCK)
        {
            const int c = e % CA, j = e / CA;
            const float2 y = res[c * lfs + j];
            gstore (&dst[(unsigned) j * n1c + (unsigned) c], make_float2 (y.y, y.x));
        }
    }
}
} // namespace cfft
