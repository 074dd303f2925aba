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
constexpr int kReal4GTile = 16;   // complex columns (pass A) / rows (pass B) per tile
constexpr int kReal4GLoads = 4;   // loads in flight per work-item while a tile is brought in

template <int BLOCK>
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
    const int n2 = plan.n, NR = n2 / 2 + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs;
    const unsigned tiles = n1c / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float2* __restrict__ src = in + (size_t) f * in_stride + (size_t) tile * CA;
        float4* __restrict__ dst = reinterpret_cast<float4*> (work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) NR * CA * 2));
        const float4* __restrict__ twp = reinterpret_cast<const float4*> (big_tw + (size_t) tile * ((size_t) NR * CA * 2));
        __syncthreads(); // previous tile fully stored
        for (int e0 = tid; e0 < total; e0 += BLOCK * kReal4GLoads)
        {
            float2 x[kReal4GLoads];
#pragma unroll
            for (int u = 0; u < kReal4GLoads; ++u)
            {
                const int e = min (e0 + u * BLOCK, total - 1);
                x[u] = gload (&src[(size_t) (e / CA) * n1c + (e % CA)]);
            }
#pragma unroll
            for (int u = 0; u < kReal4GLoads; ++u)
            {
                const int e = e0 + u * BLOCK;
                if (e < total)
                    bufA[(e % CA) * lfs + e / CA] = x[u];
            }
        }
        __syncthreads();
        const float2* res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, CA, tid, lfs);
        for (int e = tid; e < CA * NR; e += BLOCK)
        {
            const int c = e % CA, kb = e / CA;
            const float2 zk = res[c * lfs + kb];
            const float2 zm = res[c * lfs + (kb == 0 ? 0 : n2 - kb)];
            // even column: (zk + conj zm) / 2;  odd column: (zk - conj zm) / (2 i)
            const float2 ce = make_float2 (0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            const float2 co = make_float2 (0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
            const float4 w4 = twp[e];
            const float2 re = cmul (ce, make_float2 (w4.x, w4.y)), ro = cmul (co, make_float2 (w4.z, w4.w));
            dst[e] = make_float4 (re.x, re.y, ro.x, ro.y); // plain store: the workspace should stay cached
        }
    }
}

template <int BLOCK>
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
    const int R1 = plan.n;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs;
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * R1;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        const float2* __restrict__ src = work + (size_t) f * ws_frame_len;
        float2* __restrict__ dst = out + (size_t) f * out_stride;
        __syncthreads();
        for (int e0 = tid; e0 < total; e0 += BLOCK * kReal4GLoads)
        {
            float2 x[kReal4GLoads];
#pragma unroll
            for (int u = 0; u < kReal4GLoads; ++u)
            {
                // al fastest (32 contiguous values), then the tile's row, then pass A's tile
                const unsigned e = (unsigned) min (e0 + u * BLOCK, total - 1);
                const unsigned al = e & 31u, kbl = min ((e >> 5) % CB, (unsigned) (valid - 1)), tA = (e >> 5) / CB;
                x[u] = src[((size_t) tA * NR + kb0 + kbl) * 32 + al];
            }
#pragma unroll
            for (int u = 0; u < kReal4GLoads; ++u)
            {
                const int e = e0 + u * BLOCK;
                if (e < total)
                {
                    const unsigned al = (unsigned) e & 31u, kbl = ((unsigned) e >> 5) % CB, tA = ((unsigned) e >> 5) / CB;
                    bufA[kbl * lfs + tA * 32 + al] = (int) kbl < valid ? x[u] : make_float2 (0.f, 0.f);
                }
            }
        }
        __syncthreads();
        const float2* res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, CB, tid, lfs);
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
                    gstore (&dst[(size_t) k * n2 + kb], y);
            }
            else if (kb != 0 && 2 * kb != n2)
                gstore (&dst[(size_t) (R1 - 1 - k) * n2 + (n2 - kb)], make_float2 (y.x, -y.y));
        }
    }
}

// Backward, first sweep: row kb (0 .. R2/2) of the spectrum gathered from the packed input, transformed backward (re <-> im swap on
// both sides), times conj W_N^{a kb}, into the workspace
template <int BLOCK>
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
    const int R1 = plan.n;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CB * lfs;
    const unsigned NR = n2 / 2 + 1;
    const unsigned tiles = (NR + CB - 1) / CB;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CB * R1;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const unsigned kb0 = tile * CB;
        const int valid = (int) min ((unsigned) CB, NR - kb0);
        const float2* __restrict__ src = in + (size_t) f * in_stride;
        float2* __restrict__ dst = work + (size_t) f * ws_frame_len;
        __syncthreads();
        for (int e0 = tid; e0 < total; e0 += BLOCK * kReal4GLoads)
        {
            float2 x[kReal4GLoads];
#pragma unroll
            for (int u = 0; u < kReal4GLoads; ++u)
            {
                const int e = min (e0 + u * BLOCK, total - 1);
                const int c = min (e % CB, valid - 1), k = e / CB; // row fastest: 16 contiguous bins per access
                const unsigned kb = kb0 + (unsigned) c;
                // bins R2 k + kb directly for k < R1/2; X_{N/2} from slot 0; the rest as conjugates of X[N - R2 k - kb] (rows 0 and R2/2
                // mirror onto themselves)
                size_t idx;
                if (k < R1 / 2)
                    idx = (size_t) k * n2 + kb;
                else if (kb == 0 && k == R1 / 2)
                    idx = 0;
                else
                    idx = (kb == 0) ? (size_t) (R1 - k) * n2 : (size_t) (R1 - 1 - k) * n2 + (n2 - kb);
                x[u] = gload (&src[idx]);
            }
#pragma unroll
            for (int u = 0; u < kReal4GLoads; ++u)
            {
                const int e = e0 + u * BLOCK;
                if (e >= total)
                    continue;
                const int c = e % CB, k = e / CB;
                const unsigned kb = kb0 + (unsigned) c;
                float2 v = x[u];
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
        }
        __syncthreads();
        const float2* res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, CB, tid, lfs);
        for (int e = tid; e < total; e += BLOCK)
        {
            const unsigned al = (unsigned) e & 31u, kbl = ((unsigned) e >> 5) % CB, tA = ((unsigned) e >> 5) / CB;
            if ((int) kbl < valid)
            {
                const float2 ys = res[kbl * lfs + tA * 32 + al];
                const size_t idx = ((size_t) tA * NR + kb0 + kbl) * 32 + al;
                dst[idx] = cmulc (make_float2 (ys.y, ys.x), big_tw[idx]); // un-swap, times conj W_N^{a kb}
            }
        }
    }
}

// Backward, second sweep: the columns.  Z_c[kb] = C_2c[kb] + i C_2c+1[kb] for kb <= R2/2 and conj C_2c[R2 - kb] + i conj C_2c+1[R2 - kb]
// above; backward transform over kb, stored as the float2 column c of the real frame.
template <int BLOCK>
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
    const int n2 = plan.n, NR = n2 / 2 + 1;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) CA * lfs;
    const unsigned tiles = n1c / CA;
    const size_t nwork = (size_t) batch * tiles;
    const int total = CA * n2;
    for (size_t w = blockIdx.x; w < nwork; w += gridDim.x)
    {
        const unsigned f = (unsigned) (w / tiles), tile = (unsigned) (w % tiles);
        const float4* __restrict__ src = reinterpret_cast<const float4*> (work + (size_t) f * ws_frame_len + (size_t) tile * ((size_t) NR * CA * 2));
        float2* __restrict__ dst = out + (size_t) f * out_stride + (size_t) tile * CA;
        __syncthreads();
        for (int e0 = tid; e0 < CA * NR; e0 += BLOCK * 2)
        {
            float4 p[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                p[u] = src[min (e0 + u * BLOCK, CA * NR - 1)]; // (C_2c[kb], C_2c+1[kb])
#pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const int e = e0 + u * BLOCK;
                if (e >= CA * NR)
                    continue;
                const int c = e % CA, kb = e / CA;
                // Z[kb] = Ce + i Co, Z[R2 - kb] = conj Ce + i conj Co; stored swapped for the backward transform
                const float2 zk = make_float2 (p[u].x - p[u].w, p[u].y + p[u].z);
                const float2 zm = make_float2 (p[u].x + p[u].w, p[u].z - p[u].y);
                bufA[c * lfs + kb] = make_float2 (zk.y, zk.x);
                if (kb != 0 && 2 * kb != n2)
                    bufA[c * lfs + (n2 - kb)] = make_float2 (zm.y, zm.x);
            }
        }
        __syncthreads();
        const float2* res = generic_core<BLOCK, false, false, false> (plan, twtab, bufA, bufB, nullptr, nullptr, CA, tid, lfs);
        for (int e = tid; e < total; e += BLOCK)
        {
            const int c = e % CA, j = e / CA;
            const float2 y = res[c * lfs + j];
            gstore (&dst[(size_t) j * n1c + c], make_float2 (y.y, y.x));
        }
    }
}
} // namespace cfft
