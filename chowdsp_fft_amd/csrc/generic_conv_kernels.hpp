// Fused convolution of mixed-radix frames: forward transform -> * H * scaling -> backward transform in ONE launch, the
// spectrum never leaving the workgroup's LDS image(s).  The call sequence it replaces (reference test/test.cpp:161-166,
// 212-218: fft_transform_unordered FORWARD, fft_convolve_unordered, fft_transform_unordered BACKWARD) ran as three launches
// through `y` for these sizes until round 2: three reads and three writes of every frame instead of one each.
//
// Built from the stages of generic_kernels.hpp: the forward core reads its first-stage operands from global memory and
// leaves the spectrum -- natural order, the Stockham stages sort as they go -- in an LDS image, multiplied by H * scaling and
// re <-> im swapped by the forward transform's LAST STAGE as it writes its outputs (the backward transform is the forward one
// between two swaps, butterflies.hpp; a sweep of its own for the product cost an LDS round trip and a barrier: c480 0.435 ->
// see profiles/r03_conv_frames_rates.txt); the backward core starts from that image and its last stage stores to global
// memory, swapped back.  REAL frames (N = 2 n samples as n
// complex values): the pair stage of the forward transform, the product with the PACKED spectrum H -- DC and Nyquist as
// two real products, reference simd/chowdsp_fft_impl_sse.cpp:1640-1644 -- and the pair stage of the backward transform
// collapse into one sweep over the pairs (k, n - k), each pair read and rewritten by one work-item.
// Two forms, as for the plain transforms: several short frames per workgroup in two images with a compile-time stage list
// (the thirty lengths up to 2048 points, the reference's "other sizes" 96 ... 768 among them), one long frame per workgroup in
// ONE image with in-place stages (compile-time list for 4608 / 9216, run-time list otherwise, up to 20000 points).
#pragma once
#include "generic_kernels.hpp"

namespace cfft
{
// spectrum sweep of `count` real frames of n complex points at `img` (frame f at f * n): split -> * H -> merge on every pair (k, n - k), k = 0 .. n / 2; w = the pair twiddles W_N^k (LDS or L2)
__device__ __forceinline__ void conv_sweep_real (float2* __restrict__ img, const float2* __restrict__ H, const float2* __restrict__ w, float scaling, int n, int count, int tid, int block)
{
    const int half = n / 2 + 1;
    const unsigned magic_half = div_magic ((unsigned) half);
    for (int idx = tid; idx < count * half; idx += block)
    {
        const int f = (count == 1) ? 0 : fast_div (idx, magic_half), k = idx - f * half;
        const int kp = (k == 0) ? 0 : n - k;
        float2* fr = img + (size_t) f * n;
        const float2 A = fr[k];
        const float2 Zp = fr[kp];
        const float2 hk = H[k], hp = H[kp];
        float2 zk, zp;
        if (k == 0)
        {
            // packed bin 0 = (X0, X_{N/2}), both real: real products
            const float y0 = (A.x + A.y) * hk.x * scaling, ym = (A.x - A.y) * hk.y * scaling;
            zk = make_float2 (y0 + ym, y0 - ym);
            zp = zk;
        }
        else if (2 * k == n)
        {
            const float2 y = cmul (make_float2 (A.x, -A.y), make_float2 (hk.x * scaling, hk.y * scaling)); // X[n/2] = conj Z[n/2]
            zk = make_float2 (2.f * y.x, -2.f * y.y);
            zp = zk;
        }
        else
        {
            float2 Xk, Xp;
            real_split_pair (A, Zp, w[k], Xk, Xp);
            const float2 Yk = cmul (Xk, make_float2 (hk.x * scaling, hk.y * scaling));
            const float2 Yp = cmul (Xp, make_float2 (hp.x * scaling, hp.y * scaling));
            real_merge_pair (Yk, Yp, w[k], zk, zp);
        }
        fr[k] = make_float2 (zk.y, zk.x);
        if (kp != k)
            fr[kp] = make_float2 (zp.y, zp.x);
    }
}

// ---- short frames: several per workgroup, two images, compile-time stage list ------------------------------------------
template <bool REAL, int BLOCK, class SP>
__global__ __launch_bounds__ (BLOCK) void generic_conv_kernel (const float2* __restrict__ in,
                                                              const float2* __restrict__ H,
                                                              float2* __restrict__ out,
                                                              const float2* __restrict__ twtab,
                                                              const float2* __restrict__ wtab,
                                                              GenericPlan plan,
                                                              float scaling,
                                                              unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    const int tid = threadIdx.x;
    constexpr int n = SP::n;
    const int F = plan.frames;
    float2* bufA = smem;
    float2* bufB = smem + (size_t) F * n;
    float2* twl = smem + (size_t) 2 * F * n;
    float2* wlds = twl + (n - 1);
    const bool pair_lds = REAL && plan.pair_lds != 0;
    lds_table_copy<BLOCK> (twl, twtab + plan.tw_offset[0], n - 1, tid);
    if (pair_lds)
        lds_table_copy<BLOCK> (wlds, wtab, n / 2 + 1, tid);
    const unsigned ngroups = (batch + F - 1) / F;
    // software pipeline (generic_kernels.hpp: generic_first_prefetch): the next group's first-stage operands are requested while
    // this group's remaining forward stages, the product and the whole backward transform run
    constexpr int R0 = SP::radix (0);
    constexpr bool PIPE = CFFT_GENERIC_CONV_PIPE != 0;
    float2 pre[R0];
    if constexpr (PIPE)
    {
        if (blockIdx.x < ngroups)
            generic_first_prefetch<R0, BLOCK, false> (pre, in + (size_t) blockIdx.x * F * n, n, (int) min ((unsigned) F, batch - blockIdx.x * F), tid, (unsigned) n);
    }
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int nframes = (int) min ((unsigned) F, batch - g * F);
        const float2* __restrict__ src = in + (size_t) g * F * n;
        float2* __restrict__ dst = out + (size_t) g * F * n;
        __syncthreads(); // previous group's reads of the images are finished (and, first time round, the tables are written)
        // complex frames: the product rides on the forward transform's last stage (its outputs are multiplied as they are
        // written to the image); real frames need both bins of a pair and get a sweep of their own
        float2* res;
        if constexpr (PIPE)
        {
            generic_first_stage_pre<R0, BLOCK, false> (pre, src, bufB, n, nframes, tid, (unsigned) n, (unsigned) n);
            __syncthreads();
            const unsigned gn = g + gridDim.x;
            if (gn < ngroups)
                generic_first_prefetch<R0, BLOCK, false> (pre, in + (size_t) gn * F * n, n, (int) min ((unsigned) F, batch - gn * F), tid, (unsigned) n);
            res = static_core<SP, BLOCK, true, false, false, 1, ! REAL> (twl, bufB, bufA, src, dst, nframes, tid, (unsigned) n, (unsigned) n, H, scaling);
        }
        else
            res = static_core<SP, BLOCK, true, false, false, 0, ! REAL> (twl, bufA, bufB, src, dst, nframes, tid, (unsigned) n, (unsigned) n, H, scaling);
        float2* other = (res == bufA) ? bufB : bufA;
        if constexpr (REAL)
        {
            if (pair_lds)
                conv_sweep_real (res, H, wlds, scaling, n, nframes, tid, BLOCK);
            else
                conv_sweep_real (res, H, wtab, scaling, n, nframes, tid, BLOCK);
            __syncthreads();
        }
        (void) static_core<SP, BLOCK, false, true, true> (twl, res, other, src, dst, nframes, tid, (unsigned) n, (unsigned) n);
    }
}

// ---- long frames: one per workgroup, one image, stages in place ---------------------------------------------------------
template <bool REAL, int BLOCK, bool TWLDS, class SP = void>
__global__ __launch_bounds__ (BLOCK, BLOCK == 512 ? 4 : 1) void generic_inplace_conv_kernel (const float2* __restrict__ in,
                                                                                            const float2* __restrict__ H,
                                                                                            float2* __restrict__ out,
                                                                                            const float2* __restrict__ twtab,
                                                                                            const float2* __restrict__ wtab,
                                                                                            GenericPlan plan,
                                                                                            float scaling,
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
    }
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        const float2* __restrict__ src = in + (size_t) g * n;
        float2* __restrict__ dst = out + (size_t) g * n;
        int tid = tid0;
        CFFT_LAUNDER_INDEX (tid);
        __syncthreads(); // previous frame's reads of the image are finished
        if constexpr (STATIC)
            static_inplace_core<SP, BLOCK, true, false, false, TWLDS, 0, ! REAL> (plan, twtab, twl, buf, src, dst, tid, H, scaling);
        else
            inplace_core<BLOCK, true, false, false, TWLDS, ! REAL> (plan, twtab, twl, buf, src, dst, tid, H, scaling);
        // (every in-place stage that writes the image ends with a barrier: the spectrum -- complex frames: already multiplied
        // and swapped by the last stage -- is complete here)
        if constexpr (REAL)
        {
            conv_sweep_real (buf, H, wtab, scaling, n, 1, tid, BLOCK);
            __syncthreads();
        }
        if constexpr (STATIC)
            static_inplace_core<SP, BLOCK, false, true, true, TWLDS> (plan, twtab, twl, buf, src, dst, tid);
        else
            inplace_core<BLOCK, false, true, true, TWLDS> (plan, twtab, twl, buf, src, dst, tid);
    }
}
} // namespace cfft
