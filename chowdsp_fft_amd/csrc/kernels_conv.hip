// Instantiation + launch of the frequency-domain multiply-accumulate / accumulate / fused convolution kernels.
#include "conv_kernels.hpp"
#include "launch_util.hpp"

namespace cfft::host
{
namespace
{
unsigned stream_grid (size_t items)
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
    const size_t want = (items + 255) / 256;
    return (unsigned) std::max<size_t> (1, std::min<size_t> (want, (size_t) dev.num_cus * 8));
}

template <int N>
hipError_t launch_cf_c2c_n (const float2* in, const float2* H, float2* out, const float2* tw, float scaling, size_t batch, hipStream_t stream)
{
    using P = typename CorePlan<N>::type;
    using G = Geometry<P>;
    // CFFT_CCONV_PF_MIN: smallest N that takes the software-pipelined kernel; CFFT_CCONV_FPB2048: frames per workgroup
    // of the N = 2048 plan (see launch_cf_real_m).  Measured on MI355X (profiles/r01_variants_cconv_pf.txt):
    // N=4096 4.89 -> 5.20 TB/s, N=2048 with 128-thread workgroups 4.97 -> 5.53 TB/s, N <= 1024 unchanged or slower.
#ifndef CFFT_CCONV_PF_MIN
#define CFFT_CCONV_PF_MIN 2048
#endif
#ifndef CFFT_CCONV_FPB2048
#define CFFT_CCONV_FPB2048 1
#endif
    constexpr int FPB = (N == 2048) ? CFFT_CCONV_FPB2048 : G::FPB;
    constexpr int threads = P::T * FPB;
    constexpr size_t lds_bytes = G::lds_bytes / G::FPB * FPB;
    constexpr bool use_pf = N >= CFFT_CCONV_PF_MIN;
    auto kernel = [] { // only the variant in use is instantiated
        if constexpr (use_pf)
            return &conv_frames_c2c_kernel_pf<P, FPB>;
        else
            return &conv_frames_c2c_kernel<P, FPB>;
    }();
    const size_t ngroups = (batch + FPB - 1) / FPB;
    const unsigned grid = persistent_grid (kernel, threads, lds_bytes, ngroups);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (threads), lds_bytes, stream, in, H, out, tw, scaling, (unsigned) batch);
    return hipGetLastError();
}

template <int M>
hipError_t launch_cf_real_m (const float2* in, const float2* H, float2* out, const float2* tw, const float2* rtw, float scaling, size_t batch, hipStream_t stream)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    // Variant per plan, from A/B timing on MI355X (profiles/r01_variants_conv_hregs_pf.txt):
    //  - software prefetch of the next group (conv_frames_real_kernel_pf) pays for M = 1024 (+10 %) and M = 4096 (+6 %),
    //    costs 2 % for M <= 512, and needs more than 256 VGPRs for M = 2048 unless the filter stays in L2;
    //  - CFFT_CONV_FPB2048: frames per workgroup of the M = 2048 plan (T = 128): 1 -> 128-thread workgroups, so that a
    //    barrier stalls two waves instead of four.
#ifndef CFFT_CONV_PREFETCH
#define CFFT_CONV_PREFETCH 2
#endif
#ifndef CFFT_CONV_PF2048
#define CFFT_CONV_PF2048 1
#endif
#ifndef CFFT_CONV_FPB2048
#define CFFT_CONV_FPB2048 1
#endif
    constexpr int FPB = (M == 2048) ? CFFT_CONV_FPB2048 : G::FPB;
    constexpr bool use_pf = (CFFT_CONV_PREFETCH == 1) || (CFFT_CONV_PREFETCH == 2 && (M == 1024 || M == 4096 || (M == 2048 && CFFT_CONV_PF2048)));
    constexpr bool hregs = ! (use_pf && M == 2048);
    constexpr int threads = P::T * FPB;
#ifndef CFFT_CONV_REAL_TP
#define CFFT_CONV_REAL_TP 1
#endif
    constexpr bool use_tp_decl = CFFT_CONV_REAL_TP && use_pf && kRealPairInRegs<P> && M == 2048 && ! CFFT_CONV_STREAM_TW;
    constexpr size_t lds_bytes = G::lds_bytes / G::FPB * FPB // no staging area for T >= 16
                                 + (use_tp_decl ? (size_t) (P::radix (1) - 1) * P::prod (1) * sizeof (float2) // + the stage-1 twiddle table of the round-4 kernel
                                                      + (size_t) 2 * CFFT_CONV_TP_MAP_LDS * P::T * sizeof (float4) // + the part of the pair map kept in LDS
                                                : 0);
    // round 4 (CFFT_CONV_REAL_TP, default 1): pair stage in registers + transposed backward network for the plans whose last
    // stage gives a work-item an even number of butterflies (M = 1024, 2048)
    // (M = 1024, whose last radix-4 stage gives a work-item four butterflies, spills 144 bytes in this form: not taken)
    constexpr bool use_tp = use_tp_decl;
    auto kernel = [] { // only the variant in use is instantiated
        if constexpr (use_tp)
            return &conv_frames_real_tp_kernel_pf<P, FPB>;
        else if constexpr (use_pf)
            return &conv_frames_real_kernel_pf<P, FPB, hregs>;
        else
            return &conv_frames_real_kernel<P, FPB, hregs>;
    }();
    const size_t ngroups = (batch + FPB - 1) / FPB;
    const unsigned grid = persistent_grid (kernel, threads, lds_bytes, ngroups);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (threads), lds_bytes, stream, in, H, out, tw, rtw, scaling, (unsigned) batch);
    return hipGetLastError();
}
} // namespace

hipError_t launch_conv_mac (bool real_packed, const float* a, const float* b, float* ab, float scaling, size_t cplx_per_frame, size_t batch, bool b_is_shared, hipStream_t stream, bool accumulate)
{
    const size_t total_pairs = batch * cplx_per_frame / 2;
    if (total_pairs == 0)
        return hipSuccess;
    const unsigned grid = stream_grid (total_pairs);
    if (grid == 0)
        return hipErrorNoDevice;
    auto kernel = real_packed ? (accumulate ? conv_mac_kernel<true, true> : conv_mac_kernel<true, false>)
                              : (accumulate ? conv_mac_kernel<false, true> : conv_mac_kernel<false, false>);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (256), 0, stream, (const float4*) a, (const float4*) b, (float4*) ab, scaling, (unsigned) (cplx_per_frame / 2), total_pairs, b_is_shared ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_accumulate (const float* a, const float* b, float* ab, size_t count, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    const unsigned grid = stream_grid ((count + 3) / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (accumulate_kernel, dim3 (grid), dim3 (256), 0, stream, a, b, ab, count);
    return hipGetLastError();
}

hipError_t launch_fdl_mac (const float* fdl, const float* H, float* Y, float scale, size_t frame_floats, unsigned channels, unsigned P, unsigned head, bool h_shared, hipStream_t stream)
{
    const unsigned ppf = (unsigned) (frame_floats / 4);
    const unsigned grid = stream_grid ((size_t) channels * ppf);
    if (grid == 0)
        return hipErrorNoDevice;
#ifndef CFFT_FDL_UNROLL
#define CFFT_FDL_UNROLL 4
#endif
    hipLaunchKernelGGL (fdl_mac_kernel<CFFT_FDL_UNROLL>, dim3 (grid), dim3 (256), 0, stream, (const float4*) fdl, (const float4*) H, (float4*) Y, scale, ppf, channels, P, head, h_shared ? 1 : 0);
    return hipGetLastError();
}

namespace
{
template <int M>
hipError_t launch_ols_r2c_m (float2* buf0, float2* buf1, const float2* cur, float2* fdl, const float2* tw, const float2* rtw, unsigned channels, const OlsState* state, hipStream_t stream)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    {
        // narrow teams (block < 128) load and store directly too: they touch 32- or 64-byte pieces of a line per
        // instruction, but a block is two launches instead of five, which is what counts at these sizes
        auto kernel = ols_r2c_kernel<P, G::FPB>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, (channels + G::FPB - 1) / G::FPB);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, buf0, buf1, cur, fdl, tw, rtw, channels, state);
        return hipGetLastError();
    }
}

template <int M>
hipError_t launch_ols_mac_c2r_m (const float2* fdl, const float2* H, float2* out, const float2* tw, const float2* rtw, float scale, unsigned channels, unsigned P_, OlsState* state, bool h_shared, hipStream_t stream)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    {
        auto kernel = ols_mac_c2r_kernel<P, G::FPB>;
        const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, (channels + G::FPB - 1) / G::FPB);
        if (grid == 0)
            return hipErrorNoDevice;
        hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, fdl, H, out, tw, rtw, scale, channels, P_, state, h_shared ? 1 : 0);
        return hipGetLastError();
    }
}
} // namespace

namespace
{
template <int M>
hipError_t launch_ols_block_m (float2* buf0, float2* buf1, const float2* cur, float2* fdl, const float2* H, float2* out, const float2* tw, const float2* rtw, float scale, unsigned channels, unsigned P_, OlsState* state, bool h_shared, hipStream_t stream)
{
    using P = typename CorePlan<M>::type;
    using G = Geometry<P>;
    auto kernel = ols_block_kernel<P, G::FPB>;
    const unsigned grid = persistent_grid (kernel, G::threads, G::lds_bytes, (channels + G::FPB - 1) / G::FPB);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, buf0, buf1, cur, fdl, H, out, tw, rtw, scale, channels, P_, state, h_shared ? 1 : 0);
    return hipGetLastError();
}
} // namespace

hipError_t launch_ols_block (int m, float* buf0, float* buf1, const float* cur, float* fdl, const float* H, float* out, const float2* stage_tw, const float2* real_tw, float scale, unsigned channels, unsigned P, void* state, bool h_shared, hipStream_t stream)
{
    CFFT_DISPATCH_N (m, return (launch_ols_block_m<N_> ((float2*) buf0, (float2*) buf1, (const float2*) cur, (float2*) fdl, (const float2*) H, (float2*) out, stage_tw, real_tw, scale, channels, P, (OlsState*) state, h_shared, stream)));
    return hipErrorInvalidValue;
}

bool ols_fused_supported (int m)
{
    PlanDesc d;
    return m <= kMaxFusedN && core_plan_desc (m, d);
}

hipError_t launch_ols_r2c (int m, float* buf0, float* buf1, const float* cur, float* fdl, const float2* stage_tw, const float2* real_tw, unsigned channels, const void* state, hipStream_t stream)
{
    CFFT_DISPATCH_N (m, return (launch_ols_r2c_m<N_> ((float2*) buf0, (float2*) buf1, (const float2*) cur, (float2*) fdl, stage_tw, real_tw, channels, (const OlsState*) state, stream)));
    return hipErrorInvalidValue;
}

hipError_t launch_ols_mac_c2r (int m, const float* fdl, const float* H, float* out, const float2* stage_tw, const float2* real_tw, float scale, unsigned channels, unsigned P, void* state, bool h_shared, hipStream_t stream)
{
    CFFT_DISPATCH_N (m, return (launch_ols_mac_c2r_m<N_> ((const float2*) fdl, (const float2*) H, (float2*) out, stage_tw, real_tw, scale, channels, P, (OlsState*) state, h_shared, stream)));
    return hipErrorInvalidValue;
}

hipError_t launch_ols_advance (void* state, unsigned partitions, hipStream_t stream)
{
    hipLaunchKernelGGL (ols_advance_kernel, dim3 (1), dim3 (64), 0, stream, (OlsState*) state, partitions);
    return hipGetLastError();
}

size_t ols_state_bytes() { return sizeof (OlsState); }

hipError_t launch_ols_assemble (const float* in, float* prev, float* frame, size_t block, unsigned channels, hipStream_t stream)
{
    const unsigned grid = stream_grid ((size_t) channels * block / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (ols_assemble_kernel, dim3 (grid), dim3 (256), 0, stream, (const float4*) in, (float4*) prev, (float4*) frame, (unsigned) (block / 4), channels);
    return hipGetLastError();
}

hipError_t launch_ols_extract (const float* y, float* out, size_t block, unsigned channels, hipStream_t stream)
{
    const unsigned grid = stream_grid ((size_t) channels * block / 4);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (ols_extract_kernel, dim3 (grid), dim3 (256), 0, stream, (const float4*) y, (float4*) out, (unsigned) (block / 4), channels);
    return hipGetLastError();
}

hipError_t launch_conv_frames_c2c (int n, const float2* in, const float2* H, float2* out, const float2* stage_tw, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_N (n, return (launch_cf_c2c_n<N_> (in, H, out, stage_tw, scaling, batch, stream)));
    return hipErrorInvalidValue;
}

hipError_t launch_conv_frames_real (int m, const float2* in, const float2* H, float2* out, const float2* stage_tw, const float2* real_tw, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    CFFT_DISPATCH_N (m, return (launch_cf_real_m<N_> (in, H, out, stage_tw, real_tw, scaling, batch, stream)));
    return hipErrorInvalidValue;
}
} // namespace cfft::host
