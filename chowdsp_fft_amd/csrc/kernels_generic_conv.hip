// Instantiation + launch of the fused convolution kernels for mixed-radix frames (generic_conv_kernels.hpp).
#include "generic_conv_kernels.hpp"
#include "generic_static_inplace.hpp"
#include "tile_static.hpp"
#include "launch_util.hpp"
#include "tables.hpp"

#include <cstdlib>

namespace cfft::host
{
namespace
{
template <bool REAL, class SP>
hipError_t launch_conv_static (GenericPlan plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream)
{
    constexpr int BLOCK = 256;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    auto kernel = generic_conv_kernel<REAL, BLOCK, SP>;
    const size_t base = (size_t) 2 * plan.frames * plan.n + (size_t) plan.n; // two images + the n - 1 stage twiddles
    const size_t with_pair = base + (size_t) plan.n / 2 + 1;
    // the pair twiddles ride in LDS for the short frames, as in launch_generic_mb (kernels_generic.hip)
    plan.pair_lds = (REAL && plan.n <= 640 && with_pair * sizeof (float2) <= dev.lds_per_cu) ? 1 : 0;
    const size_t lds = (plan.pair_lds ? with_pair : base) * sizeof (float2);
    if (lds > dev.lds_per_cu)
        return hipErrorNotSupported;
    if (lds > 64 * 1024)
    {
        const hipError_t e = hipFuncSetAttribute ((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess)
            return e;
    }
    const size_t ngroups = (batch + plan.frames - 1) / plan.frames;
    const unsigned grid = persistent_grid (kernel, BLOCK, lds, ngroups);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (BLOCK), lds, stream, in, H, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, scaling, (unsigned) batch);
    return hipGetLastError();
}

template <bool REAL, int BLOCK, bool TWLDS, class SP = void>
hipError_t launch_conv_inplace_tw (const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream)
{
    auto kernel = generic_inplace_conv_kernel<REAL, BLOCK, TWLDS, SP>;
    const size_t lds = ((size_t) plan.n + (TWLDS ? (size_t) plan.lds_tw_len : 0)) * sizeof (float2);
    const unsigned grid = persistent_grid (kernel, BLOCK, lds, batch); // raises the dynamic-LDS cap when lds > 64 KiB
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (BLOCK), lds, stream, in, H, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, scaling, (unsigned) batch);
    return hipGetLastError();
}

// as launch_generic_inplace (kernels_generic.hip): the tables of as many leading stages as fit behind the image ride in LDS,
// as long as that costs no workgroup per CU
template <bool REAL, int BLOCK>
hipError_t launch_conv_inplace (GenericPlan plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream)
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    plan.lds_stages = 0;
    plan.lds_tw_len = 0;
    const size_t image = (size_t) plan.n * sizeof (float2);
    const int wgs = blocks_per_cu_cached ((const void*) generic_inplace_conv_kernel<REAL, BLOCK, false>, BLOCK, image);
    const size_t per_wg = dev.lds_per_cu / (size_t) std::max (1, wgs);
    const size_t room = per_wg > image ? (per_wg - image) / sizeof (float2) : 0;
    int p = 1, stages = 0;
    size_t len = 0;
    for (int s = 0; s < plan.ns; ++s)
    {
        const size_t stage_len = (size_t) (plan.radix[s] - 1) * (size_t) p;
        if (len + stage_len > room)
            break;
        len += stage_len;
        stages = s + 1;
        p *= plan.radix[s];
    }
    const bool table_granted = stages > 1 && blocks_per_cu_cached ((const void*) generic_inplace_conv_kernel<REAL, BLOCK, true>, BLOCK, image + len * sizeof (float2)) >= wgs;
    if (table_granted)
    {
        plan.lds_stages = stages;
        plan.lds_tw_len = (int) len;
    }
    // compile-time lists with or without the leading stages' tables in LDS (kernels_generic.hip: launch_generic_inplace)
    {
        // (generic_static_inplace.hpp: every length of this kernel has a list; kernels_generic_conv_static_{a,b,c,d}.hip)
        static const bool static_off = [] {
            const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_STATIC");
            return e != nullptr && std::atoi (e) == 0;
        }();
        hipError_t rc = hipSuccess;
        if (! static_off
            && (launch_static_inplace_conv_a (REAL, BLOCK, plan, t, in, H, out, scaling, batch, stream, rc) || launch_static_inplace_conv_b (REAL, BLOCK, plan, t, in, H, out, scaling, batch, stream, rc)
                || launch_static_inplace_conv_c (REAL, BLOCK, plan, t, in, H, out, scaling, batch, stream, rc) || launch_static_inplace_conv_d (REAL, BLOCK, plan, t, in, H, out, scaling, batch, stream, rc)))
            return rc;
    }
    if (table_granted)
        return launch_conv_inplace_tw<REAL, BLOCK, true> (plan, t, in, H, out, scaling, batch, stream);
    return launch_conv_inplace_tw<REAL, BLOCK, false> (plan, t, in, H, out, scaling, batch, stream);
}

template <bool REAL>
hipError_t launch_conv_m (GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream)
{
    int block = 256;
    generic_geometry (plan.n, plan.frames, block, REAL, true);
    if (generic_uses_inplace (plan.n))
    {
        if ((plan.n + block - 1) / block > kInplaceMaxPoints)
            return hipErrorNotSupported;
        switch (block)
        {
            case 256: return launch_conv_inplace<REAL, 256> (plan, t, in, H, out, scaling, batch, stream);
            case 512: return launch_conv_inplace<REAL, 512> (plan, t, in, H, out, scaling, batch, stream);
            default: return launch_conv_inplace<REAL, 1024> (plan, t, in, H, out, scaling, batch, stream);
        }
    }
#define CFFT_TRY_STATIC_CONV(N_)                                                                                   \
    if (plan.n == N_)                                                                                              \
    {                                                                                                              \
        using SP = typename StaticPlanFor<N_>::type;                                                               \
        bool same = plan.ns == SP::ns;                                                                             \
        for (int i = 0; same && i < SP::ns; ++i)                                                                   \
            same = plan.radix[i] == SP::radix (i);                                                                 \
        if (same)                                                                                                  \
            return launch_conv_static<REAL, SP> (plan, t, in, H, out, scaling, batch, stream);                     \
    }
    CFFT_TRY_STATIC_CONV (48)
    CFFT_TRY_STATIC_CONV (80)
    CFFT_TRY_STATIC_CONV (96)
    CFFT_TRY_STATIC_CONV (144)
    CFFT_TRY_STATIC_CONV (160)
    CFFT_TRY_STATIC_CONV (192)
    CFFT_TRY_STATIC_CONV (240)
    CFFT_TRY_STATIC_CONV (288)
    CFFT_TRY_STATIC_CONV (320)
    CFFT_TRY_STATIC_CONV (384)
    CFFT_TRY_STATIC_CONV (400)
    CFFT_TRY_STATIC_CONV (432)
    CFFT_TRY_STATIC_CONV (480)
    CFFT_TRY_STATIC_CONV (576)
    CFFT_TRY_STATIC_CONV (640)
    CFFT_TRY_STATIC_CONV (720)
    CFFT_TRY_STATIC_CONV (768)
    CFFT_TRY_STATIC_CONV (800)
    CFFT_TRY_STATIC_CONV (864)
    CFFT_TRY_STATIC_CONV (960)
    CFFT_TRY_STATIC_CONV (1152)
    CFFT_TRY_STATIC_CONV (1200)
    CFFT_TRY_STATIC_CONV (1280)
    CFFT_TRY_STATIC_CONV (1296)
    CFFT_TRY_STATIC_CONV (1440)
    CFFT_TRY_STATIC_CONV (1536)
    CFFT_TRY_STATIC_CONV (1600)
    CFFT_TRY_STATIC_CONV (1728)
    CFFT_TRY_STATIC_CONV (1920)
    CFFT_TRY_STATIC_CONV (2000)
#undef CFFT_TRY_STATIC_CONV
    return hipErrorNotSupported; // a planner list the templates were not written for: the caller composes three launches
}
} // namespace

// hipErrorNotSupported: no one-launch form for this plan (the caller composes forward, multiply, backward)
hipError_t launch_generic_conv (bool real, int nc, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || nc > kGenericMaxN || t.gen_ns < 1 || t.gen_ns > kGenericMaxStages || t.mixed_n2 > 0)
        return hipErrorNotSupported;
    GenericPlan plan {};
    plan.n = nc;
    plan.ns = t.gen_ns;
    plan.in_stride = plan.out_stride = (unsigned) nc;
    for (int i = 0; i < t.gen_ns; ++i)
    {
        plan.radix[i] = t.gen_radix[i];
        plan.tw_offset[i] = t.gen_tw_offset[i];
    }
    return real ? launch_conv_m<true> (plan, t, in, H, out, scaling, batch, stream) : launch_conv_m<false> (plan, t, in, H, out, scaling, batch, stream);
}
} // namespace cfft::host
