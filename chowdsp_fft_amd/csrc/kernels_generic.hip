// Instantiation + launch of the generic mixed-radix kernel (frames that are not powers of two).
#include "generic_kernels.hpp"
#include "launch_util.hpp"
#include "tables.hpp"
#include "tile_static.hpp"

#include <cstdlib>

namespace cfft::host
{
namespace
{
// CHOWDSP_FFT_GENERIC_TW_LDS: 0 = stage twiddles from L2 everywhere (the round-1 form), default 1 = from LDS where they fit
inline bool generic_tw_lds_enabled()
{
    static const bool off = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_TW_LDS");
        return e != nullptr && std::atoi (e) == 0;
    }();
    return ! off;
}

// float2 of twiddles generic_kernel keeps in LDS: the n - 1 stage twiddles, and for the real modes the n / 2 + 1 pair twiddles
inline size_t generic_table_len (int mode, int n) { return (size_t) n + (mode != 0 ? (size_t) n / 2 + 1 : 0); }

template <int MODE, bool INV, int BLOCK, bool TWLDS, class SP = void>
hipError_t launch_generic_mb_tw (const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    auto kernel = generic_kernel<MODE, INV, BLOCK, TWLDS, SP>;
    const size_t lds = ((size_t) 2 * plan.frames * plan.n + (TWLDS ? generic_table_len (plan.pair_lds ? MODE : 0, plan.n) : 0)) * sizeof (float2);
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
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (BLOCK), lds, stream, in, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, (unsigned) batch);
    return hipGetLastError();
}

// the stage twiddles ride in LDS wherever the n - 1 values fit behind the two images (generic_kernels.hpp: TWLDS)
template <int MODE, bool INV, int BLOCK>
hipError_t launch_generic_mb (const GenericPlan& plan_in, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    const GenericPlan& plan = plan_in;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const size_t with_table = ((size_t) 2 * plan.frames * plan.n + generic_table_len (0, plan.n)) * sizeof (float2);
    if (generic_tw_lds_enabled() && with_table <= dev.lds_per_cu)
    {
        // real modes: the pair twiddles too, for the short frames (several per workgroup).  Measured on MI355X, fraction of 8 TB/s
        // with / without: real N = 96 0.50 / 0.39, 384 0.58 / 0.46, 480 0.56 / 0.45, 1280 0.63 / 0.60 -- but 3072 0.50 / 0.56,
        // 3840 0.31 / 0.45 (one frame per workgroup: the longer copy at kernel start and the LDS it takes cost more than the
        // loads from L2).  Round 5 sweep of the bound: real N = 1440 / 1536 / 1600 / 1728 +1 / +5 / +2 / +6 % with the copy, 1920 even, 2304 -25 %.
        GenericPlan plan = plan_in;
        const size_t with_pair = ((size_t) 2 * plan.frames * plan.n + generic_table_len (MODE, plan.n)) * sizeof (float2);
        static const int pair_lds_max_n = [] { // (developer sweep: CHOWDSP_FFT_PAIR_LDS_MAX_N)
            const char* e = std::getenv ("CHOWDSP_FFT_PAIR_LDS_MAX_N");
            const int v = e != nullptr ? std::atoi (e) : 0;
            return v > 0 ? v : 864; // (640 until round 5; 720 ... 864 points gain 1 ... 6 % with the copy, 960 is even, 1152 collapses: profiles/r05_generic_lists.txt step 8)
        }();
        plan.pair_lds = (MODE != 0 && plan.n <= pair_lds_max_n && with_pair <= dev.lds_per_cu) ? 1 : 0;
        // the reference's "other sizes" (generic_kernels.hpp: StaticPlanFor): compile-time stage list where the planner's list
        // (by which the twiddle table is laid out) is the one the template was written for
        static const bool static_off = [] {
            const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_STATIC");
            return e != nullptr && std::atoi (e) == 0;
        }();
        if constexpr (BLOCK == 256)
        {
            if (! static_off)
            {
#define CFFT_TRY_STATIC(N_)                                                                                        \
    if (plan.n == N_)                                                                                              \
    {                                                                                                              \
        using SP = typename StaticPlanFor<N_>::type;                                                               \
        bool same = plan.ns == SP::ns;                                                                             \
        for (int i = 0; same && i < SP::ns; ++i)                                                                   \
            same = plan.radix[i] == SP::radix (i);                                                                 \
        if (same)                                                                                                  \
            return launch_generic_mb_tw<MODE, INV, BLOCK, true, SP> (plan, t, in, out, batch, stream);             \
    }
                CFFT_TRY_STATIC (48)
                CFFT_TRY_STATIC (80)
                CFFT_TRY_STATIC (96)
                CFFT_TRY_STATIC (144)
                CFFT_TRY_STATIC (160)
                CFFT_TRY_STATIC (192)
                CFFT_TRY_STATIC (240)
                CFFT_TRY_STATIC (288)
                CFFT_TRY_STATIC (320)
                CFFT_TRY_STATIC (384)
                CFFT_TRY_STATIC (400)
                CFFT_TRY_STATIC (432)
                CFFT_TRY_STATIC (480)
                CFFT_TRY_STATIC (576)
                CFFT_TRY_STATIC (640)
                CFFT_TRY_STATIC (720)
                CFFT_TRY_STATIC (768)
                CFFT_TRY_STATIC (800)
                CFFT_TRY_STATIC (864)
                CFFT_TRY_STATIC (960)
                CFFT_TRY_STATIC (1152)
                CFFT_TRY_STATIC (1200)
                CFFT_TRY_STATIC (1280)
                CFFT_TRY_STATIC (1296)
                CFFT_TRY_STATIC (1440)
                CFFT_TRY_STATIC (1536)
                CFFT_TRY_STATIC (1600)
                CFFT_TRY_STATIC (1728)
                CFFT_TRY_STATIC (1920)
                CFFT_TRY_STATIC (2000)
#undef CFFT_TRY_STATIC
            }
        }
        return launch_generic_mb_tw<MODE, INV, BLOCK, true> (plan, t, in, out, batch, stream);
    }
    return launch_generic_mb_tw<MODE, INV, BLOCK, false> (plan, t, in, out, batch, stream);
}

template <int MODE, bool INV, int BLOCK, bool TWLDS, class SP = void>
hipError_t launch_generic_inplace_tw (const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    auto kernel = generic_inplace_kernel<MODE, INV, BLOCK, TWLDS, SP>;
    const size_t lds = ((size_t) plan.n + (TWLDS ? (size_t) plan.lds_tw_len : 0)) * sizeof (float2);
    const unsigned grid = persistent_grid (kernel, BLOCK, lds, batch); // raises the dynamic-LDS cap when lds > 64 KiB
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (BLOCK), lds, stream, in, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, (unsigned) batch);
    return hipGetLastError();
}

// The tables of as many leading stages as fit ride in LDS behind the image (generic_kernels.hpp: TWLDS) -- as long as that
// costs no workgroup per CU: the budget is the CU's LDS divided by the workgroups the plain kernel gets, and the table
// kernel (a few more registers) must reach the same count with its larger allocation.
template <int MODE, bool INV, int BLOCK>
hipError_t launch_generic_inplace (GenericPlan plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    plan.lds_stages = 0;
    plan.lds_tw_len = 0;
    if (generic_tw_lds_enabled())
    {
        const size_t image = (size_t) plan.n * sizeof (float2);
        const int wgs = blocks_per_cu_cached ((const void*) generic_inplace_kernel<MODE, INV, BLOCK, false>, BLOCK, image);
        const size_t per_wg = dev.lds_per_cu / (size_t) std::max (1, wgs);
        const size_t room = per_wg > image ? (per_wg - image) / sizeof (float2) : 0;
        int p = 1;
        size_t len = 0;
        int stages = 0;
        for (int s = 0; s < plan.ns; ++s)
        {
            const size_t stage_len = (size_t) (plan.radix[s] - 1) * (size_t) p;
            if (len + stage_len > room)
                break;
            len += stage_len;
            stages = s + 1;
            p *= plan.radix[s];
        }
        const bool table_granted = stages > 1 // stage 0 has no twiddles (p == 1)
                                   && blocks_per_cu_cached ((const void*) generic_inplace_kernel<MODE, INV, BLOCK, true>, BLOCK, image + len * sizeof (float2)) >= wgs;
        if (table_granted)
        {
            plan.lds_stages = stages;
            plan.lds_tw_len = (int) len;
        }
        // compile-time stage lists (generic_kernels.hpp: StaticInplacePlanFor), where the planner's list is the template's -- with
        // the leading stages' tables in LDS where granted, every table from L2 where not (n = 5120 and 10240: the image alone
        // takes the workgroup's share of the CU's LDS; until round 3 those two fell back to the run-time list: 9000 scalar and
        // 1850 table-load instructions per wave)
        static const bool static_off = [] {
            const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_STATIC");
            return e != nullptr && std::atoi (e) == 0;
        }();
        if (! static_off)
        {
            // (generic_static_inplace.hpp: every length of this kernel has a list; four translation units)
            hipError_t rc = hipSuccess;
            const int mode_arg = MODE;
            if (launch_static_inplace_a (mode_arg, INV, BLOCK, plan, t, in, out, batch, stream, rc) || launch_static_inplace_b (mode_arg, INV, BLOCK, plan, t, in, out, batch, stream, rc)
                || launch_static_inplace_c (mode_arg, INV, BLOCK, plan, t, in, out, batch, stream, rc) || launch_static_inplace_d (mode_arg, INV, BLOCK, plan, t, in, out, batch, stream, rc))
                return rc;
        }
        if (table_granted)
            return launch_generic_inplace_tw<MODE, INV, BLOCK, true> (plan, t, in, out, batch, stream);
    }
    return launch_generic_inplace_tw<MODE, INV, BLOCK, false> (plan, t, in, out, batch, stream);
}

template <int MODE, bool INV>
hipError_t launch_generic_m (GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    int block = 256;
    generic_geometry (plan.n, plan.frames, block, MODE != 0); // tables.hpp: the stage radices were chosen for this geometry
    // CHOWDSP_FFT_GENERIC_INPLACE (experiment): 0 = two-image kernel wherever it fits (n <= 9216), 1 = in-place kernel for
    // every n > 2048; default: see use_inplace below
    static const int force = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_INPLACE");
        return e != nullptr ? std::atoi (e) : -1;
    }();
    // measured on MI355X (fwd+inv, fraction of 8 TB/s, in-place vs two images, round 2 after the stage loop stopped carrying every
    // radix case's registers): complex n = 2304 / 4608 / 5120 / 6400 / 7680 / 9216: 0.62 / 0.59 / 0.56 / 0.51 / 0.59 / 0.55 vs
    // 0.44 / 0.36 / 0.38 / 0.41 / 0.40 / 0.43; real N = 6144 / 9216 / 18432: 0.49 / 0.49 / 0.47 vs 0.37 / 0.32 / 0.37
    bool use_inplace = generic_uses_inplace (plan.n);
    if (force == 0 && plan.n <= 9216)
        use_inplace = false;
    if (force == 1 && plan.n > 2048)
        use_inplace = true;
    if (! use_inplace && plan.n > 2048)
    {
        // the two-image kernel for a long frame: one frame per workgroup, 1024 work-items when the images exceed 64 KiB
        plan.frames = 1;
        if (2 * (size_t) plan.n * sizeof (float2) > 64 * 1024)
            return launch_generic_mb<MODE, INV, 1024> (plan, t, in, out, batch, stream);
        return launch_generic_mb<MODE, INV, 256> (plan, t, in, out, batch, stream);
    }
    if (use_inplace)
    {
        if ((plan.n + block - 1) / block > kInplaceMaxPoints)
            return hipErrorInvalidValue;
        switch (block)
        {
            case 256: return launch_generic_inplace<MODE, INV, 256> (plan, t, in, out, batch, stream);
            case 512: return launch_generic_inplace<MODE, INV, 512> (plan, t, in, out, batch, stream);
            default: return launch_generic_inplace<MODE, INV, 1024> (plan, t, in, out, batch, stream);
        }
    }
    return launch_generic_mb<MODE, INV, 256> (plan, t, in, out, batch, stream);
}
} // namespace

hipError_t launch_generic (int mode, bool inverse, int nc, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (in_stride > kMaxFrameStride || out_stride > kMaxFrameStride)
        return hipErrorInvalidValue;
    if (batch > 0x7FFFFFFFull || nc > kGenericMaxN || t.gen_ns < 1 || t.gen_ns > kGenericMaxStages)
        return hipErrorInvalidValue;
    GenericPlan plan {};
    plan.n = nc;
    plan.ns = t.gen_ns;
    plan.in_stride = in_stride ? (unsigned) in_stride : (unsigned) nc;
    plan.out_stride = out_stride ? (unsigned) out_stride : (unsigned) nc;
    for (int i = 0; i < t.gen_ns; ++i)
    {
        plan.radix[i] = t.gen_radix[i];
        plan.tw_offset[i] = t.gen_tw_offset[i];
    }
    switch (mode)
    {
        case 0: return inverse ? launch_generic_m<0, true> (plan, t, in, out, batch, stream) : launch_generic_m<0, false> (plan, t, in, out, batch, stream);
        case 1: return launch_generic_m<1, false> (plan, t, in, out, batch, stream);
        case 2: return launch_generic_m<2, true> (plan, t, in, out, batch, stream);
        default: return hipErrorInvalidValue;
    }
}

// column pass of the 2-D transform: rows-point FFTs along the columns of `batch` [rows][cols] arrays, in place
namespace
{
template <int C, int BLOCK>
ColumnFftFn pick_column_cb (bool inverse, bool pipe)
{
    return inverse ? (pipe ? column_fft_kernel<true, C, BLOCK, true> : column_fft_kernel<true, C, BLOCK, false>)
                   : (pipe ? column_fft_kernel<false, C, BLOCK, true> : column_fft_kernel<false, C, BLOCK, false>);
}
ColumnFftFn pick_column (int c, int block, bool inverse, bool pipe)
{
    if (block == 1024)
        return c == 16 ? pick_column_cb<16, 1024> (inverse, pipe) : (c == 8 ? pick_column_cb<8, 1024> (inverse, pipe) : pick_column_cb<4, 1024> (inverse, pipe));
    return c == 16 ? pick_column_cb<16, 256> (inverse, pipe) : (c == 8 ? pick_column_cb<8, 256> (inverse, pipe) : pick_column_cb<4, 256> (inverse, pipe));
}
} // namespace

hipError_t launch_column_fft (bool inverse, int rows, int cols, const DeviceTables& t, float2* data, size_t batch, hipStream_t stream, const float2* src, const float2* tw_lo, const float2* tw_hi, unsigned tw_shift)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || rows < 2 || cols < 1 || t.gen_ns < 1 || t.gen_ns > kGenericMaxStages)
        return hipErrorInvalidValue;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const int c = general_tile_width (rows);
    GenericPlan plan {};
    plan.n = rows;
    plan.ns = t.gen_ns;
    plan.frames = c;
    plan.in_stride = plan.out_stride = (unsigned) rows;
    for (int i = 0; i < t.gen_ns; ++i)
    {
        plan.radix[i] = t.gen_radix[i];
        plan.tw_offset[i] = t.gen_tw_offset[i];
    }
    unsigned lfs = 0;
    size_t lds = 0;
    int block = 0;
    StaticTileLaunch sl {};
    ColumnFftFn k = pick_static_column (plan, c, dev.lds_per_cu, inverse, sl);
    if (k != nullptr)
    {
        lfs = sl.lfs;
        lds = sl.lds_bytes;
        block = sl.block;
    }
    else
    {
        lfs = (unsigned) rows + 1u;
        if ((size_t) 2 * c * lfs * sizeof (float2) > dev.lds_per_cu)
            lfs = (unsigned) rows;
        lds = (size_t) 2 * c * lfs * sizeof (float2);
        if (lds > dev.lds_per_cu)
            return hipErrorInvalidValue;
        block = generic_tile_block (lds, dev.lds_per_cu);
        const bool pipe = tile_pipe_fits (lds, (size_t) rows, dev.lds_per_cu, 1, block);
        if (pipe)
            lds += (size_t) (rows - 1) * sizeof (float2);
        k = pick_column (c, block, inverse, pipe);
    }
    const unsigned grid = persistent_grid (k, block, lds, batch * (size_t) ((cols + c - 1) / c));
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (block), lds, stream, data, (const float2*) t.stage_twiddles, plan, (unsigned) cols, (unsigned) batch, (size_t) rows * (size_t) cols, lfs, src, tw_lo, tw_hi, tw_shift);
    return hipGetLastError();
}
} // namespace cfft::host
