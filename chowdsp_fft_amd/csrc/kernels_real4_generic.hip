// Instantiation + launch of the real four-step with generic cores (real4_generic_kernels.hpp): long mixed-radix REAL frames in two
// sweeps.  A translation unit of its own (builds in parallel with the other long-plan kernels).
#include "launch_util.hpp"
#include "real4_static.hpp"
#include "tables.hpp"

#include <cstdlib>

namespace cfft::host
{
namespace
{
GenericPlan make_plan (int n, int ns, const int* radix, const int* tw_offset)
{
    GenericPlan plan {};
    plan.n = n;
    plan.ns = ns;
    plan.frames = kReal4GTile;
    plan.in_stride = plan.out_stride = (unsigned) n;
    for (int i = 0; i < ns; ++i)
    {
        plan.radix[i] = radix[i];
        plan.tw_offset[i] = tw_offset[i];
    }
    return plan;
}
// two images of 16 frames of n + 1 points (the + 1 spreads the lanes that walk across frames over the banks)
// ... and the stage twiddles' LDS copy behind them (n - 1 entries)
size_t tile_lds (int n, unsigned& lfs)
{
    lfs = (unsigned) n + 1u;
    return ((size_t) 2 * kReal4GTile * lfs + (size_t) (n - 1)) * sizeof (float2);
}
template <class K>
unsigned tile_grid (K kernel, int block, size_t lds, size_t work, int divisor)
{
    const unsigned grid = persistent_grid (kernel, block, lds, work);
    return grid == 0 ? 0u : std::max (1u, grid / (unsigned) std::max (divisor, 1));
}
} // namespace

// real N = r1 x r2 (tables.cpp split_real4_generic); in / out: the real frame resp. its packed spectrum, both as N/2 float2;
// work: real4_workspace_frame_len (r1, r2) float2 per frame
hipError_t launch_real4_generic (const DeviceTables& t, bool inverse, const float2* in, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    const int r1 = t.real4_r1, r2 = t.real4_r2;
    if (batch > 0x7FFFFFFFull || t.real4_generic == 0 || r1 < 32 || (r1 % 32) != 0 || r2 < 2 || (r2 & 1) != 0 || t.gen_ns < 1 || t.gen2_ns < 1
        || t.gen_ns > kGenericMaxStages || t.gen2_ns > kGenericMaxStages)
        return hipErrorInvalidValue;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const size_t m = (size_t) r1 * (size_t) r2 / 2; // float2 per frame on the caller's side
    in_stride = in_stride ? in_stride : m;
    out_stride = out_stride ? out_stride : m;
    const size_t ws_len = real4_workspace_frame_len (r1, r2);
    const GenericPlan plan_c = make_plan (r2, t.gen_ns, t.gen_radix, t.gen_tw_offset);    // columns
    const GenericPlan plan_r = make_plan (r1, t.gen2_ns, t.gen2_radix, t.gen2_tw_offset); // rows
    unsigned lfs_c = 0, lfs_r = 0;
    const size_t lds_c = tile_lds (r2, lfs_c), lds_r = tile_lds (r1, lfs_r);
    if (lds_c > dev.lds_per_cu || lds_r > dev.lds_per_cu)
        return hipErrorInvalidValue;
    const int block_c = generic_tile_block (lds_c, dev.lds_per_cu), block_r = generic_tile_block (lds_r, dev.lds_per_cu);

    const int n1c = r1 / 2;
    const size_t col_tiles = (size_t) (n1c / kReal4GTile), row_tiles = ((size_t) r2 / 2 + 1 + kReal4GTile - 1) / kReal4GTile;
    const float2* tw = t.stage_twiddles;
    const float2* big = t.big_twiddles;
    // request registers per work-item: the fewest of 4 / 6 / 8 / 10 that hold a tile (fewer registers, more workgroups per CU)
    // (the column kernels also move 16 (r2 / 2 + 1) float4 -- the rows kept -- through SLOTS / 2 registers: 32 float2 more than the tile)
    const auto slots_for = [] (int n, int block, int extra) {
        const int need = (kReal4GTile * n + extra + block - 1) / block;
        return need <= 4 ? 4 : (need <= 6 ? 6 : (need <= 8 ? 8 : (need <= 10 ? 10 : 0)));
    };
    const int slots_c = slots_for (r2, block_c, 32), slots_r = slots_for (r1, block_r, 0);
    if (slots_c == 0 || slots_r == 0)
        return hipErrorInvalidValue;
#define CFFT_R4G_LAUNCH_BS(KERNEL, B_, S_, LDS, WORK, ...)                                                     \
    {                                                                                                          \
        const unsigned g_ = tile_grid (KERNEL<B_, S_>, B_, LDS, WORK, grid_divisor);                           \
        if (g_ == 0)                                                                                           \
            return hipErrorNoDevice;                                                                           \
        hipLaunchKernelGGL ((KERNEL<B_, S_>), dim3 (g_), dim3 (B_), LDS, stream, __VA_ARGS__);                 \
    }
#define CFFT_R4G_LAUNCH_B(KERNEL, B_, SLOTSV, LDS, WORK, ...)                                                  \
    switch (SLOTSV)                                                                                            \
    {                                                                                                          \
        case 4: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 4, LDS, WORK, __VA_ARGS__) break;                              \
        case 6: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 6, LDS, WORK, __VA_ARGS__) break;                              \
        case 8: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 8, LDS, WORK, __VA_ARGS__) break;                              \
        default: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 10, LDS, WORK, __VA_ARGS__) break;                            \
    }
#define CFFT_R4G_LAUNCH(KERNEL, BLOCKV, SLOTSV, LDS, WORK, ...)                                                \
    {                                                                                                          \
        if ((BLOCKV) == 1024)                                                                                  \
            CFFT_R4G_LAUNCH_B (KERNEL, 1024, SLOTSV, LDS, WORK, __VA_ARGS__)                                   \
        else                                                                                                   \
            CFFT_R4G_LAUNCH_B (KERNEL, 256, SLOTSV, LDS, WORK, __VA_ARGS__)                                    \
        const hipError_t e_ = hipGetLastError();                                                               \
        if (e_ != hipSuccess)                                                                                  \
            return e_;                                                                                         \
    }
    // a compile-time stage list for the length (kernels_real4_static_*.hip) brings its own geometry; else the run-time list
    const bool runtime_lists_only = t.real4_generic == 2; // (a plan made under option "real4_generic" = 2)
    const auto static_off = [runtime_lists_only] { static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }(); return off || runtime_lists_only; };
#define CFFT_R4G_PICK(NAME, PLAN, OUT)                                                         \
    if (! static_off())                                                                        \
    {                                                                                          \
        if ((OUT = pick_real4_##NAME##_a (PLAN, dev.lds_per_cu, sl)) == nullptr)               \
            if ((OUT = pick_real4_##NAME##_b (PLAN, dev.lds_per_cu, sl)) == nullptr)           \
                OUT = pick_real4_##NAME##_c (PLAN, dev.lds_per_cu, sl);                        \
    }
#define CFFT_R4G_STATIC_LAUNCH(FN, WORK, ...)                                                  \
    {                                                                                          \
        const unsigned g_ = tile_grid (FN, sl.block, sl.lds_bytes, WORK, grid_divisor);        \
        if (g_ == 0)                                                                           \
            return hipErrorNoDevice;                                                           \
        hipLaunchKernelGGL (FN, dim3 (g_), dim3 (sl.block), sl.lds_bytes, stream, __VA_ARGS__); \
        const hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                                  \
            return e_;                                                                         \
    }
    Real4GLaunch sl {};
    if (! inverse)
    {
        Real4GTwFn ka = nullptr;
        CFFT_R4G_PICK (cols_fwd, plan_c, ka)
        if (ka != nullptr)
            CFFT_R4G_STATIC_LAUNCH (ka, batch * col_tiles, in, work, tw, big, plan_c, (unsigned) n1c, (unsigned) batch, in_stride, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_pass_a_kernel, block_c, slots_c, lds_c, batch * col_tiles, in, work, tw, big, plan_c, (unsigned) n1c, (unsigned) batch, in_stride, lfs_c, ws_len);
        Real4GFn kb = nullptr;
        CFFT_R4G_PICK (rows_fwd, plan_r, kb)
        if (kb != nullptr)
            CFFT_R4G_STATIC_LAUNCH (kb, batch * row_tiles, (const float2*) work, out, tw, plan_r, (unsigned) r2, (unsigned) batch, out_stride, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_pass_b_kernel, block_r, slots_r, lds_r, batch * row_tiles, (const float2*) work, out, tw, plan_r, (unsigned) r2, (unsigned) batch, out_stride, lfs_r, ws_len);
    }
    else
    {
        Real4GTwFn kb = nullptr;
        CFFT_R4G_PICK (rows_inv, plan_r, kb)
        if (kb != nullptr)
            CFFT_R4G_STATIC_LAUNCH (kb, batch * row_tiles, in, work, tw, big, plan_r, (unsigned) r2, (unsigned) batch, in_stride, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_pass_b_inv_kernel, block_r, slots_r, lds_r, batch * row_tiles, in, work, tw, big, plan_r, (unsigned) r2, (unsigned) batch, in_stride, lfs_r, ws_len);
        Real4GFn ka = nullptr;
        CFFT_R4G_PICK (cols_inv, plan_c, ka)
        if (ka != nullptr)
            CFFT_R4G_STATIC_LAUNCH (ka, batch * col_tiles, (const float2*) work, out, tw, plan_c, (unsigned) n1c, (unsigned) batch, out_stride, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_pass_a_inv_kernel, block_c, slots_c, lds_c, batch * col_tiles, (const float2*) work, out, tw, plan_c, (unsigned) n1c, (unsigned) batch, out_stride, lfs_c, ws_len);
    }
#undef CFFT_R4G_PICK
#undef CFFT_R4G_STATIC_LAUNCH
#undef CFFT_R4G_LAUNCH
#undef CFFT_R4G_LAUNCH_B
#undef CFFT_R4G_LAUNCH_BS
    return hipSuccess;
}

// fft_convolve_frames_batch of the same frames: y = BACKWARD (FORWARD (x) * H * scaling), H the packed spectrum of a real filter.
// Three sweeps: columns forward (x -> work), the fused row pass (forward rows, product, backward rows: work -> work), columns backward (work -> y)
hipError_t launch_real4_generic_conv (const DeviceTables& t, const float2* in, const float2* H, float scaling, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    const int r1 = t.real4_r1, r2 = t.real4_r2;
    if (batch > 0x7FFFFFFFull || t.real4_generic == 0 || r1 < 32 || (r1 % 32) != 0 || r2 < 2 || (r2 & 1) != 0 || t.gen_ns < 1 || t.gen2_ns < 1
        || t.gen_ns > kGenericMaxStages || t.gen2_ns > kGenericMaxStages)
        return hipErrorInvalidValue;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const size_t m = (size_t) r1 * (size_t) r2 / 2; // float2 per frame on the caller's side
    in_stride = in_stride ? in_stride : m;
    out_stride = out_stride ? out_stride : m;
    const size_t ws_len = real4_workspace_frame_len (r1, r2);
    const GenericPlan plan_c = make_plan (r2, t.gen_ns, t.gen_radix, t.gen_tw_offset);    // columns
    const GenericPlan plan_r = make_plan (r1, t.gen2_ns, t.gen2_radix, t.gen2_tw_offset); // rows
    unsigned lfs_c = 0, lfs_r = 0;
    const size_t lds_c = tile_lds (r2, lfs_c), lds_r = tile_lds (r1, lfs_r);
    if (lds_c > dev.lds_per_cu || lds_r > dev.lds_per_cu)
        return hipErrorInvalidValue;
    const int block_c = generic_tile_block (lds_c, dev.lds_per_cu), block_r = generic_tile_block (lds_r, dev.lds_per_cu);

    const int n1c = r1 / 2;
    const size_t col_tiles = (size_t) (n1c / kReal4GTile), row_tiles = ((size_t) r2 / 2 + 1 + kReal4GTile - 1) / kReal4GTile;
    const float2* tw = t.stage_twiddles;
    const float2* big = t.big_twiddles;
    // request registers per work-item: the fewest of 4 / 6 / 8 / 10 that hold a tile (fewer registers, more workgroups per CU)
    // (the column kernels also move 16 (r2 / 2 + 1) float4 -- the rows kept -- through SLOTS / 2 registers: 32 float2 more than the tile)
    const auto slots_for = [] (int n, int block, int extra) {
        const int need = (kReal4GTile * n + extra + block - 1) / block;
        return need <= 4 ? 4 : (need <= 6 ? 6 : (need <= 8 ? 8 : (need <= 10 ? 10 : 0)));
    };
    const int slots_c = slots_for (r2, block_c, 32), slots_r = slots_for (r1, block_r, 0);
    if (slots_c == 0 || slots_r == 0)
        return hipErrorInvalidValue;
#define CFFT_R4G_LAUNCH_BS(KERNEL, B_, S_, LDS, WORK, ...)                                                     \
    {                                                                                                          \
        const unsigned g_ = tile_grid (KERNEL<B_, S_>, B_, LDS, WORK, grid_divisor);                           \
        if (g_ == 0)                                                                                           \
            return hipErrorNoDevice;                                                                           \
        hipLaunchKernelGGL ((KERNEL<B_, S_>), dim3 (g_), dim3 (B_), LDS, stream, __VA_ARGS__);                 \
    }
#define CFFT_R4G_LAUNCH_B(KERNEL, B_, SLOTSV, LDS, WORK, ...)                                                  \
    switch (SLOTSV)                                                                                            \
    {                                                                                                          \
        case 4: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 4, LDS, WORK, __VA_ARGS__) break;                              \
        case 6: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 6, LDS, WORK, __VA_ARGS__) break;                              \
        case 8: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 8, LDS, WORK, __VA_ARGS__) break;                              \
        default: CFFT_R4G_LAUNCH_BS (KERNEL, B_, 10, LDS, WORK, __VA_ARGS__) break;                            \
    }
#define CFFT_R4G_LAUNCH(KERNEL, BLOCKV, SLOTSV, LDS, WORK, ...)                                                \
    {                                                                                                          \
        if ((BLOCKV) == 1024)                                                                                  \
            CFFT_R4G_LAUNCH_B (KERNEL, 1024, SLOTSV, LDS, WORK, __VA_ARGS__)                                   \
        else                                                                                                   \
            CFFT_R4G_LAUNCH_B (KERNEL, 256, SLOTSV, LDS, WORK, __VA_ARGS__)                                    \
        const hipError_t e_ = hipGetLastError();                                                               \
        if (e_ != hipSuccess)                                                                                  \
            return e_;                                                                                         \
    }
    // a compile-time stage list for the length (kernels_real4_static_*.hip) brings its own geometry; else the run-time list
    const bool runtime_lists_only = t.real4_generic == 2; // (a plan made under option "real4_generic" = 2)
    const auto static_off = [runtime_lists_only] { static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }(); return off || runtime_lists_only; };
#define CFFT_R4G_PICK(NAME, PLAN, OUT)                                                         \
    if (! static_off())                                                                        \
    {                                                                                          \
        if ((OUT = pick_real4_##NAME##_a (PLAN, dev.lds_per_cu, sl)) == nullptr)               \
            if ((OUT = pick_real4_##NAME##_b (PLAN, dev.lds_per_cu, sl)) == nullptr)           \
                OUT = pick_real4_##NAME##_c (PLAN, dev.lds_per_cu, sl);                        \
    }
#define CFFT_R4G_STATIC_LAUNCH(FN, WORK, ...)                                                  \
    {                                                                                          \
        const unsigned g_ = tile_grid (FN, sl.block, sl.lds_bytes, WORK, grid_divisor);        \
        if (g_ == 0)                                                                           \
            return hipErrorNoDevice;                                                           \
        hipLaunchKernelGGL (FN, dim3 (g_), dim3 (sl.block), sl.lds_bytes, stream, __VA_ARGS__); \
        const hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                                  \
            return e_;                                                                         \
    }
    Real4GLaunch sl {};
    {
        Real4GTwFn ka = nullptr;
        CFFT_R4G_PICK (cols_fwd, plan_c, ka)
        if (ka != nullptr)
            CFFT_R4G_STATIC_LAUNCH (ka, batch * col_tiles, in, work, tw, big, plan_c, (unsigned) n1c, (unsigned) batch, in_stride, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_pass_a_kernel, block_c, slots_c, lds_c, batch * col_tiles, in, work, tw, big, plan_c, (unsigned) n1c, (unsigned) batch, in_stride, lfs_c, ws_len);
        Real4GConvFn kb = nullptr;
        CFFT_R4G_PICK (rows_conv, plan_r, kb)
        if (kb != nullptr)
            CFFT_R4G_STATIC_LAUNCH (kb, batch * row_tiles, work, tw, big, H, scaling, plan_r, (unsigned) r2, (unsigned) batch, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_rows_conv_kernel, block_r, slots_r, lds_r, batch * row_tiles, work, tw, big, H, scaling, plan_r, (unsigned) r2, (unsigned) batch, lfs_r, ws_len);
        Real4GFn kc = nullptr;
        CFFT_R4G_PICK (cols_inv, plan_c, kc)
        if (kc != nullptr)
            CFFT_R4G_STATIC_LAUNCH (kc, batch * col_tiles, (const float2*) work, out, tw, plan_c, (unsigned) n1c, (unsigned) batch, out_stride, sl.lfs, ws_len)
        else
            CFFT_R4G_LAUNCH (real4g_pass_a_inv_kernel, block_c, slots_c, lds_c, batch * col_tiles, (const float2*) work, out, tw, plan_c, (unsigned) n1c, (unsigned) batch, out_stride, lfs_c, ws_len);
    }
#undef CFFT_R4G_PICK
#undef CFFT_R4G_STATIC_LAUNCH
#undef CFFT_R4G_LAUNCH
#undef CFFT_R4G_LAUNCH_B
#undef CFFT_R4G_LAUNCH_BS
    return hipSuccess;
}

// ---- complex frames, general two-pass plans with both factors at most 639 points: the three-sweep fused convolution (conv4_generic_kernels.hpp)
bool conv4_generic_applies (const DeviceTables& t)
{
    return t.mixed_n2 > 0 && t.mixed_cb == kReal4GTile && t.mixed_ca == kReal4GTile && t.mixed_n1 <= 639 && t.mixed_n2 <= 639 && (t.mixed_n1 % kReal4GTile) == 0
           && (t.mixed_n2 % kReal4GTile) == 0 && t.gen_ns >= 1 && t.gen2_ns >= 1 && t.gen_ns <= kGenericMaxStages && t.gen2_ns <= kGenericMaxStages;
}

// y = BACKWARD (FORWARD (x) * H * scaling), H = N complex bins in canonical order; work: n1 * n2 float2 per frame
hipError_t launch_conv4_generic (const DeviceTables& t, const float2* in, const float2* H, float scaling, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || ! conv4_generic_applies (t))
        return hipErrorInvalidValue;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const int n1 = t.mixed_n1, n2 = t.mixed_n2; // rows x columns
    const size_t m = (size_t) n1 * (size_t) n2;
    in_stride = in_stride ? in_stride : m;
    out_stride = out_stride ? out_stride : m;
    const size_t ws_len = m;
    const GenericPlan plan_c = make_plan (n2, t.gen_ns, t.gen_radix, t.gen_tw_offset);    // columns
    const GenericPlan plan_r = make_plan (n1, t.gen2_ns, t.gen2_radix, t.gen2_tw_offset); // rows
    unsigned lfs_c = 0, lfs_r = 0;
    const size_t lds_c = tile_lds (n2, lfs_c), lds_r = tile_lds (n1, lfs_r);
    if (lds_c > dev.lds_per_cu || lds_r > dev.lds_per_cu)
        return hipErrorInvalidValue;
    const int block_c = generic_tile_block (lds_c, dev.lds_per_cu), block_r = generic_tile_block (lds_r, dev.lds_per_cu);
    const auto slots_for = [] (int n, int block) {
        const int need = (kReal4GTile * n + block - 1) / block;
        return need <= 4 ? 4 : (need <= 6 ? 6 : (need <= 8 ? 8 : (need <= 10 ? 10 : 0)));
    };
    const int slots_c = slots_for (n2, block_c), slots_r = slots_for (n1, block_r);
    if (slots_c == 0 || slots_r == 0)
        return hipErrorInvalidValue;
    const size_t col_tiles = (size_t) (n1 / kReal4GTile), row_tiles = (size_t) (n2 / kReal4GTile);
    const float2* tw = t.stage_twiddles;
    const float2* big = t.big_twiddles;
    static const bool lists_off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
#define CFFT_C4G_PICK(NAME, PLAN, OUT)                                                         \
    if (! lists_off)                                                                           \
    {                                                                                          \
        if ((OUT = pick_conv4_##NAME##_a (PLAN, dev.lds_per_cu, sl)) == nullptr)               \
            if ((OUT = pick_conv4_##NAME##_b (PLAN, dev.lds_per_cu, sl)) == nullptr)           \
                OUT = pick_conv4_##NAME##_c (PLAN, dev.lds_per_cu, sl);                        \
    }
#define CFFT_C4G_STATIC_LAUNCH(FN, WORK, ...)                                                  \
    {                                                                                          \
        const unsigned g_ = tile_grid (FN, sl.block, sl.lds_bytes, WORK, grid_divisor);        \
        if (g_ == 0)                                                                           \
            return hipErrorNoDevice;                                                           \
        hipLaunchKernelGGL (FN, dim3 (g_), dim3 (sl.block), sl.lds_bytes, stream, __VA_ARGS__); \
        const hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                                  \
            return e_;                                                                         \
    }
#define CFFT_C4G_RT_BS(KERNEL, B_, S_, LDS, WORK, ...)                                          \
    {                                                                                          \
        const unsigned g_ = tile_grid (KERNEL<B_, S_>, B_, LDS, WORK, grid_divisor);           \
        if (g_ == 0)                                                                           \
            return hipErrorNoDevice;                                                           \
        hipLaunchKernelGGL ((KERNEL<B_, S_>), dim3 (g_), dim3 (B_), LDS, stream, __VA_ARGS__); \
    }
#define CFFT_C4G_RT_B(KERNEL, B_, SLOTSV, LDS, WORK, ...)                                       \
    switch (SLOTSV)                                                                            \
    {                                                                                          \
        case 4: CFFT_C4G_RT_BS (KERNEL, B_, 4, LDS, WORK, __VA_ARGS__) break;                  \
        case 6: CFFT_C4G_RT_BS (KERNEL, B_, 6, LDS, WORK, __VA_ARGS__) break;                  \
        case 8: CFFT_C4G_RT_BS (KERNEL, B_, 8, LDS, WORK, __VA_ARGS__) break;                  \
        default: CFFT_C4G_RT_BS (KERNEL, B_, 10, LDS, WORK, __VA_ARGS__) break;                \
    }
#define CFFT_C4G_RT(KERNEL, BLOCKV, SLOTSV, LDS, WORK, ...)                                     \
    {                                                                                          \
        if ((BLOCKV) == 1024)                                                                  \
            CFFT_C4G_RT_B (KERNEL, 1024, SLOTSV, LDS, WORK, __VA_ARGS__)                       \
        else                                                                                   \
            CFFT_C4G_RT_B (KERNEL, 256, SLOTSV, LDS, WORK, __VA_ARGS__)                        \
        const hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                                  \
            return e_;                                                                         \
    }
    Real4GLaunch sl {};
    Real4GTwFn ka = nullptr;
    CFFT_C4G_PICK (cols_fwd, plan_c, ka)
    if (ka != nullptr)
        CFFT_C4G_STATIC_LAUNCH (ka, batch * col_tiles, in, work, tw, big, plan_c, (unsigned) n1, (unsigned) batch, in_stride, sl.lfs, ws_len)
    else
        CFFT_C4G_RT (c4g_cols_fwd_kernel, block_c, slots_c, lds_c, batch * col_tiles, in, work, tw, big, plan_c, (unsigned) n1, (unsigned) batch, in_stride, lfs_c, ws_len);
    Real4GConvFn kb = nullptr;
    CFFT_C4G_PICK (rows_conv, plan_r, kb)
    if (kb != nullptr)
        CFFT_C4G_STATIC_LAUNCH (kb, batch * row_tiles, work, tw, big, H, scaling, plan_r, (unsigned) n2, (unsigned) batch, sl.lfs, ws_len)
    else
        CFFT_C4G_RT (c4g_rows_conv_kernel, block_r, slots_r, lds_r, batch * row_tiles, work, tw, big, H, scaling, plan_r, (unsigned) n2, (unsigned) batch, lfs_r, ws_len);
    Real4GFn kc = nullptr;
    CFFT_C4G_PICK (cols_inv, plan_c, kc)
    if (kc != nullptr)
        CFFT_C4G_STATIC_LAUNCH (kc, batch * col_tiles, (const float2*) work, out, tw, plan_c, (unsigned) n1, (unsigned) batch, out_stride, sl.lfs, ws_len)
    else
        CFFT_C4G_RT (c4g_cols_inv_kernel, block_c, slots_c, lds_c, batch * col_tiles, (const float2*) work, out, tw, plan_c, (unsigned) n1, (unsigned) batch, out_stride, lfs_c, ws_len);
#undef CFFT_C4G_PICK
#undef CFFT_C4G_STATIC_LAUNCH
#undef CFFT_C4G_RT
#undef CFFT_C4G_RT_B
#undef CFFT_C4G_RT_BS
    return hipSuccess;
}
} // namespace cfft::host
