// Tile kernels (pass A / pass B of the long mixed-radix plans, generic_kernels.hpp) with a compile-time stage list: chosen by
// kernels_multipass.hip, instantiated in four translation units of their own (kernels_tile_static_{a,b,c,d}.hip, one per group of
// lengths) so that the 380-odd kernels build beside the others.
#pragma once
#include "generic_kernels.hpp"
#include "host_common.hpp"

#include <cstdlib>

namespace cfft::host
{
using PassAGenericFn = void (*) (const float2*, float2*, const float2*, const float2*, GenericPlan, unsigned, unsigned, size_t, unsigned, size_t);
using PassBGenericFn = void (*) (const float2*, float2*, const float2*, GenericPlan, unsigned, unsigned, unsigned, size_t, unsigned, size_t);
using ColumnFftFn = void (*) (float2*, const float2*, GenericPlan, unsigned, unsigned, size_t, unsigned, const float2*, const float2*, const float2*, unsigned); // the 2-D transforms' column pass
// nullptr: no compile-time list for this (plan, tile width) in the group, or its LDS does not fit the device.  Otherwise `launch`
// holds what the kernel was instantiated for.
struct StaticTileLaunch
{
    int block;        // work-items per workgroup
    unsigned lfs;     // frame pitch inside the LDS image(s)
    size_t lds_bytes; // dynamic LDS: the image(s) and the stage twiddles' copy
};
PassAGenericFn pick_static_pass_a_a (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassAGenericFn pick_static_pass_a_b (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassAGenericFn pick_static_pass_a_c (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassAGenericFn pick_static_pass_a_d (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassBGenericFn pick_static_pass_b_a (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassBGenericFn pick_static_pass_b_b (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassBGenericFn pick_static_pass_b_c (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
PassBGenericFn pick_static_pass_b_d (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
ColumnFftFn pick_static_column_a (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
ColumnFftFn pick_static_column_b (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
ColumnFftFn pick_static_column_c (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
ColumnFftFn pick_static_column_d (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch);
// CHOWDSP_FFT_TILE_STATIC = 0: the run-time list everywhere (A/B)
inline bool static_tiles_off()
{
    static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
    return off;
}
inline PassAGenericFn pick_static_pass_a (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch)
{
    if (static_tiles_off())
        return nullptr;
    if (auto k = pick_static_pass_a_a (plan, c, lds_per_cu, inverse, launch))
        return k;
    if (auto k = pick_static_pass_a_b (plan, c, lds_per_cu, inverse, launch))
        return k;
    if (auto k = pick_static_pass_a_c (plan, c, lds_per_cu, inverse, launch))
        return k;
    return pick_static_pass_a_d (plan, c, lds_per_cu, inverse, launch);
}
inline PassBGenericFn pick_static_pass_b (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch)
{
    if (static_tiles_off())
        return nullptr;
    if (auto k = pick_static_pass_b_a (plan, c, lds_per_cu, inverse, launch))
        return k;
    if (auto k = pick_static_pass_b_b (plan, c, lds_per_cu, inverse, launch))
        return k;
    if (auto k = pick_static_pass_b_c (plan, c, lds_per_cu, inverse, launch))
        return k;
    return pick_static_pass_b_d (plan, c, lds_per_cu, inverse, launch);
}
inline ColumnFftFn pick_static_column (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch)
{
    if (static_tiles_off())
        return nullptr;
    if (auto k = pick_static_column_a (plan, c, lds_per_cu, inverse, launch))
        return k;
    if (auto k = pick_static_column_b (plan, c, lds_per_cu, inverse, launch))
        return k;
    if (auto k = pick_static_column_c (plan, c, lds_per_cu, inverse, launch))
        return k;
    return pick_static_column_d (plan, c, lds_per_cu, inverse, launch);
}
// The in-place mixed-radix kernel with compile-time stage lists (generic_static_inplace.hpp, kernels_generic_static_{a,b,c,d}.hip):
// true = the length is in the group, the list is the planner's, the launch was made (rc = its result)
bool launch_static_inplace_a (int mode, bool inverse, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, hipError_t& rc);
bool launch_static_inplace_b (int mode, bool inverse, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, hipError_t& rc);
bool launch_static_inplace_c (int mode, bool inverse, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, hipError_t& rc);
bool launch_static_inplace_d (int mode, bool inverse, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, hipError_t& rc);

// ... and the fused convolution on the same lists (kernels_generic_conv_static_{a,b,c,d}.hip)
bool launch_static_inplace_conv_a (bool real, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream, hipError_t& rc);
bool launch_static_inplace_conv_b (bool real, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream, hipError_t& rc);
bool launch_static_inplace_conv_c (bool real, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream, hipError_t& rc);
bool launch_static_inplace_conv_d (bool real, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream, hipError_t& rc);

// PIPE variants of the tile kernels need an LDS copy of the n - 1 stage twiddles behind the two images: granted where that leaves
// the workgroups per CU as they are (by LDS or by the CU's 32 waves, whichever binds).  CHOWDSP_FFT_TILE_PIPE (experiments): bit 0 =
// pass A / the 2-D column pass, bit 1 = pass B; a set bit grants it wherever the copy fits at all, a clear one withholds it.
inline bool tile_pipe_fits (size_t images_bytes, size_t n, size_t lds_per_cu, int which, int block)
{
    static const int forced = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_TILE_PIPE");
        return e != nullptr ? std::atoi (e) : -1;
    }();
    const size_t with_tw = images_bytes + (n - 1) * sizeof (float2);
    if ((forced >= 0 && (forced & which) == 0) || with_tw > lds_per_cu)
        return false;
    const size_t by_waves = (size_t) 2048 / (size_t) block; // 32 waves per CU
    const size_t a = lds_per_cu / with_tw, b = lds_per_cu / images_bytes;
    return forced >= 0 || (a < by_waves ? a : by_waves) == (b < by_waves ? b : by_waves);
}
} // namespace cfft::host
