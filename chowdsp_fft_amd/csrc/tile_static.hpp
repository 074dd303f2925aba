// Tile kernels (pass A / pass B of the long mixed-radix plans, generic_kernels.hpp) with a compile-time stage list: chosen by
// kernels_multipass.hip, instantiated in three translation units of their own (kernels_tile_static_{a,b,c}.hip, one per group of
// lengths) so that the 240-odd kernels build beside the others.
#pragma once
#include "generic_kernels.hpp"

#include <cstdlib>

namespace cfft::host
{
using PassAGenericFn = void (*) (const float2*, float2*, const float2*, const float2*, GenericPlan, unsigned, unsigned, size_t, unsigned, size_t);
using PassBGenericFn = void (*) (const float2*, float2*, const float2*, GenericPlan, unsigned, unsigned, unsigned, size_t, unsigned, size_t);
// nullptr: no compile-time list for this (plan, tile width, frame pitch) in the group.  Otherwise `block` becomes the workgroup
// size the kernel was instantiated for.
PassAGenericFn pick_static_pass_a_a (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
PassAGenericFn pick_static_pass_a_b (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
PassAGenericFn pick_static_pass_a_c (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
PassBGenericFn pick_static_pass_b_a (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
PassBGenericFn pick_static_pass_b_b (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
PassBGenericFn pick_static_pass_b_c (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
// CHOWDSP_FFT_TILE_STATIC = 0: the run-time list everywhere (A/B)
inline bool static_tiles_off()
{
    static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
    return off;
}
inline PassAGenericFn pick_static_pass_a (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse)
{
    if (static_tiles_off())
        return nullptr;
    if (auto k = pick_static_pass_a_a (plan, c, block, lfs, inverse))
        return k;
    if (auto k = pick_static_pass_a_b (plan, c, block, lfs, inverse))
        return k;
    return pick_static_pass_a_c (plan, c, block, lfs, inverse);
}
inline PassBGenericFn pick_static_pass_b (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse)
{
    if (static_tiles_off())
        return nullptr;
    if (auto k = pick_static_pass_b_a (plan, c, block, lfs, inverse))
        return k;
    if (auto k = pick_static_pass_b_b (plan, c, block, lfs, inverse))
        return k;
    return pick_static_pass_b_c (plan, c, block, lfs, inverse);
}
} // namespace cfft::host
