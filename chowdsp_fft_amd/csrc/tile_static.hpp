// Tile kernels (pass A / pass B of the long mixed-radix plans, generic_kernels.hpp) with a compile-time stage list: chosen by
// kernels_multipass.hip, instantiated in a translation unit of their own (kernels_tile_static.hip) so that the hundred-odd
// kernels build beside the others.
#pragma once
#include "generic_kernels.hpp"

namespace cfft::host
{
using PassAGenericFn = void (*) (const float2*, float2*, const float2*, const float2*, GenericPlan, unsigned, unsigned, size_t, unsigned, size_t);
using PassBGenericFn = void (*) (const float2*, float2*, const float2*, GenericPlan, unsigned, unsigned, unsigned, size_t, unsigned, size_t);
// nullptr: no compile-time list for this (plan, tile width, frame pitch) -- run the run-time list.  Otherwise `block` becomes
// the workgroup size the kernel was instantiated for.
PassAGenericFn pick_static_pass_a (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
PassBGenericFn pick_static_pass_b (const GenericPlan& plan, int c, int& block, unsigned lfs, bool inverse);
} // namespace cfft::host
