// The real four-step's tile kernels (real4_generic_kernels.hpp) with compile-time stage lists: chosen by kernels_real4_generic.hip,
// instantiated in translation units of their own (kernels_real4_static_{a,b,c}.hip, one per group of tables.hpp's lengths).
#pragma once
#include "conv4_generic_kernels.hpp"
#include "host_common.hpp"

namespace cfft::host
{
// pass A forward / pass B backward (they multiply by the inter-pass twiddles) and pass B forward / pass A backward
using Real4GTwFn = void (*) (const float2*, float2*, const float2*, const float2*, GenericPlan, unsigned, unsigned, size_t, unsigned, size_t);
using Real4GFn = void (*) (const float2*, float2*, const float2*, GenericPlan, unsigned, unsigned, size_t, unsigned, size_t);
using Real4GConvFn = void (*) (float2*, const float2*, const float2*, const float2*, float, GenericPlan, unsigned, unsigned, unsigned, size_t); // the fused row pass of the convolution
struct Real4GLaunch
{
    int block;        // work-items per workgroup
    unsigned lfs;     // frame pitch inside the LDS images
    size_t lds_bytes; // two images + the stage twiddles' copy
};
// nullptr: no compile-time list for this plan in the group (or it differs from the planner's list)
#define CFFT_R4S_DECLARE(G)                                                                                        \
    Real4GTwFn pick_real4_cols_fwd_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);         \
    Real4GFn pick_real4_cols_inv_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);           \
    Real4GFn pick_real4_rows_fwd_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);           \
    Real4GTwFn pick_real4_rows_inv_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);           \
    Real4GConvFn pick_real4_rows_conv_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);         \
    /* ... and the complex three-sweep fused convolution (conv4_generic_kernels.hpp) */                            \
    Real4GTwFn pick_conv4_cols_fwd_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);         \
    Real4GConvFn pick_conv4_rows_conv_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);      \
    Real4GFn pick_conv4_cols_inv_##G (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch);
CFFT_R4S_DECLARE (a)
CFFT_R4S_DECLARE (b)
CFFT_R4S_DECLARE (c)
#undef CFFT_R4S_DECLARE
} // namespace cfft::host
