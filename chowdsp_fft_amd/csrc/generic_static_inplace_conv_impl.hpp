// Body of kernels_generic_conv_static_{a,b,c,d}.hip: the fused convolution of mixed-radix frames on the in-place kernel with
// compile-time stage lists for ONE group of lengths (generic_static_inplace.hpp; round 3 had ten of the 52 lengths).  The
// including file defines CFFT_SIP_GROUP and CFFT_SIPC_ENTRY (declared in tile_static.hpp).
#include "generic_conv_kernels.hpp"
#include "generic_static_inplace.hpp"
#include "launch_util.hpp"
#include "tables.hpp"
#include "tile_static.hpp"

namespace cfft::host
{
namespace
{
template <bool REAL, int BLOCK, class SP>
hipError_t launch_sipc (const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream)
{
    auto kernel = generic_inplace_conv_kernel<REAL, BLOCK, true, SP>;
    const size_t lds = ((size_t) plan.n + (size_t) plan.lds_tw_len) * sizeof (float2);
    const unsigned grid = persistent_grid (kernel, BLOCK, lds, batch); // raises the dynamic-LDS cap when lds > 64 KiB
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (BLOCK), lds, stream, in, H, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, scaling, (unsigned) batch);
    return hipGetLastError();
}
} // namespace

// true: the length is in this group, the planner's list is the template's, the launch was made (rc = its result); `plan`
// arrives with lds_stages / lds_tw_len set by the caller (kernels_generic_conv.hip)
bool CFFT_SIPC_ENTRY (bool real, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream, hipError_t& rc)
{
#define CFFT_TRY(N_, B_, ...)                                                                                     \
    if (plan.n == N_)                                                                                             \
    {                                                                                                             \
        using SP = typename StaticInplacePlanFor<N_>::type;                                                       \
        constexpr int CB_ = inplace_conv_block (N_, B_);                                                          \
        bool same = plan.ns == SP::ns && block == CB_;                                                            \
        for (int i = 0; same && i < SP::ns; ++i)                                                                  \
            same = plan.radix[i] == SP::radix (i);                                                                \
        if (! same)                                                                                               \
            return false;                                                                                         \
        rc = real ? launch_sipc<true, CB_, SP> (plan, t, in, H, out, scaling, batch, stream)                      \
                  : launch_sipc<false, CB_, SP> (plan, t, in, H, out, scaling, batch, stream);                    \
        return true;                                                                                              \
    }
    CFFT_SIP_GROUP (CFFT_TRY)
#undef CFFT_TRY
    return false;
}
} // namespace cfft::host
