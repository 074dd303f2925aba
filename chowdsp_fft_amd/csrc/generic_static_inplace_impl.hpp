// Body of kernels_generic_static_{a,b,c,d}.hip: the in-place mixed-radix kernel with compile-time stage lists for ONE group of
// lengths (generic_static_inplace.hpp).  The including file defines CFFT_SIP_GROUP (the group's list macro) and CFFT_SIP_ENTRY
// (the name of the function it provides; declared in tile_static.hpp next to the tile kernels' pickers).
#include "generic_static_inplace.hpp"
#include "launch_util.hpp"
#include "tables.hpp"
#include "tile_static.hpp"

namespace cfft::host
{
namespace
{
template <int MODE, bool INV, int BLOCK, class SP>
hipError_t launch_sip (const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    auto kernel = generic_inplace_kernel<MODE, INV, BLOCK, true, SP>;
    const size_t lds = ((size_t) plan.n + (size_t) plan.lds_tw_len) * sizeof (float2);
    const unsigned grid = persistent_grid (kernel, BLOCK, lds, batch); // raises the dynamic-LDS cap when lds > 64 KiB
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (BLOCK), lds, stream, in, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, (unsigned) batch);
    return hipGetLastError();
}
template <int N_>
bool sip_matches (const GenericPlan& plan, int block)
{
    using S = StaticInplacePlanFor<N_>;
    using SP = typename S::type;
    bool same = plan.n == N_ && plan.ns == SP::ns && block == S::block;
    for (int i = 0; same && i < SP::ns; ++i)
        same = plan.radix[i] == SP::radix (i);
    return same;
}
} // namespace

// true: the length belongs to this group, the planner's list is the template's and the launch was made (rc = its result).
// `plan` arrives with lds_stages / lds_tw_len set by the caller (kernels_generic.hip: the leading stages' tables in LDS where
// that costs no workgroup per CU, every table from L2 where not).  mode 0 complex (inverse as given), 1 real forward, 2 real backward.
bool CFFT_SIP_ENTRY (int mode, bool inverse, int block, const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, hipError_t& rc)
{
#define CFFT_TRY(N_, B_, ...)                                                                                     \
    if (plan.n == N_)                                                                                             \
    {                                                                                                             \
        if (! sip_matches<N_> (plan, block))                                                                      \
            return false;                                                                                         \
        using SP = typename StaticInplacePlanFor<N_>::type;                                                       \
        rc = mode == 0 ? (inverse ? launch_sip<0, true, B_, SP> (plan, t, in, out, batch, stream)                 \
                                  : launch_sip<0, false, B_, SP> (plan, t, in, out, batch, stream))               \
                       : (mode == 1 ? launch_sip<1, false, B_, SP> (plan, t, in, out, batch, stream)              \
                                    : launch_sip<2, true, B_, SP> (plan, t, in, out, batch, stream));             \
        return true;                                                                                              \
    }
    CFFT_SIP_GROUP (CFFT_TRY)
#undef CFFT_TRY
    return false;
}
} // namespace cfft::host
