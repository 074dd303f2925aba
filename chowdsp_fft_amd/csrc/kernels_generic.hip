// Instantiation + launch of the generic mixed-radix kernel (frames that are not powers of two).
#include "generic_kernels.hpp"
#include "launch_util.hpp"

namespace cfft::host
{
namespace
{
template <int MODE, bool INV>
hipError_t launch_generic_m (const GenericPlan& plan, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    auto kernel = generic_kernel<MODE, INV>;
    const size_t lds = (size_t) 2 * plan.n * sizeof (float2);
    if (lds > 64 * 1024)
    {
        const hipError_t e = hipFuncSetAttribute ((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess)
            return e;
    }
    const unsigned grid = persistent_grid (kernel, 256, lds, batch);
    if (grid == 0)
        return hipErrorNoDevice;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (256), lds, stream, in, out, (const float2*) t.stage_twiddles, (const float2*) t.real_twiddles, plan, (unsigned) batch);
    return hipGetLastError();
}
} // namespace

hipError_t launch_generic (int mode, bool inverse, int nc, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || nc > kGenericMaxN || t.gen_ns < 1 || t.gen_ns > kGenericMaxStages)
        return hipErrorInvalidValue;
    GenericPlan plan {};
    plan.n = nc;
    plan.ns = t.gen_ns;
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
} // namespace cfft::host
