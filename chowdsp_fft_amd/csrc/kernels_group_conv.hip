// Instantiation + launch of the fused convolution of N = 65536 complex frames on the static groups (group_twopass_kernels.hpp:
// group256_conv_kernel).  A translation unit of its own because it is built with the SCALAR complex arithmetic (CFFT_PACKED = 0,
// Makefile): with the packed form, which the two-pass transform kernels next door gain 13 % from, this kernel's three
// transforms per frame next to two resident twiddle sets need 34 registers more than the 256 that two workgroups per CU leave.
#include "launch_util.hpp"
#include "group_twopass_kernels.hpp"
#include "tables.hpp"

namespace cfft::host
{
namespace
{
constexpr size_t kGroupConvLds = (size_t) 16 * TileFrame<256>::stride * sizeof (float2);
}

unsigned group_conv_grid()
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
    const int per_cu_x10 = std::min (group_wg_per_cu_tenths(), 10 * blocks_per_cu_cached ((const void*) group256_conv_kernel<false>, 256, kGroupConvLds));
    return (unsigned) (dev.num_cus * std::max (per_cu_x10, 0) / 10);
}

// fused convolution y = BACKWARD (FORWARD (x) * H * scaling) of N = 65536 complex frames in one launch
hipError_t launch_group_conv (const float2* in, const float2* H, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, float scaling, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch >= 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync (state, 0, sizeof (GroupState), stream); // ranks, mode and flags start from zero in every launch
    if (e != hipSuccess)
        return e;
    const unsigned grid = group_conv_grid();
    if (grid == 0)
        return hipErrorNoDevice;
    int force_solo = 0, reg_polls = 200;
    get_group_options (force_solo, reg_polls);
    hipLaunchKernelGGL (group256_conv_kernel<false>, dim3 (grid), dim3 (256), kGroupConvLds, stream, in, work, out, H, stage_tw, big_tw, (GroupState*) state, host_err, (unsigned) batch, scaling, seq, (unsigned) reg_polls, (unsigned) force_solo);
    return hipGetLastError();
}
} // namespace cfft::host
