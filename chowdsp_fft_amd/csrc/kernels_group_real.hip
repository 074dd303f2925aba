// Instantiation + launch of the one-launch REAL transform of N = 131072 on the static XCD groups (group_real_kernels.hpp).
// A translation unit of its own: it builds in parallel with kernels_multipass.hip, and its arithmetic form (CFFT_PACKED) is
// chosen for this kernel alone (Makefile).
#include "launch_util.hpp"
#include "group_real_kernels.hpp"
#include "tables.hpp"

#include <cstdlib>

namespace cfft::host
{
namespace
{
constexpr size_t kGroupRealLds = (size_t) 16 * TileFrame<256>::stride * sizeof (float2);
}

unsigned group_real_grid (bool inverse)
{
    // every workgroup of the grid must be resident at once: forward launches take either instantiation (with / without the product), so the
    // grid is sized for the one with the LOWER occupancy (register allocation is not monotonic in what a kernel does)
    auto k = inverse ? group256_real_kernel<true, false> : group256_real_kernel<false, true>;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
    if (! inverse)
    {
        const int plain = blocks_per_cu_cached ((const void*) group256_real_kernel<false, false>, 256, group_lds_bytes (kGroupRealLds));
        const int mulh = blocks_per_cu_cached ((const void*) k, 256, group_lds_bytes (kGroupRealLds));
        const int per_cu_x10 = std::min (get_group_handoff() != 0 ? 10 : group_wg_per_cu_tenths(), 10 * std::min (plain, mulh));
        return (unsigned) (dev.num_cus * std::max (per_cu_x10, 0) / 10);
    }
    const int per_cu_x10 = std::min (get_group_handoff() != 0 ? 10 : group_wg_per_cu_tenths(), 10 * blocks_per_cu_cached ((const void*) k, 256, group_lds_bytes (kGroupRealLds)));
    return (unsigned) (dev.num_cus * std::max (per_cu_x10, 0) / 10);
}

// real N = 131072: forward (real frame -> packed spectrum) or backward, one launch; strides in float2 (0 = 65536)
hipError_t launch_group_real (bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, const float2* real_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride, const float2* mul_h, float mul_scale)
{
    if (batch == 0)
        return hipSuccess;
    if (batch >= 0x7FFFFFFFull || (mul_h != nullptr && inverse))
        return hipErrorInvalidValue;
    in_stride = in_stride ? in_stride : 65536;
    out_stride = out_stride ? out_stride : 65536;
    hipError_t e = hipMemsetAsync (state, 0, sizeof (GroupState), stream); // ranks, mode and flags start from zero in every launch
    if (e != hipSuccess)
        return e;
    const unsigned grid = group_real_grid (inverse);
    if (grid == 0)
        return hipErrorNoDevice;
    int force_solo = 0, reg_polls = 200;
    get_group_options (force_solo, reg_polls);
    // where the next frame's operands are requested: 3 (default) a quarter at a time between the steps of pass B, 0 in one burst
    // behind the pass-B loads (CHOWDSP_FFT_GROUP_PREFETCH, as for the complex kernel)
    static const int pf_env = std::getenv ("CHOWDSP_FFT_GROUP_PREFETCH") != nullptr ? std::atoi (std::getenv ("CHOWDSP_FFT_GROUP_PREFETCH")) : -1;
    const unsigned form = (pf_env >= 0 ? pf_env : 3) == 3 ? 8u : 0u;
    // (the product with a packed filter spectrum rides on the FORWARD transform's stores: fft_convolve_frames_batch)
    auto k = inverse ? group256_real_kernel<true, false> : (mul_h != nullptr ? group256_real_kernel<false, true> : group256_real_kernel<false, false>);
    (void) blocks_per_cu_cached ((const void*) k, 256, group_lds_bytes (kGroupRealLds)); // (raises the launched instantiation's dynamic-LDS cap where the listed form asks for 84 KiB)
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (256), group_lds_bytes (kGroupRealLds), stream, in, work, out, stage_tw, big_tw, real_tw, (GroupState*) state, host_err, (unsigned) batch, in_stride, out_stride, seq, (unsigned) reg_polls, (unsigned) force_solo, form, mul_h, mul_scale);
    return hipGetLastError();
}
} // namespace cfft::host
