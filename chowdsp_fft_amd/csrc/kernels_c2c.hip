// Instantiation + launch of the batched complex-to-complex kernels (gfx950).
#include "c2c_kernels.hpp"
#include "launch.hpp"
#include "plans.hpp"

#include <algorithm>

namespace cfft::host
{
namespace
{
// Workgroups resident per CU for a kernel, queried once per (kernel, device).
template <class K>
int resident_blocks_per_cu (K kernel, int threads, size_t lds_bytes)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, kernel, threads, lds_bytes) != hipSuccess || n < 1)
        n = 1;
    return n;
}

template <int N, bool INV>
hipError_t launch_c2c_n (const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream)
{
    using P = typename CorePlan<N>::type;
    using G = Geometry<P>;
    auto kernel = c2c_kernel<P, G::FPB, INV>;

    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    static int per_cu = resident_blocks_per_cu (kernel, G::threads, G::lds_bytes);

    const size_t ngroups = (batch + G::FPB - 1) / G::FPB;
    const size_t resident = (size_t) dev.num_cus * (size_t) per_cu;
    const unsigned grid = (unsigned) std::max<size_t> (1, std::min (ngroups, resident));
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (G::threads), G::lds_bytes, stream, in, out, tw, (unsigned) batch);
    return hipGetLastError();
}

template <bool INV>
hipError_t dispatch_c2c (int n, const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream)
{
    switch (n)
    {
        case 16: return launch_c2c_n<16, INV> (in, out, tw, batch, stream);
        case 32: return launch_c2c_n<32, INV> (in, out, tw, batch, stream);
        case 64: return launch_c2c_n<64, INV> (in, out, tw, batch, stream);
        case 128: return launch_c2c_n<128, INV> (in, out, tw, batch, stream);
        case 256: return launch_c2c_n<256, INV> (in, out, tw, batch, stream);
        case 512: return launch_c2c_n<512, INV> (in, out, tw, batch, stream);
        case 1024: return launch_c2c_n<1024, INV> (in, out, tw, batch, stream);
        case 2048: return launch_c2c_n<2048, INV> (in, out, tw, batch, stream);
        case 4096: return launch_c2c_n<4096, INV> (in, out, tw, batch, stream);
        default: return hipErrorInvalidValue;
    }
}
} // namespace

hipError_t launch_c2c (int n, bool inverse, const float2* in, float2* out, const float2* stage_tw, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0xFFFFFFFFull)
        return hipErrorInvalidValue;
    return inverse ? dispatch_c2c<true> (n, in, out, stage_tw, batch, stream)
                   : dispatch_c2c<false> (n, in, out, stage_tw, batch, stream);
}
} // namespace cfft::host
