// Helpers shared by the kernels_*.hip launch files.
#pragma once
#include "launch.hpp"
#include "plans.hpp"

#include <algorithm>

namespace cfft::host
{
// persistent grid: one workgroup per resident slot, never more than there is work
template <class K>
inline unsigned persistent_grid (K kernel, int threads, size_t lds_bytes, size_t work_items)
{
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
#if defined(CFFT_PERSISTENT) && CFFT_PERSISTENT == 0
    return (unsigned) std::max<size_t> (1, work_items); // experiment: one workgroup per work item
#endif
    const int per_cu = blocks_per_cu_cached ((const void*) kernel, threads, lds_bytes);
    const size_t resident = (size_t) dev.num_cus * (size_t) per_cu;
    return (unsigned) std::max<size_t> (1, std::min (work_items, resident));
}

// run-time n -> compile-time plan dispatch
#define CFFT_DISPATCH_N(n, CALL)                  \
    switch (n)                                    \
    {                                             \
        case 16: { constexpr int N_ = 16; CALL; } break;     \
        case 32: { constexpr int N_ = 32; CALL; } break;     \
        case 64: { constexpr int N_ = 64; CALL; } break;     \
        case 128: { constexpr int N_ = 128; CALL; } break;   \
        case 256: { constexpr int N_ = 256; CALL; } break;   \
        case 512: { constexpr int N_ = 512; CALL; } break;   \
        case 1024: { constexpr int N_ = 1024; CALL; } break; \
        case 2048: { constexpr int N_ = 2048; CALL; } break; \
        case 4096: { constexpr int N_ = 4096; CALL; } break; \
        default: return hipErrorInvalidValue;     \
    }

// the plain transforms also have the 8192-point single-pass plan (kMaxCoreN); the fused kernels stop at kMaxFusedN
#define CFFT_DISPATCH_CORE_N(n, CALL)                        \
    if ((n) == 8192)                                         \
    {                                                        \
        constexpr int N_ = 8192;                             \
        CALL;                                                \
    }                                                        \
    else if ((n) == 16384)                                   \
    {                                                        \
        constexpr int N_ = 16384;                            \
        CALL;                                                \
    }                                                        \
    else                                                     \
        CFFT_DISPATCH_N (n, CALL)
} // namespace cfft::host
