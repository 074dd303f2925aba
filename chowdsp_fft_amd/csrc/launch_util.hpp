// Helpers shared by the kernels_*.hip launch files.
#pragma once
#include <cstdlib>
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
    size_t resident = (size_t) dev.num_cus * (size_t) per_cu;
    // developer knob (percent of the resident grid; read once): how does a kernel respond to fewer workgroups in flight?
    static const long grid_percent = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_GRID_PERCENT");
        const long v = e != nullptr ? std::atol (e) : 100;
        return (v >= 10 && v <= 100) ? v : 100L;
    }();
    if (grid_percent != 100)
        resident = std::max<size_t> (1, resident * (size_t) grid_percent / 100);
    return (unsigned) std::max<size_t> (1, std::min (work_items, resident));
}

// work-items per workgroup of the generic tile kernels (pass_a_generic / pass_b_generic / column_fft): 1024 where the two LDS
// images leave room for fewer than four workgroups per CU -- with 256 that CU would run 4 .. 12 waves.  CHOWDSP_FFT_GENERIC_TILE_BLOCK
// (256 / 1024) overrides for experiments.
inline int generic_tile_block (size_t lds_bytes, size_t lds_per_cu)
{
    static const int forced = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_TILE_BLOCK");
        const int v = e != nullptr ? std::atoi (e) : 0;
        return (v == 256 || v == 1024) ? v : 0;
    }();
    if (forced != 0)
        return forced;
    // measured (MI355X, complex, fraction of 8 TB/s with 256 / 1024): 46080 0.10 / 0.18, 61440 0.15 / 0.19, 245760 0.15 / 0.18,
    // 655360 0.11 / 0.18, 2^21 0.07 / 0.10 -- but 20736 (two 18 KiB images, four workgroups per CU) 0.16 / 0.12
    return 4 * lds_bytes > lds_per_cu ? 1024 : 256;
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
