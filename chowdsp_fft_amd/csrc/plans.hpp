// The compile-time radix plans of the LDS-resident core, one per supported frame length, shared by the
// kernels (template instantiation) and by the host (twiddle-table layout).
//
// Counterpart of the reference's run-time factorisation `decompose` / `cffti1_ps`
// (simd/chowdsp_fft_impl_common.hpp:51-82, 120-160): there N/V is split into radix-4 (+ one radix-2)
// passes; here N is split into at most four radix <= 16 stages so that a whole frame needs at most
// three LDS exchanges.
#pragma once
#include "stockham.hpp"

namespace cfft
{
template <int N>
struct CorePlan;
// clang-format off
template <> struct CorePlan<16>   { using type = Plan<16,   4,   4, 4>;          };
template <> struct CorePlan<32>   { using type = Plan<32,   4,   8, 4>;          };
template <> struct CorePlan<64>   { using type = Plan<64,   8,   8, 8>;          };
#if defined(CFFT_PLAN128_ALT) && CFFT_PLAN128_ALT == 0
template <> struct CorePlan<128>  { using type = Plan<128,  8,  16, 8>;          }; // first version: 4.3 TB/s (8 lanes x 16 points, 102 VGPRs)
#else
template <> struct CorePlan<128>  { using type = Plan<128,  16, 8, 4, 4>;        }; // 16-lane teams x 8 points: 5.5 TB/s
#endif
#if defined(CFFT_PLAN256_ALT) && CFFT_PLAN256_ALT == 0
template <> struct CorePlan<256>  { using type = Plan<256,  16, 16, 16>;         }; // first version: 5.42 TB/s
#else
template <> struct CorePlan<256>  { using type = Plan<256,  32, 8, 8, 4>;        }; // 32-lane teams x 8 points: 5.74 TB/s
#endif
template <> struct CorePlan<512>  { using type = Plan<512,  64,  8, 8, 8>;       };
#if defined(CFFT_PLAN1024_ALT) && CFFT_PLAN1024_ALT == 1
template <> struct CorePlan<1024> { using type = Plan<1024, 128, 8, 8, 4, 4>;    }; // experiment
#else
template <> struct CorePlan<1024> { using type = Plan<1024, 64, 16, 16, 4>;      };
#endif
#if defined(CFFT_PLAN2048_ALT) && CFFT_PLAN2048_ALT == 1
template <> struct CorePlan<2048> { using type = Plan<2048, 256, 8, 8, 8, 4>;    }; // experiment: 8 points per work-item
#else
template <> struct CorePlan<2048> { using type = Plan<2048, 128, 16, 16, 8>;     };
#endif
#if defined(CFFT_PLAN4096_ALT) && CFFT_PLAN4096_ALT == 1
template <> struct CorePlan<4096> { using type = Plan<4096, 512, 8, 8, 8, 8>;    }; // experiment: 8 waves per frame
#else
template <> struct CorePlan<4096> { using type = Plan<4096, 256, 16, 16, 16>;    };
#endif
// 8192 points = 64 KiB: the frame still fits the CU's 160 KiB LDS (one 512-thread workgroup per CU, 68 KiB image), so it
// stays a single pass over HBM instead of the two-pass four-step (2x the traffic)
template <> struct CorePlan<8192> { using type = Plan<8192, 512, 16, 16, 16, 2>; };
// 16384 points = 128 KiB: one 1024-thread workgroup per CU (136 KiB image); 128 VGPRs per work-item, so the twiddles
// are streamed from L2 one stage ahead instead of living in registers (fft_core_streamed, stockham.hpp)
template <> struct CorePlan<16384> { using type = Plan<16384, 1024, 16, 16, 16, 4>; };
static constexpr int kStreamedCoreN = 16384; // plans from this length on use the *_streamed kernels
// clang-format on

// the 65536 = 256 x 256 fast path (multipass_kernels.hpp) is written for this plan, independent of CorePlan<256>
using Plan256 = Plan<256, 16, 16, 16>;

#ifndef CFFT_MAX_CORE_N
#define CFFT_MAX_CORE_N 16384
#endif
static constexpr int kMaxCoreN = CFFT_MAX_CORE_N;
// 32768 points: still one pass over HBM, but through its own kernels (split_plane_kernels.hpp: three radix-32 stages in
// registers, the exchanges one component at a time) and its own twiddle-table layout, not through a CorePlan
static constexpr int kSplitCoreN = 32768;
static constexpr int kMaxFusedN = 4096; // longest core the fused convolution / engine kernels are instantiated for
static constexpr int kMinCoreN = 16;
static constexpr int kBlockThreads = 256;

template <class P>
struct Geometry
{
    // frames per workgroup.  CFFT_FPB_T128: the 128-lane teams (N = 2048) either share a 256-thread workgroup (2) or
    // get a 128-thread workgroup each (1), where a barrier stalls two waves instead of four.
#ifndef CFFT_FPB_T128
#define CFFT_FPB_T128 2
#endif
    static constexpr int FPB = P::T == 128 ? CFFT_FPB_T128 : ((kBlockThreads / P::T) > 0 ? (kBlockThreads / P::T) : 1);
    static constexpr int threads = P::T * FPB;
    // Teams narrower than 16 lanes (N <= 128) would touch less than one 128-byte line per team and instruction:
    // their workgroup instead copies its FPB contiguous frames between HBM and a staging area in LDS with
    // full-width accesses ("coalesced HBM loads of many independent frames at once") and the teams work from there.
    static constexpr bool staged = (P::T < 16);
    static constexpr size_t image_floats2 = (size_t) FPB * LdsFrame<P::N>::floats2;
    static constexpr size_t lds_bytes = (image_floats2 + (staged ? (size_t) FPB * P::N : 0)) * sizeof (float2);
    // The real kernels also touch the mirrored bins M-k, whose 16-lane runs start one element off a 128-byte line:
    // 16-lane teams (M = 128, real N = 256) are staged as well there (CFFT_REAL_STAGED_MAX_T, stockham.hpp).
    static constexpr bool staged_real = (P::T <= CFFT_REAL_STAGED_MAX_T);
    static constexpr size_t lds_bytes_real = (image_floats2 + (staged_real ? (size_t) FPB * P::N : 0)) * sizeof (float2);
};

// Width (adjacent columns / rows) of the tiles the two-pass kernels work on, for a core of length n:
// at least 16 (128 contiguous bytes per global access), at least one frame per team of the 256-thread workgroup,
// and an LDS image small enough for two workgroups per CU.
constexpr int tile_width (int n, int team_threads)
{
    const int fpb = 256 / team_threads;
    // CFFT_TILE1024_WIDTH: 16 columns of a 1024-point core are a 139 KiB image, i.e. one workgroup per CU.  With 256
    // work-items doing the 16 frames in rounds that measured slower than 8 columns x two workgroups (N = 2^20: 1.47 vs
    // 1.67 TB/s); with one 64-lane team per frame (1024 work-items, streamed twiddles) it is faster: 1.61 -> 1.87 TB/s.
#ifndef CFFT_TILE1024_WIDTH
#define CFFT_TILE1024_WIDTH 16
#endif
    // CFFT_TILE_BLOCK_MIN_N: cores from this length on run their tile kernels with one team per tile frame (up to 1024
    // work-items per workgroup) instead of 256 work-items doing the frames in rounds.  Measured on MI355X: N = 2^18
    // (512 x 512) 1.98 -> 2.07 TB/s, 2^19 (1024 x 512) 1.78 -> 1.93, 2^20 unchanged; from 256 on: 2^15 -3 %.
#ifndef CFFT_TILE_BLOCK_MIN_N
#define CFFT_TILE_BLOCK_MIN_N 512
#endif
    // n = 64 (32 teams per workgroup) also takes 16: only the long mixed-radix plans still use a 64-point row pass, and
    // their column length (odd part * 16) is a multiple of 16, not of 32; half of the teams idle there
    return n >= 1024 ? CFFT_TILE1024_WIDTH : (fpb > 16 && n != 64 ? fpb : 16);
}
inline int tile_width_for (int n); // run-time version, defined after core_plan_desc

// Run-time mirror of a Plan, for the host-side table builder.
struct PlanDesc
{
    int N = 0, T = 0, ns = 0;
    int radix[4] = { 1, 1, 1, 1 };
    int tw_total = 0;
};

template <class P>
constexpr PlanDesc describe()
{
    PlanDesc d;
    d.N = P::N;
    d.T = P::T;
    d.ns = P::NS;
    for (int s = 0; s < 4; ++s)
        d.radix[s] = s < P::NS ? P::radix (s) : 1;
    d.tw_total = P::tw_total;
    return d;
}

inline bool core_plan_desc (int n, PlanDesc& d)
{
    switch (n)
    {
        case 16: d = describe<CorePlan<16>::type>(); return true;
        case 32: d = describe<CorePlan<32>::type>(); return true;
        case 64: d = describe<CorePlan<64>::type>(); return true;
        case 128: d = describe<CorePlan<128>::type>(); return true;
        case 256: d = describe<CorePlan<256>::type>(); return true;
        case 512: d = describe<CorePlan<512>::type>(); return true;
        case 1024: d = describe<CorePlan<1024>::type>(); return true;
        case 2048: d = describe<CorePlan<2048>::type>(); return true;
        case 4096: d = describe<CorePlan<4096>::type>(); return true;
        case 8192: d = describe<CorePlan<8192>::type>(); return true;
        case 16384: d = describe<CorePlan<16384>::type>(); return true;
        default: return false;
    }
}
inline int tile_width_for (int n)
{
    PlanDesc d;
    return core_plan_desc (n, d) ? tile_width (d.N, d.T) : 0;
}
} // namespace cfft
