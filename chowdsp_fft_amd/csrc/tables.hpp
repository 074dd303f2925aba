// Host-side construction of the twiddle tables a plan needs on the device.
//
// Counterpart of the table section of the reference's common::fft_new_setup
// (simd/chowdsp_fft_impl_common.hpp:195-214).  The reference forms its angles in float before calling
// cosf/sinf (:90-113, :126-151, :201-203); here every angle is formed and evaluated in double and rounded
// to float once, which is at least as accurate and keeps the result within the 1e-6 parity bound.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace cfft::host
{
struct HostTables
{
    int core_n = 0;                     // length of the LDS-resident complex core
    std::vector<float2> stage_twiddles; // Stockham per-stage tables, layout of Plan::tw_offset
    std::vector<float2> real_twiddles;  // W_N^k, k = 0..N/4   (REAL plans)
    std::vector<float2> big_twiddles;   // multi-pass plans, see multipass.hpp
    std::vector<int> generic_radix;     // non-power-of-two plans: stage radices of the generic kernel
    std::vector<int> generic_tw_offset; // ... and the offset of each stage's table inside stage_twiddles
    // long mixed-radix plans (complex length > 20000): two passes, n2-point generic column FFTs (the radices above)
    // then n1-point power-of-two row FFTs whose stage table starts at mixed_tw1_offset inside stage_twiddles
    int mixed_n1 = 0, mixed_n2 = 0, mixed_tw1_offset = 0, mixed_ca = 0; // mixed_ca: columns per pass-A tile (16, 8 or 4)
    // "general" form of the same plans (no power-of-two row length with a compile-time core): the rows run the generic kernel
    // too, with their own stage list; mixed_cb rows per pass-B tile; both tile widths may leave a ragged last tile
    std::vector<int> generic2_radix, generic2_tw_offset;
    int mixed_cb = 0;
    int fast_n1 = 0, fast_n2 = 0;   // complex 2^17 .. 2^20: register-direct passes (tile_fast_kernels.hpp)
    int r2_n = 0;                   // complex 2^21: radix-2 around the r2_n x r2_n register-direct two-pass transform of the two half-length sub-frames
    int real4_r1 = 0, real4_r2 = 0; // long real frames: real four-step N = r1 * r2 (multipass_kernels.hpp)
    int real4_fast = 0;             // ... on the register-direct tiles (tile_fast_kernels.hpp)
    int real4_generic = 0;          // ... long MIXED-RADIX real frames: both cores by the run-time stage lists (real4_generic_kernels.hpp);
                                    //     generic_* = the r2-point column core, generic2_* = the r1-point row core
    // frames beyond 2^21 complex points (split_huge): huge_r x huge_c, both factors transformed by plans of their own;
    // W_N^j = huge_tw_hi[j >> huge_shift] * huge_tw_lo[j & ((1 << huge_shift) - 1)]
    int huge_r = 0, huge_c = 0, huge_shift = 0;
    std::vector<float2> huge_tw_lo, huge_tw_hi;
};

// Validates (N, transform) and fills the tables.  false = unsupported size.
bool build_host_tables (int N, int transform, HostTables& out);

// Column pass of the 2-D transform: stage list (for general_tile_width (rows) frames per workgroup) and tables of a
// rows-point FFT, rows = 2^a 3^b 5^c, 2 <= rows <= kGeneralMaxFactor.  false otherwise.
bool build_column_tables (int rows, HostTables& out);

// Real N = 2^17 .. 2^20: the real four-step split N = r1 * r2 (r1 >= r2, both in 256..1024); false for any other N.
bool split_real4 (int n_real, int& r1, int& r2);
// Long real frames that are not a power of two (complex length beyond the one-pass kernels): N = r1 x r2 for the real four-step with
// generic cores -- r1 a multiple of 32 (tiles of 16 complex = 32 real columns), r2 even, both 2^a 3^b 5^c and at most 639 points (two
// LDS images of 16 frames of r + 1 points).  false where no such split exists (the frame then runs as a complex two-pass transform of
// N / 2 points + the stand-alone pair stage, three sweeps).  Option "real4_generic": 1 (default) only splits whose two factors both
// have a compile-time stage list, 2 every admissible split, 0 none.
bool split_real4_generic (int n_real, int& r1, int& r2);
void set_real4_generic (int mode);
int get_real4_generic();

// Size policy shared with the C API.
bool size_supported (int N, int transform);
// stage radices of the LDS-resident core of length n (0-padded); false if n has no core plan
bool core_radices (int n, int radix[4]);
// launch geometry of the generic mixed-radix kernel for complex length n: frames per workgroup pass and work-items
// per workgroup (shared by the radix choice in tables.cpp and the launcher in kernels_generic.hip)
inline void generic_geometry (int n, int& frames, int& block, bool real_transform = false, bool fused_conv = false)
{
    // Long frames (n > 2048): one frame per workgroup in ONE LDS image with in-place stages (generic_inplace_kernel); the
    // workgroup grows with n so that a work-item never holds more than 20 points of a stage.
    // Short frames: two images, several frames per workgroup so that each of the 256 work-items has butterflies (~2048
    // points in flight; measured: N=480 +4 % with 4 frames, N=640 +86 % with 3).
    if (n > 2048)
    {
    // Round 5 (profiles/r05_generic_lists.txt step 7): 512 work-items from 2880 points on, not from 5184 -- c2880 ... c5120 +5 ... +17 % (c5120 0.60 -> 0.70 of
    // 8 TB/s), real N = 6400 / 10240 +6 / +11 %; below (2160 ... 2592 points) 256 work-items stay 4 ... 18 % faster, and 1024 work-items before 10240 points lose
    // 7 ... 24 % (c8640, c9216, c10240).
#ifndef CFFT_INPLACE_B256_MAX
#define CFFT_INPLACE_B256_MAX 2592
#endif
#ifndef CFFT_INPLACE_CONV_B256_MAX
#define CFFT_INPLACE_CONV_B256_MAX 4800
#endif
#ifndef CFFT_INPLACE_B512_MAX
#define CFFT_INPLACE_B512_MAX 10240
#endif
        // (the fused convolution keeps 256 work-items up to 4800 points: c2880 / c3840 / real N = 7680 lose 5 ... 7 % with 512, 5120 points gain 7 %)
        block = n <= (fused_conv ? CFFT_INPLACE_CONV_B256_MAX : CFFT_INPLACE_B256_MAX) ? 256 : (n <= CFFT_INPLACE_B512_MAX ? 512 : 1024);
        frames = 1;
        return;
    }
    block = 256;
#ifndef CFFT_GENERIC_MULTI_MAX
#define CFFT_GENERIC_MULTI_MAX 1024
#endif
#ifndef CFFT_GENERIC_POINTS
#define CFFT_GENERIC_POINTS 2048
#endif
    // Points in flight per workgroup.  2048 is the best single value; a few lengths of the reference's list and their neighbours
    // do measurably better with another (A/B per length, complex and real, profiles/r02_generic_points_ab.txt -- the stage
    // list does not depend on the choice: every candidate keeps a quarter of the work-items busy at radix 16):
    //   n = 96 (complex 96, real 192): 1536 -> +1.5 % / +8 %;  n = 144: +11 % / +11 %;  n = 160: 0 / +16 %;
    //   complex 48 / 80 / 192 / 432 / 576 / 768 / 800 / 960: 3072 -> +5 / +10 / +4 / +8 / +7 / +10 / +5 / +6 %
    //   (their real counterparts lose 10-20 % with 3072 and keep 2048).  CFFT_GENERIC_POINTS_TUNED = 0: 2048 everywhere.
    // Round 3, after the radix lists were re-ordered and 48 / 80 moved to the staged-I/O form (tools/gpu_r3_points_ab.sh,
    // profiles/r03_generic_points_ab.txt): complex 48 / 80 now do best with 1536 (0.56 -> 0.66 / 0.64); complex 240 / 320 / 400 /
    // 640 / 864 gain 8 / 5 / 2 / 7 / 5 % with 3072; the real transforms keep their choices.
#ifndef CFFT_GENERIC_POINTS_TUNED
#define CFFT_GENERIC_POINTS_TUNED 1
#endif
    int points = CFFT_GENERIC_POINTS;
    if (CFFT_GENERIC_POINTS_TUNED)
    {
        // The fused convolution (two transforms between a group's loads and stores) was measured on its own in round 5 (step 10 of the same file: it had kept
        // the round-2 choices): 1536 points for most cores up to 432 points and complex 768, 2048 elsewhere (3072 no longer wins anywhere) --
        // complex 240 +33 %, 768 +26 %, 48 / 192 / 288 +10 ... 13 %, 960 +12 %; real N = 480 / 576 / 864 +11 ... 18 %, 96 +13 %, 768 +9 %.
        if (fused_conv)
        {
            const bool small = n == 48 || n == 80 || n == 96 || n == 144 || n == 160 || n == 240 || n == 288 || n == 384 || n == 400 || n == 432;
            points = (small || (! real_transform && (n == 192 || n == 768))) ? 1536 : 2048;
        }
        else
        // Round 5 (profiles/r05_generic_lists.txt step 9, every length at 1536 / 2048 / 3072 / 4096 again): the REAL transforms of the 48- and 80-point cores
        // do best with 1536 as well (real N = 96 +9 %, 160 +8 %), those of the 240- and 288-point cores +2 / +4 %; complex 480 gains 3 % with 3072.
        if (n == 96 || n == 144 || n == 160 || (! fused_conv && (n == 48 || n == 80)) || (real_transform && ! fused_conv && (n == 240 || n == 288)))
            points = 1536;
        else if (! real_transform && (n == 192 || n == 432 || n == 576 || n == 768 || n == 800 || n == 960 || (fused_conv && (n == 48 || n == 80))
                                      || (! fused_conv && (n == 240 || n == 320 || n == 400 || n == 480 || n == 640 || n == 864))))
            points = 3072;
    }
    frames = (n <= CFFT_GENERIC_MULTI_MAX) ? (points / n > 1 ? points / n : 1) : 1;
    // Round 5 (profiles/r05_generic_lists.txt step 12): two frames per workgroup for the complex transforms of 1152, 1280 and 1536 points (+9 / +5 / +7 %); the
    // other lengths above 1024 points, every real transform and the fused convolution keep one (1200 / 1440 -7 / -10 %, real N = 2304 ... 3072 -9 ... -22 %).
    if (CFFT_GENERIC_POINTS_TUNED && ! real_transform && ! fused_conv && (n == 1152 || n == 1280 || n == 1536))
        frames = 2;
}
inline bool generic_uses_inplace (int n) { return n > 2048; }
// work-items of the fused convolution's in-place kernel for a length whose transform kernel runs `transform_block` (generated lists: X (n, work-items, ...))
constexpr int inplace_conv_block (int n, int transform_block) { return (n > 2048 && n <= CFFT_INPLACE_CONV_B256_MAX) ? 256 : transform_block; }

// stage radices of the generic mixed-radix kernel for a complex length that is not a power of two
void generic_plan_radices (int n_complex, std::vector<int>& radix);
// For multi-pass sizes: N = n1 * n2 with both factors LDS-resident; (0,0) if N is LDS-resident itself.
void split_large (int n_complex, int& n1, int& n2);
// Long mixed-radix lengths (2^a 3^b 5^c, not a power of two, > kGenericMaxComplexLen): n = n1 * n2 with n2 = (odd part)
// * 2^j the generic column length (a multiple of 16, at most kMixedMaxN2) and n1 a power of two in [64, 1024].
// false if no such split exists (then the size is not supported).
bool split_mixed (int n_complex, int& n1, int& n2);
// The general split for every other accepted length up to kMaxComplexLen (2^a 3^b 5^c, powers of two beyond 2^20
// included): n = n1 * n2 with both factors 2-3-5-smooth and as close to sqrt(n) as possible; columns AND rows run the
// generic kernel.  false only if a factor would exceed kGeneralMaxFactor (never below 2^21).
bool split_general (int n_complex, int& n1, int& n2);
// Column / row lengths for which the tile kernels have a compile-time stage list (generic_kernels.hpp: StaticPlanFor holds the
// lists): every 2-3-5-smooth length from 144 to 1200 and the shorter multiples of 16 (shorter lengths never come out of a split) -- so every accepted length up to about
// 1.4 million points has a factorisation into two of them.  The splits below prefer those: such a pass runs 1.4-1.7x faster.
// Four groups, instantiated by kernels_tile_static_a/b/c/d.hip (four translation units build side by side).
#define CFFT_STATIC_TILE_SIZES_A(X) \
    X (48) X (80) X (96) X (144) X (160) X (192) X (240) X (288) X (320) X (384) X (400) X (432) X (480) X (576) X (640) X (720) \
    X (768) X (800) X (864) X (960) X (1152) X (1200)
#define CFFT_STATIC_TILE_SIZES_B(X) \
    X (150) X (162) X (180) X (200) X (216) X (225) X (243) X (250) X (256) X (270) \
    X (300) X (324) X (360) X (375)
#define CFFT_STATIC_TILE_SIZES_C(X) \
    X (405) X (450) X (486) X (500) X (512) X (540) X (600) X (625) X (648) X (675) X (729) X (750) X (810) X (900) X (972) X (1000) \
    X (1024) X (1080) X (1125)
// ... and the lengths beyond 1200 that have a list (8-column tiles, one image: kTileInplaceMinN below)
#define CFFT_STATIC_TILE_SIZES_D(X) X (1280) X (1296) X (1440) X (1536) X (1600) X (1728) X (1920) X (2000) X (2048)
#define CFFT_STATIC_TILE_SIZES(X) CFFT_STATIC_TILE_SIZES_A (X) CFFT_STATIC_TILE_SIZES_B (X) CFFT_STATIC_TILE_SIZES_C (X) CFFT_STATIC_TILE_SIZES_D (X)
bool has_static_tile (int n);
// Static tile lengths from kTileInplaceMinN on run their stages in ONE LDS image (generic_kernels.hpp: tile_inplace_stage): tiles of
// 16 columns / rows up to 1200 points, where two images allow 8 beyond 639.  Measured on MI355X (tools/gpu_r3_tile_inplace_ab.sh,
// fraction of 8 TB/s, 8-column two-image / 16-column one-image): 2-D 720 x 1280 0.245 / 0.282, 1080 x 1920 0.250 / 0.297,
// c1327104 0.210 / 0.222, c655360 0.238 / 0.244.  For n = 324 .. 639, where one image would let two or three 512-work-item
// workgroups share a CU instead of one of 1024, the two-image form stays ahead (c230400 0.286 against 0.269, c192000 0.275 / 0.255,
// 2-D 512 x 512 0.310 / 0.284): the threshold is 640.  false for every length when CHOWDSP_FFT_TILE_STATIC = 0 (then the run-time
// kernels and their widths apply).
#ifndef CFFT_TILE_INPLACE_MIN_N
#define CFFT_TILE_INPLACE_MIN_N 640
#endif
static constexpr int kTileInplaceMinN = CFFT_TILE_INPLACE_MIN_N;
bool tile_runs_inplace (int n);
// n = n1 * n2 with both factors static tile lengths of at most max_factor (which pair and which of them is the column length
// n2: see tables.cpp)
bool split_static_pair (int n_complex, int max_factor, int& n1, int& n2);
static constexpr int kGeneralMaxFactor = 2559; // two LDS images of 4 frames of n + 1 points
static constexpr int kMaxComplexLen = 1 << 21;
// Frames longer than that (the reference has no upper bound, simd/chowdsp_fft_impl_common.hpp:168-177; its header advises a heap
// `work` buffer for large N, chowdsp_fft.h:131-134): n = r * c with BOTH factors accepted complex lengths of at most
// kMaxComplexLen, as close to each other as the divisors allow; the frame is transformed as c batched r-point transforms, a
// twiddle, r batched c-point transforms and three transposes (capi.cpp: huge_transform).  Up to 2^27 complex points (1 GiB per
// frame).  false if no such pair of factors exists (then the size is not supported).
static constexpr int kMaxHugeComplexLen = 1 << 27;
bool split_huge (int n_complex, int& r, int& c);
// ... as columns x rows (round 4): r = the largest column length the 2-D column kernel takes (a multiple of 16, at most 2048) that divides n_complex
// with c = n_complex / r an accepted complex length of at most 2^21 points that is not huge itself.  false if there is none.
bool split_huge_columns (int n_complex, int& r, int& c);
// ... except the lengths up to 2^22 that are the product of two static tile lengths (at most 2048 points each, tables below): those
// stay two-pass plans of the general form (2^22 = 2048 x 2048, 2304000 = 2048 x 1125: two sweeps instead of five)
static constexpr int kMaxStaticTwoPassLen = 1 << 22;
bool runs_as_huge (int n_complex);
// tile width (columns or rows) for a frame length n of the generic passes: as wide as two padded LDS images allow
static constexpr int kTileInplaceWideMaxN = 1200; // one image of 16 frames of n + 1 points and the n - 1 stage twiddles within 160 KiB
inline int general_tile_width (int n) { return tile_runs_inplace (n) ? (n <= kTileInplaceWideMaxN ? 16 : 8) : (n <= 639 ? 16 : (n <= 1279 ? 8 : 4)); }
// columns per pass-A tile of the mixed two-pass plans: 16 (128-byte rows) while two LDS images of 16 padded columns,
// 2 * 16 * (n2 + 1) * 8 bytes, fit the CU's 160 KiB (n2 <= 639), else 8 (n2 <= 1280; n2 = 1280 runs unpadded)
static constexpr int kMixedMaxN2Wide = 639;
static constexpr int kMixedMaxN2 = 1280;
inline int mixed_tile_width (int n2) { return (n2 <= kMixedMaxN2Wide || (tile_runs_inplace (n2) && n2 <= kTileInplaceWideMaxN)) ? 16 : 8; }
} // namespace cfft::host
