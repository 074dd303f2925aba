// Twiddle-table construction (host, double precision, rounded once to fp32).
#include "tables.hpp"

#include <atomic>
#include <algorithm>
#include <functional>
#include "plans.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace cfft::host
{
static void generic_radices (int n, std::vector<int>& radix, int frames_override);
namespace
{
bool is_pow2 (int n) { return n > 0 && (n & (n - 1)) == 0; }

float2 unit_root (long long num, long long den) // exp(-2*pi*i*num/den)
{
    const long long r = ((num % den) + den) % den;
    const double a = -2.0 * M_PI * (double) r / (double) den;
    return make_float2 ((float) std::cos (a), (float) std::sin (a));
}

// split-plane single pass (split_plane_kernels.hpp), N = 32 T: per twiddled stage rows [j][k], k < p, of W_{r p}^{a k}:
// stage 1 (p = 32, r = 32): a = 1..7, 8, 16, 24; stage 2 (p = 1024): the same ten rows for r = 32 (N = 32768), or
// a = 1, 2, 3, 4, 8, 12 for r = 16 (N = 16384), or a = 1..7 for r = 8 (N = 8192)
void append_split_plane_twiddles (int n, std::vector<float2>& out)
{
    static const int a32[10] = { 1, 2, 3, 4, 5, 6, 7, 8, 16, 24 };
    static const int a16[6] = { 1, 2, 3, 4, 8, 12 };
    for (int j = 0; j < 10; ++j)
        for (int k = 0; k < 32; ++k)
            out.push_back (unit_root ((long long) a32[j] * k, 1024));
    const int last = n / 1024;
    for (int j = 0; j < (last == 32 ? 10 : (last == 16 ? 6 : last - 1)); ++j) // last == 8 / 2 (N = 8192 / 2048): a = 1 .. last - 1, the head of a32
        for (int k = 0; k < 1024; ++k)
            out.push_back (unit_root ((long long) (last == 16 ? a16[j] : a32[j]) * k, n));
}

void append_stage_twiddles (const PlanDesc& d, std::vector<float2>& out)
{
    // stage s >= 1 : (r-1) * p entries, [a-1][k] = W_{p*r}^{a*k}
    int p = d.radix[0];
    for (int s = 1; s < d.ns; ++s)
    {
        const int r = d.radix[s];
        for (int a = 1; a < r; ++a)
            for (int k = 0; k < p; ++k)
                out.push_back (unit_root ((long long) a * k, (long long) p * r));
        p *= r;
    }
}
} // namespace

void generic_plan_radices (int n_complex, std::vector<int>& radix) { generic_radices (n_complex, radix, 0); }

bool core_radices (int n, int radix[4])
{
    PlanDesc d;
    if (! core_plan_desc (n, d))
        return false;
    for (int i = 0; i < 4; ++i)
        radix[i] = i < d.ns ? d.radix[i] : 0;
    return true;
}

void split_large (int n_complex, int& n1, int& n2)
{
    n1 = n2 = 0;
    if (n_complex <= kMaxCoreN || n_complex == kSplitCoreN)
        return;
    int lg = 0;
    while ((1 << lg) < n_complex)
        ++lg;
    const int l2 = lg / 2; // pass A (column) length 2^l2, pass B (row) length 2^(lg-l2)
    n2 = 1 << l2;
    n1 = 1 << (lg - l2);
}

bool has_static_tile (int n)
{
#define CFFT_IS(N_) if (n == N_) return true;
    CFFT_STATIC_TILE_SIZES (CFFT_IS)
#undef CFFT_IS
    return false;
}

bool tile_runs_inplace (int n)
{
    static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
    return ! off && n >= kTileInplaceMinN && has_static_tile (n);
}

bool split_static_pair (int n_complex, int max_factor, int& n1, int& n2)
{
    // Among the factorisations into two static tile lengths of at most max_factor points: the most balanced one in which a factor
    // is a multiple of 16 (and the other at most five times as long) -- both, if the length allows --, else the most balanced one.  The COLUMN
    // length n2 is the pitch of pass B's 16-point store runs: the smaller factor, unless that is not a multiple of 8 while the
    // other has more factors of two (100000 = 250 x 400, not 400 x 250; but 48000 = 240 x 200: 0.300 against 200 x 240 0.285).  Measured on MI355X (tools/gpu_r3_general_split_ab.sh, n1 x n2, fraction
    // of 8 TB/s): 34992 = 216 x 162 0.222, 162 x 216 0.253, 243 x 144 0.267; 100000 = 400 x 250 0.223, 250 x 400 0.252;
    // 30000 = 200 x 150 0.236, 150 x 200 0.259, 375 x 80 0.265 -- but 625 x 48 0.209 and 50000 = 625 x 80 0.210 against
    // 250 x 200 0.257 (too lopsided).
    n1 = n2 = 0;
    int best_d[3] = { 0, 0, 0 }; // by the number of factors that are multiples of 16; within a class the largest d (closest to sqrt(n))
    for (int d = 1; (long long) d * d <= (long long) n_complex; ++d)
    {
        if (n_complex % d != 0)
            continue;
        const int e = n_complex / d;
        if (e > max_factor || ! has_static_tile (d) || ! has_static_tile (e))
            continue;
        const int aligned = (d % 16 == 0 ? 1 : 0) + (e % 16 == 0 ? 1 : 0);
        best_d[(aligned > 0 && (long long) e > 5ll * d) ? 0 : aligned] = d;
    }
    const int d = best_d[2] != 0 ? best_d[2] : (best_d[1] != 0 ? best_d[1] : best_d[0]);
    if (d == 0)
        return false;
    const int e = n_complex / d;
    const auto twos = [] (int v) { int c = 0; while (v % 2 == 0) { v /= 2; ++c; } return c; };
    // d <= e: the smaller factor is the column length unless it is not even a multiple of 8 and the other has more factors of two
    const bool d_is_column = twos (d) >= 3 || twos (d) >= twos (e);
    n2 = d_is_column ? d : e;
    n1 = d_is_column ? e : d;
    return true;
}

bool split_mixed (int n_complex, int& n1, int& n2)
{
    n1 = n2 = 0;
    if (n_complex <= 0 || is_pow2 (n_complex))
        return false;
    int odd = n_complex, e = 0;
    while (odd % 2 == 0) { odd /= 2; ++e; }
    for (int j = 4; j <= e; ++j)
    {
        const long long c2 = (long long) odd << j;
        if (c2 > kMixedMaxN2)
            break;
        const int r1 = 1 << (e - j);
        // pass B works on tiles of tile_width_for (n1) adjacent rows (32 for n1 = 64): n2 must be a whole number of them
        if (r1 >= 64 && r1 <= 1024 && c2 % tile_width_for (r1) == 0)
        {
            // columns longer than 639 points run in tiles of 8 (64-byte pieces of the caller's rows): where the length is also
            // the product of two static tile lengths of at most 639 points, the general form (16-column tiles, compile-time
            // lists on both sides) is faster -- 46080 = 64 x 720: 0.26 of 8 TB/s, = 240 x 192: see profiles/r03_general_two_pass.txt
            int s1 = 0, s2 = 0;
            if (c2 > kMixedMaxN2Wide && split_static_pair (n_complex, kMixedMaxN2Wide, s1, s2))
                return false;
            // experiment (CHOWDSP_FFT_MIXED_TO_GENERAL = 1): the general form wherever two static lengths exist
            static const bool to_general = [] { const char* e = std::getenv ("CHOWDSP_FFT_MIXED_TO_GENERAL"); return e != nullptr && std::atoi (e) == 1; }();
            if (to_general && split_static_pair (n_complex, 2048, s1, s2))
                return false;
            n1 = r1;
            n2 = (int) c2;
            return true;
        }
    }
    return false;
}

bool split_real4 (int n_real, int& r1, int& r2)
{
    r1 = r2 = 0;
    if (n_real < (1 << 17) || n_real > (1 << 20) || ! is_pow2 (n_real))
        return false;
    int lg = 0;
    while ((1 << lg) < n_real)
        ++lg;
    r2 = 1 << (lg / 2);        // column length
    r1 = 1 << (lg - lg / 2);   // row length
    return true;
}

// option "real4_generic" (initial value: CHOWDSP_FFT_REAL4_GENERIC), read when a plan is MADE (the device-table cache keeps plans of the
// different kinds apart, host_common.cpp): 1 (default) = where BOTH factors have a compile-time stage list (kernels_real4_static_*.hip:
// 0.25-0.30 of 8 TB/s against 0.21-0.24 for the three sweeps, profiles/r05_real4_generic.txt); 2 = for every admissible split, run-time
// lists ONLY (0.17-0.18: slower than the three sweeps -- it keeps the run-time kernels tested and measurable); 0 = off.
static std::atomic<int> g_real4_generic { -1 };
void set_real4_generic (int mode) { g_real4_generic = (mode >= 0 && mode <= 2) ? mode : 1; }
int get_real4_generic()
{
    int v = g_real4_generic.load();
    if (v < 0)
    {
        const char* e = std::getenv ("CHOWDSP_FFT_REAL4_GENERIC");
        v = (e != nullptr && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
        g_real4_generic = v;
    }
    return v;
}

bool split_real4_generic (int n_real, int& r1, int& r2)
{
    r1 = r2 = 0;
    const int mode = get_real4_generic();
    const bool off = mode == 0;
    static const bool no_lists = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
    constexpr int kOnePassMaxComplex = 20000; // (= kGenericMaxComplexLen below: shorter frames stay on chip)
    if (off || n_real <= 2 * kOnePassMaxComplex || is_pow2 (n_real) || (n_real % 64) != 0)
        return false;
    const auto smooth = [] (int v) {
        for (int q : { 2, 3, 5 })
            while (v % q == 0)
                v /= q;
        return v == 1;
    };
    if (! smooth (n_real))
        return false;
    // the split nearest to square (both passes then work on tiles of similar size); r1 = 32 j
    double best = 1e30;
    for (int a = 32; a <= 639; a += 32)
    {
        if (n_real % a != 0 || ! smooth (a))
            continue;
        const int b = n_real / a;
        if (b > 639 || b < 16 || (b & 1) != 0)
            continue;
        if (mode == 1 && (no_lists || ! has_static_tile (a) || ! has_static_tile (b)))
            continue; // (default: only where both cores have a compile-time stage list)
        const double score = a > b ? (double) a / b : (double) b / a;
        if (score < best)
        {
            best = score;
            r1 = a;
            r2 = b;
        }
    }
    return r1 != 0;
}

// not one of the power-of-two lengths that run as a radix-2 split around the register-direct two-pass tiles (2^min_log2 .. 2^21)
static bool real_or_small_r2 (int nc, int min_log2)
{
    if (min_log2 < 19)
        min_log2 = 19;
    return ! (is_pow2 (nc) && nc <= (1 << 21) && nc >= (1 << min_log2));
}

bool split_general (int n_complex, int& n1, int& n2)
{
    n1 = n2 = 0;
    if (n_complex <= 0)
        return false;
    // developer knob: CHOWDSP_FFT_GENERAL_SPLIT = "<n1>x<n2>" forces the general split of the one length it multiplies to (A/B of splits)
    if (const char* e = std::getenv ("CHOWDSP_FFT_GENERAL_SPLIT"))
    {
        int a = 0, b = 0;
        if (std::sscanf (e, "%dx%d", &a, &b) == 2 && a > 1 && b > 1 && (long long) a * b == (long long) n_complex)
        {
            n1 = a;
            n2 = b;
            return n1 <= kGeneralMaxFactor && n2 <= kGeneralMaxFactor;
        }
    }
    // two static tile lengths where the length has such a factorisation (38400 = 240 x 160 rather than 200 x 192), both of at
    // most 639 points (16-column tiles on both sides) if possible ...
    if (split_static_pair (n_complex, kMixedMaxN2Wide, n1, n2) || split_static_pair (n_complex, kTileInplaceWideMaxN, n1, n2) || split_static_pair (n_complex, 2048, n1, n2))
        return true;
    // ... else the divisor closest to sqrt(n) from below becomes the column length n2, the cofactor the row length n1
    int best = 0;
    for (int d = 1; (long long) d * d <= (long long) n_complex; ++d)
        if (n_complex % d == 0)
            best = d;
    if (best <= 1)
        return false;
    n2 = best;
    n1 = n_complex / best;
    return n1 <= kGeneralMaxFactor && n2 >= 2;
}

// 2^a 3^b 5^c ?  (the reference's accepted prime factors, simd/chowdsp_fft_impl_common.hpp:51-82)
static bool is_235_smooth (int n)
{
    if (n < 1)
        return false;
    for (int f : { 2, 3, 5 })
        while (n % f == 0)
            n /= f;
    return n == 1;
}

static constexpr int kGenericMaxComplexLen = 20000; // = cfft::kGenericMaxN (one LDS image per workgroup, in-place stages)

bool runs_as_huge (int n_complex)
{
    if (n_complex <= kMaxComplexLen)
        return false;
    int a = 0, b = 0;
    return ! (n_complex <= kMaxStaticTwoPassLen && split_static_pair (n_complex, 2048, a, b));
}

bool split_huge (int n_complex, int& r, int& c)
{
    r = c = 0;
    if (! runs_as_huge (n_complex) || n_complex > kMaxHugeComplexLen)
        return false;
    // the divisor pair closest to sqrt(n) whose members are both accepted complex lengths (multiples of 16, at most 2^21)
    for (long long d = (long long) std::sqrt ((double) n_complex) + 1; d >= 16; --d)
    {
        if (n_complex % d != 0)
            continue;
        const long long e = n_complex / d;
        if (d % 16 != 0 || e % 16 != 0 || e > kMaxComplexLen || d > kMaxComplexLen)
            continue;
        if (! size_supported ((int) d, 1) || ! size_supported ((int) e, 1))
            continue;
        c = (int) std::min (d, e);
        r = (int) std::max (d, e);
        return true;
    }
    return false;
}

bool split_huge_columns (int n_complex, int& r, int& c)
{
    r = c = 0;
    if (! runs_as_huge (n_complex) || n_complex > kMaxHugeComplexLen)
        return false;
    // CHOWDSP_FFT_HUGE_COLS_MAX (A/B): the longest column length tried.  Default 640: the lengths the column kernel takes 16 columns (whole 128-byte
    // lines) at a time -- measured c2^23 / c2^24 / c3*2^21 / c2^27: 0.164 / 0.156 / 0.151 / 0.137 of 8 TB/s against 0.141 / 0.137 / 0.118 / 0.119 with
    // 2048-point columns (8-column tiles: 64-byte runs), 0.134 / 0.142 / 0.126 / 0.130 with 1024, 0.147 / 0.145 / 0.117 / 0.137 with 256
    static const int dmax = [] { const char* e = std::getenv ("CHOWDSP_FFT_HUGE_COLS_MAX"); const int v = e != nullptr ? std::atoi (e) : 0; return (v >= 256 && v <= 2048) ? v : 640; }();
    for (int d = dmax - dmax % 16; d >= 256; d -= 16)
    {
        if (n_complex % d != 0 || ! is_235_smooth (d))
            continue;
        const long long e = n_complex / d;
        if (e % 16 != 0 || e > kMaxComplexLen || runs_as_huge ((int) e) || ! size_supported ((int) e, 1))
            continue;
        r = d;
        c = (int) e;
        return true;
    }
    return false;
}

bool size_supported (int N, int transform)
{
    // the reference needs N % 16 == 0 (complex) / N % 32 == 0 (real) on its SSE path
    // (simd/chowdsp_fft_impl_common.hpp:168-177) and N/4 free of primes other than 2, 3, 5 (:216-225).
    const bool real = (transform == 0 /* FFT_REAL */);
    if (N <= 0 || N % (real ? 32 : 16) != 0)
        return false;
    const int nc = real ? N / 2 : N;
    if (! is_235_smooth (N))
        return false;
    if (runs_as_huge (nc))
    {
        int hr = 0, hc = 0;
        if (real && nc > (1 << 24))
            return false; // (the pair stage's table has N / 4 + 1 entries: 64 MiB at real N = 2^25)
        return split_huge (nc, hr, hc); // frames beyond 2^21 complex points: two plans + transposes
    }
    if (is_pow2 (N))
        return true; // single pass up to 32768 complex points, two passes beyond
    // mixed-radix sizes run LDS-resident in the generic kernel: bounded by what the frame image needs
    if (nc <= kGenericMaxComplexLen)
        return true;
    // longer frames: two passes -- generic columns x power-of-two rows where such a split exists, else generic x generic
    int n1 = 0, n2 = 0;
    return split_mixed (nc, n1, n2) || split_general (nc, n1, n2);
}

// stage radices of the generic kernel for a 2^a 3^b 5^c length: fives, threes, then the power of two in as few
// stages as the radix cap allows, the bits spread evenly (2^10 -> 16, 8, 8; 2^5 -> 8, 4).  The cap keeps at least a quarter
// of the workgroup's work-items busy with a butterfly each: fewer, larger butterflies mean fewer LDS round trips
// (N=9216: 7 -> 5 stages, +33 %), but 48 radix-16 butterflies on 256 work-items (N=768) lost 16 %.
static void generic_radices (int n, std::vector<int>& radix, int frames_override)
{
    int frames = 1, block = 256;
    generic_geometry (n, frames, block);
    if (frames_override > 0) // column FFTs of a two-pass tile: that many frames per 256-thread workgroup
    {
        frames = frames_override;
        block = 256;
    }
    int cap_bits = 4; // radices up to 16 (the 1024-thread in-place kernel stopped at 8 while its stage loop carried every radix case's offsets)
    while (cap_bits > 2 && (frames * n) >> cap_bits < block / 4)
        --cap_bits;
    const int cap = 1 << cap_bits;
    // Fewest stages first (every stage but the first costs an LDS round trip), then the smallest sum of radices (= the most even
    // split: 2^10 -> 16 8 8, 9216 -> 12 12 8 8), over the radices the kernels have butterflies for.  Composite radices
    // (6, 9, 10, 12, 15: generic_kernels.hpp) merge a 3 or a 5 with a neighbour instead of giving it a stage of its own.
    // CHOWDSP_FFT_GENERIC_COMPOSITE=0: the round-1 lists (5s, 3s, then the power of two spread evenly).
    static const bool composite = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_GENERIC_COMPOSITE");
        return ! (e != nullptr && e[0] == '0');
    }();
    static const int all[11] = { 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2 };
    std::vector<int> best, cur;
    int best_sum = 0;
    const std::function<void (int, int, int)> search = [&] (int rest, int from, int sum) {
        if (rest == 1)
        {
            if (best.empty() || cur.size() < best.size() || (cur.size() == best.size() && sum < best_sum))
            {
                best = cur;
                best_sum = sum;
            }
            return;
        }
        if (! best.empty() && cur.size() + 1 > best.size())
            return;
        for (int i = from; i < 11; ++i)
        {
            const int r = all[i];
            const bool is_composite = (r == 6 || r == 9 || r == 10 || r == 12 || r == 15);
            if (r > cap || rest % r != 0 || (is_composite && ! composite))
                continue;
            cur.push_back (r);
            search (rest / r, i, sum + r);
            cur.pop_back();
        }
    };
    radix.clear();
    if (composite)
        search (n, 0, 0);
    if (best.empty())
    {
        // round-1 rule (also the fallback: always succeeds for 2-3-5-smooth n)
        while (n % 5 == 0) { radix.push_back (5); n /= 5; }
        while (n % 3 == 0) { radix.push_back (3); n /= 3; }
        int e = 0;
        while (n % 2 == 0) { ++e; n /= 2; }
        int stages = (e + cap_bits - 1) / cap_bits;
        while (e > 0)
        {
            const int bits = (e + stages - 1) / stages;
            radix.push_back (1 << bits);
            e -= bits;
            --stages;
        }
        return;
    }
    // stages with an odd factor first (largest first), then the powers of two, largest first
    const auto odd_part = [] (int r) { while (r % 2 == 0) r /= 2; return r; };
    std::stable_sort (best.begin(), best.end(), [&] (int a, int b) {
        const bool oa = odd_part (a) > 1, ob = odd_part (b) > 1;
        if (oa != ob)
            return oa;
        return a > b;
    });
    // two-stage plans: the last stage stores runs of n / (its radix) consecutive points per butterfly output.  Where that is not
    // a whole number of 128-byte lines but the other order makes it one (192 = 12 x 16: runs of 12 points; = 16 x 12: runs of 16),
    // the order with full-line stores is taken (the first stage then READS runs of 12, which costs less than partial-line stores)
#ifndef CFFT_TWO_STAGE_FULL_LINE
#define CFFT_TWO_STAGE_FULL_LINE 1 // (0: the sorted order -- with the matching lists of generic_kernels.hpp -- for A/B builds)
#endif
    // Measured on MI355X (tools/gpu_r3_two_stage_order_ab.sh, fraction of 8 TB/s, sorted / full-line order): c192 0.576 / 0.630,
    // c240 0.551 / 0.564, real N = 384 0.618 / 0.645, real N = 480 0.607 / 0.653, real N = 320 0.568 / 0.579.  n = 160 keeps the
    // sorted order: it runs the staged-I/O form (generic_kernels.hpp: generic_staged_io), which is faster with it (0.616 / 0.592;
    // direct I/O in the full-line order: 0.584).
    // (frames_override > 0: the list of a tile kernel, whose stages run LDS to LDS -- the sorted order stays)
    if (CFFT_TWO_STAGE_FULL_LINE && frames_override == 0 && best.size() == 2 && n != 160 && (n / best[1]) % 16 != 0 && (n / best[0]) % 16 == 0)
        std::swap (best[0], best[1]);
    // Three and more stages, same idea: the last stage stores runs of n / (its radix) points; where that is not a whole number of
    // 128-byte lines and moving another radix of the list to the end makes it one, the smallest such radix goes last
    // (480 = 10 x 6 x 8 -> 10 x 8 x 6: runs of 80 points instead of 60; c480 0.594 -> 0.652 of 8 TB/s).  Lengths in
    // kKeepSortedOrder measured no gain or a loss (profiles/r03_last_stage_order_ab.txt).
    if (CFFT_TWO_STAGE_FULL_LINE && frames_override == 0 && best.size() >= 3 && (n / best.back()) % 16 != 0)
    {
        static const int kKeepSortedOrder[] = { 0 };
        bool keep = false;
        for (int k : kKeepSortedOrder)
            keep = keep || k == n;
        int pick = -1;
        for (size_t i = 0; i < best.size() && ! keep; ++i)
            if ((n / best[i]) % 16 == 0 && (pick < 0 || best[i] < best[(size_t) pick]))
                pick = (int) i;
        if (pick >= 0)
        {
            const int r = best[(size_t) pick];
            best.erase (best.begin() + pick);
            best.push_back (r);
        }
    }
#ifndef CFFT_GENERIC_LISTS_V2
#define CFFT_GENERIC_LISTS_V2 1
#endif
    if (CFFT_GENERIC_LISTS_V2 && frames_override == 0)
    {
        // one-pass lists re-chosen by measurement (generic_kernels.hpp: StaticPlanFor carries the same table)
        // 160 / 192 / 240: three stages of radix <= 8 instead of two of radix 10 ... 16 (c192 +15 %, c240 +14 %, real N = 320 +11 %).  The in-place lengths:
        // another ORDER or split of the same factors, found by sweeping every ordered list through the run-time-list kernel and confirmed with the
        // compile-time lists (c9216 +11.6 % / real N = 18432 +3.9 %, c9600 +12.5 % / r19200 +5.8 %, c4608 +5.1 % / r9216 +9.3 %, c18432 +6.2 %, c6400 +5.3 %,
        // c2592 and c6912 +2 %); candidates that gained on the run-time kernel but not as compile-time lists are not here (profiles/r05_generic_lists.txt)
        static const struct { int n; std::vector<int> list; } kMeasured[] = {
            { 160, { 4, 8, 5 } }, { 192, { 6, 8, 4 } }, { 240, { 6, 8, 5 } },
            { 2592, { 6, 6, 12, 6 } }, { 4608, { 12, 8, 4, 12 } }, { 6400, { 10, 8, 16, 5 } }, { 6912, { 9, 16, 8, 6 } }, { 9216, { 12, 8, 8, 12 } }, { 9600, { 10, 8, 8, 15 } },
            { 18432, { 8, 16, 16, 9 } },
        };
        for (const auto& m : kMeasured)
            if (m.n == n)
                best = m.list;
    }
    // developer override for list sweeps (tools/generic_list_sweep.py), read at every plan construction: CHOWDSP_FFT_GENERIC_LIST="192:6,8,4;240:5,16,3"
    // -- a list that multiplies out to n and uses radices the kernels have replaces the planner's (such a plan runs the run-time-list kernel
    // unless it happens to equal the compile-time list of its length)
    if (const char* ov = std::getenv ("CHOWDSP_FFT_GENERIC_LIST"))
    {
        if (frames_override == 0)
        {
            const char* q = ov;
            while (*q != 0)
            {
                char* end = nullptr;
                const long key = std::strtol (q, &end, 10);
                if (end == q || *end != ':')
                    break;
                q = end + 1;
                std::vector<int> lst;
                long prod = 1;
                while (true)
                {
                    const long r = std::strtol (q, &end, 10);
                    if (end == q)
                        break;
                    lst.push_back ((int) r);
                    prod *= r;
                    q = end;
                    if (*q == ',')
                        ++q;
                    else
                        break;
                }
                bool ok = key == n && prod == n && ! lst.empty() && lst.size() <= 10;
                for (int r : lst)
                    ok = ok && (r == 2 || r == 3 || r == 4 || r == 5 || r == 6 || r == 8 || r == 9 || r == 10 || r == 12 || r == 15 || r == 16);
                if (ok)
                    best = lst;
                if (*q == ';')
                    ++q;
                else
                    break;
            }
        }
    }
    radix = best;
}

bool build_column_tables (int rows, HostTables& out)
{
    if (rows < 2 || rows > kGeneralMaxFactor || ! is_235_smooth (rows))
        return false;
    out.core_n = 0;
    generic_radices (rows, out.generic_radix, general_tile_width (rows));
    int p = 1;
    for (int r : out.generic_radix)
    {
        out.generic_tw_offset.push_back ((int) out.stage_twiddles.size());
        for (int a = 1; a < r; ++a)
            for (int k = 0; k < p; ++k)
                out.stage_twiddles.push_back (unit_root ((long long) a * k, (long long) p * r));
        p *= r;
    }
    if (out.stage_twiddles.empty())
        out.stage_twiddles.push_back (make_float2 (1.f, 0.f)); // single-stage plans read no twiddles; keep the table non-empty
    return p == rows && out.generic_radix.size() <= 10;
}

bool build_host_tables (int N, int transform, HostTables& out)
{
    if (! size_supported (N, transform))
        return false;
    const bool real = (transform == 0);
    const int nc = real ? N / 2 : N; // complex length actually transformed

    if (real)
    {
        // W_N^k for k = 0 .. N/4 (the pair stage touches k and N/2-k together)
        out.real_twiddles.reserve ((size_t) N / 4 + 1);
        for (int k = 0; k <= N / 4; ++k)
            out.real_twiddles.push_back (unit_root (k, N));
    }

    if (runs_as_huge (nc))
    {
        if (! split_huge (nc, out.huge_r, out.huge_c))
            return false;
        out.core_n = 0;
        int shift = 0;
        while ((1ll << (2 * shift)) < (long long) nc)
            ++shift;
        out.huge_shift = shift;
        const long long lo_len = 1ll << shift, hi_len = ((long long) nc + lo_len - 1) / lo_len;
        out.huge_tw_lo.reserve ((size_t) lo_len);
        out.huge_tw_hi.reserve ((size_t) hi_len);
        for (long long i = 0; i < lo_len; ++i)
            out.huge_tw_lo.push_back (unit_root (i, nc));
        for (long long i = 0; i < hi_len; ++i)
            out.huge_tw_hi.push_back (unit_root (i * lo_len, nc));
        return true;
    }
    {
        // 2^21 complex points (round 4): one radix-2 decimation in time around the 1024 x 1024 register-direct two-pass transform of the two
        // half-length sub-frames (tile_fast_kernels.hpp pass_*_fast_r2_kernel), unless CHOWDSP_FFT_FAST_TILES=0 / CHOWDSP_FFT_R2=0
        const char* ev = std::getenv ("CHOWDSP_FFT_FAST_TILES");
        const char* er = std::getenv ("CHOWDSP_FFT_R2");
        // which powers of two take this form: 2^19 .. 2^21.  2^21 has no plain two-pass plan on these tiles (0.183 -> 0.325 of 8 TB/s);
        // 2^20 / 2^19 gain 9 % / 3 % on their plain 1024 x 1024 / 1024 x 512 plans (half the inter-pass twiddle traffic, 16-byte loads:
        // profiles/r04_radix2_wrap.txt); CHOWDSP_FFT_R2_MIN_LOG2 = 20 / 21 / 22 takes them back to the plain plans (A/B)
        const char* em = std::getenv ("CHOWDSP_FFT_R2_MIN_LOG2");
        // (real frames of 2^21 / 2^22 samples -- complex cores of 2^20 / 2^21 points, no real four-step plan -- take it in any case)
        const int min_log2 = real ? 20 : (em != nullptr ? std::atoi (em) : 19);
        if (! real_or_small_r2 (nc, min_log2) && ! (ev != nullptr && ev[0] == '0') && ! (er != nullptr && er[0] == '0'))
        {
            static const int a32[10] = { 1, 2, 3, 4, 5, 6, 7, 8, 16, 24 };
            static const int a16[6] = { 1, 2, 3, 4, 8, 12 };
            static const int a8[4] = { 1, 2, 3, 4 };
            const int n1 = 1024, half = nc / 2, n2 = half / n1; // sub-frames of n1 x n2 points: n2-point columns, 1024-point rows
            out.core_n = 0;
            out.r2_n = n2;
            for (int n : { n2, n1 })
            {
                const int* al = n == 1024 ? a32 : (n == 512 ? a16 : a8);
                const int rows = n == 1024 ? 10 : (n == 512 ? 6 : 4);
                for (int j = 0; j < rows; ++j)
                    for (int k = 0; k < 32; ++k)
                        out.stage_twiddles.push_back (unit_root ((long long) al[j] * k, n));
            }
            // the radix-2 twiddle W_N^k, k = n2 (ms + 32 b) + kb, factorised: W_N^kb (kb < n2), then W_2048^ms (ms < 32); W_64^b is a constant
            for (int kb = 0; kb < n2; ++kb)
                out.stage_twiddles.push_back (unit_root (kb, nc));
            for (int ms = 0; ms < 32; ++ms)
                out.stage_twiddles.push_back (unit_root (ms, 2 * n1));
            out.big_twiddles.reserve ((size_t) half);
            for (int a = 0; a < n1; ++a)
                for (int kb = 0; kb < n2; ++kb)
                    out.big_twiddles.push_back (unit_root ((long long) a * kb, half));
            return true;
        }
    }
    {
        int g1 = 0, g2 = 0;
        if (real && split_real4_generic (N, g1, g2))
        {
            // long mixed-radix REAL frame: the four-step algorithm on the real data with run-time stage lists for both cores
            // (real4_generic_kernels.hpp).  Stage tables of the r2-point column core, then of the r1-point row core; W_N^{a kb} in the
            // workspace's order [column tile tA][kb <= r2 / 2][32], a = 32 tA + al (the layout of the power-of-two form's LDS-staged tiles)
            out.core_n = 0;
            out.real4_r1 = g1;
            out.real4_r2 = g2;
            out.real4_generic = get_real4_generic() == 2 ? 2 : 1; // (2: run-time stage lists only, kernels_real4_generic.hip)
            const auto add_list = [&out] (int n, int frames, std::vector<int>& radix, std::vector<int>& offs) {
                generic_radices (n, radix, frames);
                int p = 1;
                for (int r : radix)
                {
                    offs.push_back ((int) out.stage_twiddles.size());
                    for (int a = 1; a < r; ++a)
                        for (int k = 0; k < p; ++k)
                            out.stage_twiddles.push_back (unit_root ((long long) a * k, (long long) p * r));
                    p *= r;
                }
                return p == n && radix.size() <= 10;
            };
            if (! add_list (g2, 16, out.generic_radix, out.generic_tw_offset) || ! add_list (g1, 16, out.generic2_radix, out.generic2_tw_offset))
                return false;
            const int nr = g2 / 2 + 1;
            out.big_twiddles.reserve ((size_t) g1 * nr);
            for (int ta = 0; ta < g1 / 32; ++ta)
                for (int kb = 0; kb < nr; ++kb)
                    for (int al = 0; al < 32; ++al)
                        out.big_twiddles.push_back (unit_root ((long long) (ta * 32 + al) * kb, N));
            return true;
        }
    }
    int gs1 = 0, gs2 = 0;
    if ((! is_pow2 (nc) && nc > kGenericMaxComplexLen && ! split_mixed (nc, gs1, gs2)) || (is_pow2 (nc) && nc > (1 << 20)))
    {
        // general two-pass form: nc = n1 * n2, n2-point column FFTs and n1-point row FFTs both by the generic kernel
        int n1 = 0, n2 = 0;
        if (! split_general (nc, n1, n2))
            return false;
        out.core_n = 0;
        out.mixed_n1 = n1;
        out.mixed_n2 = n2;
        const int ca = general_tile_width (n2), cb = general_tile_width (n1);
        out.mixed_ca = ca;
        out.mixed_cb = cb;
        const auto add_list = [&out] (int n, int frames, std::vector<int>& radix, std::vector<int>& offs) {
            generic_radices (n, radix, frames);
            int p = 1;
            for (int r : radix)
            {
                offs.push_back ((int) out.stage_twiddles.size());
                for (int a = 1; a < r; ++a)
                    for (int k = 0; k < p; ++k)
                        out.stage_twiddles.push_back (unit_root ((long long) a * k, (long long) p * r));
                p *= r;
            }
            return p == n && radix.size() <= 10;
        };
        if (! add_list (n2, ca, out.generic_radix, out.generic_tw_offset) || ! add_list (n1, cb, out.generic2_radix, out.generic2_tw_offset))
            return false;
        // W_nc^{a*k2} in the workspace's tile-major order [tile tA][k2][c], a = tA*ca + c; columns past n1 (ragged last tile): 1
        const int tiles = (n1 + ca - 1) / ca;
        out.big_twiddles.reserve ((size_t) tiles * n2 * ca);
        for (int ta = 0; ta < tiles; ++ta)
            for (int k2 = 0; k2 < n2; ++k2)
                for (int c = 0; c < ca; ++c)
                {
                    const int a = ta * ca + c;
                    out.big_twiddles.push_back (a < n1 ? unit_root ((long long) a * k2, nc) : make_float2 (1.f, 0.f));
                }
        return true;
    }
    if (! is_pow2 (nc) && nc > kGenericMaxComplexLen)
    {
        // long mixed-radix frame: nc = n1 * n2, generic n2-point column FFTs then power-of-two n1-point row FFTs
        int n1 = 0, n2 = 0;
        if (! split_mixed (nc, n1, n2))
            return false;
        out.core_n = 0;
        out.mixed_n1 = n1;
        out.mixed_n2 = n2;
        const int ca = mixed_tile_width (n2);
        out.mixed_ca = ca;
        generic_radices (n2, out.generic_radix, ca);
        int p = 1;
        for (int r : out.generic_radix)
        {
            out.generic_tw_offset.push_back ((int) out.stage_twiddles.size());
            for (int a = 1; a < r; ++a)
                for (int k = 0; k < p; ++k)
                    out.stage_twiddles.push_back (unit_root ((long long) a * k, (long long) p * r));
            p *= r;
        }
        if (p != n2 || out.generic_radix.size() > 10)
            return false;
        PlanDesc d1;
        if (! core_plan_desc (n1, d1))
            return false;
        out.mixed_tw1_offset = (int) out.stage_twiddles.size();
        append_stage_twiddles (d1, out.stage_twiddles);
        // W_nc^{a*k2} in the workspace's tile-major order: [tile tA][k2][c], a = tA*ca + c (multipass_kernels.hpp)
        out.big_twiddles.reserve ((size_t) nc);
        for (int ta = 0; ta < n1 / ca; ++ta)
            for (int k2 = 0; k2 < n2; ++k2)
                for (int c = 0; c < ca; ++c)
                    out.big_twiddles.push_back (unit_root ((long long) (ta * ca + c) * k2, nc));
        return true;
    }
    if (! is_pow2 (nc))
    {
        // generic mixed-radix plan: per stage (r-1)*p entries [a-1][k] = W_{p*r}^{a*k}
        out.core_n = 0;
        generic_radices (nc, out.generic_radix, 0);
        int p = 1;
        for (int r : out.generic_radix)
        {
            out.generic_tw_offset.push_back ((int) out.stage_twiddles.size());
            for (int a = 1; a < r; ++a)
                for (int k = 0; k < p; ++k)
                    out.stage_twiddles.push_back (unit_root ((long long) a * k, (long long) p * r));
            p *= r;
        }
        return p == nc && out.generic_radix.size() <= 10;
    }

    int q1 = 0, q2 = 0;
    if (real && split_real4 (N, q1, q2))
    {
        // long real frame: four-step on the real data.  Stage tables of the two cores (columns first), then W_N^{a kb} in the
        // workspace's order [column tile tA][kb <= r2/2][32], a = 32 tA + al
        const char* evr = std::getenv ("CHOWDSP_FFT_FAST_TILES");
        // measured (forward / backward, TB/s algorithmic, register-direct vs LDS-staged tiles): 2^17 2.4 / 2.6 vs 2.1, 2^18 2.3 / 2.6
        // vs 1.9, 2^19 2.3 / 2.4 vs 1.7, 2^20 2.1 / 2.2 vs 1.8 / 1.5
        if (! (evr != nullptr && evr[0] == '0'))
        {
            // register-direct tiles (tile_fast_kernels.hpp): factorised stage-1 rows of the column core (q2) then of the row
            // core (q1); forward inter-pass table [a][kb < q2/2 + 16] (W_N^{a kb}, padding rows 1), backward table [kb <= q2/2][a]
            const auto rows = [] (int n) { return n == 1024 ? 10 : (n == 512 ? 6 : 4); };
            static const int a32[10] = { 1, 2, 3, 4, 5, 6, 7, 8, 16, 24 };
            static const int a16[6] = { 1, 2, 3, 4, 8, 12 };
            static const int a8[4] = { 1, 2, 3, 4 };
            out.core_n = 0;
            out.real4_r1 = q1;
            out.real4_r2 = q2;
            out.real4_fast = 1;
            for (int n : { q2, q1 })
            {
                const int* al = n == 1024 ? a32 : (n == 512 ? a16 : a8);
                for (int j = 0; j < rows (n); ++j)
                    for (int k = 0; k < 32; ++k)
                        out.stage_twiddles.push_back (unit_root ((long long) al[j] * k, n));
            }
            const int nr = q2 / 2 + 1, nrp = q2 / 2 + 16;
            out.big_twiddles.reserve ((size_t) q1 * nrp + (size_t) nr * q1);
            for (int a = 0; a < q1; ++a)
                for (int kb = 0; kb < nrp; ++kb)
                    out.big_twiddles.push_back (kb < nr ? unit_root ((long long) a * kb, N) : make_float2 (1.f, 0.f));
            for (int kb = 0; kb < nr; ++kb)
                for (int a = 0; a < q1; ++a)
                    out.big_twiddles.push_back (unit_root ((long long) a * kb, N));
            return true;
        }
        PlanDesc dc, dr;
        if (! core_plan_desc (q2, dc) || ! core_plan_desc (q1, dr))
            return false;
        out.core_n = 0;
        out.real4_r1 = q1;
        out.real4_r2 = q2;
        append_stage_twiddles (dc, out.stage_twiddles);
        append_stage_twiddles (dr, out.stage_twiddles);
        const int nr = q2 / 2 + 1;
        out.big_twiddles.reserve ((size_t) q1 * nr);
        for (int ta = 0; ta < q1 / 32; ++ta)
            for (int kb = 0; kb < nr; ++kb)
                for (int al = 0; al < 32; ++al)
                    out.big_twiddles.push_back (unit_root ((long long) (ta * 32 + al) * kb, N));
        return true;
    }
    PlanDesc d;
    if (nc <= kMaxCoreN)
    {
        out.core_n = nc;
        if (! core_plan_desc (nc, d))
            return false;
        append_stage_twiddles (d, out.stage_twiddles);
        if (nc == 16384 || nc == 8192 || nc == 2048) // the split-plane form of this length: its table follows the interleaved-image plan's
            append_split_plane_twiddles (nc, out.stage_twiddles);
        if (nc == 4096)
        {
            // one-wavefront-per-frame form (wave_kernels.hpp), 64 x 64: W_4096^(a t), t < 64, rows a = 1..7 then 8, 16, .. 56
            for (int a = 1; a < 8; ++a)
                for (int t = 0; t < 64; ++t)
                    out.stage_twiddles.push_back (unit_root ((long long) a * t, 4096));
            for (int a = 8; a < 64; a += 8)
                for (int t = 0; t < 64; ++t)
                    out.stage_twiddles.push_back (unit_root ((long long) a * t, 4096));
        }
        return true;
    }
    if (nc == kSplitCoreN)
    {
        // split-plane single pass, 32 x 32 x 32 (split_plane_kernels.hpp)
        out.core_n = nc;
        append_split_plane_twiddles (nc, out.stage_twiddles);
        return true;
    }

    {
        // N = 2^17 .. 2^20 complex: register-direct passes (tile_fast_kernels.hpp) unless CHOWDSP_FFT_FAST_TILES=0
        int f1 = 0, f2 = 0;
        split_large (nc, f1, f2);
        const char* ev = std::getenv ("CHOWDSP_FFT_FAST_TILES");
        const bool off = ev != nullptr && ev[0] == '0';
        const auto rows = [] (int n) { return n == 1024 ? 10 : (n == 512 ? 6 : (n == 256 ? 4 : 0)); };
        const char* e16 = std::getenv ("CHOWDSP_FFT_FAST_TILES_65536"); // experiment: N = 65536 through these kernels too
        const int min_nc = (e16 != nullptr && e16[0] == '1') ? (1 << 16) : (1 << 17);
        if (! real && ! off && nc >= min_nc && rows (f1) != 0 && rows (f2) != 0)
        {
            out.core_n = 0;
            out.fast_n1 = f1;
            out.fast_n2 = f2;
            // stage-1 rows [j][k < 32] of W_n^(a k): n = 1024: a = 1..7, 8, 16, 24; 512: 1, 2, 3, 4, 8, 12; 256: 1, 2, 3, 4
            static const int a32[10] = { 1, 2, 3, 4, 5, 6, 7, 8, 16, 24 };
            static const int a16[6] = { 1, 2, 3, 4, 8, 12 };
            static const int a8[4] = { 1, 2, 3, 4 };
            for (int n : { f2, f1 })
            {
                const int* al = n == 1024 ? a32 : (n == 512 ? a16 : a8);
                for (int j = 0; j < rows (n); ++j)
                    for (int k = 0; k < 32; ++k)
                        out.stage_twiddles.push_back (unit_root ((long long) al[j] * k, n));
            }
            out.big_twiddles.reserve ((size_t) nc);
            for (int a = 0; a < f1; ++a)
                for (int kb = 0; kb < f2; ++kb)
                    out.big_twiddles.push_back (unit_root ((long long) a * kb, nc));
            return true;
        }
    }
    // multi-pass: nc = n1 * n2.  stage tables of the two cores are stored back to back
    // (n2 core first), the inter-pass twiddle W_nc^{a*k2}, a < n1, k2 < n2, in tile-major order.
    int n1 = 0, n2 = 0;
    split_large (nc, n1, n2);
    out.core_n = 0;
    PlanDesc d2, d1;
    if (n1 == 256 && n2 == 256)
        d2 = d1 = describe<Plan256>(); // the fast path's own radix-16 x 16 plan
    else if (! core_plan_desc (n2, d2) || ! core_plan_desc (n1, d1))
        return false;
    append_stage_twiddles (d2, out.stage_twiddles);
    append_stage_twiddles (d1, out.stage_twiddles);
    out.big_twiddles.reserve ((size_t) nc);
    if (n1 == 256 && n2 == 256)
    {
        // 65536 = 256 x 256 fast path: 16 x 16 block order [a>>4][k2>>4][a&15][k2&15] (multipass_kernels.hpp)
        for (int ah = 0; ah < 16; ++ah)
            for (int kh = 0; kh < 16; ++kh)
                for (int al = 0; al < 16; ++al)
                    for (int kl = 0; kl < 16; ++kl)
                        out.big_twiddles.push_back (unit_root ((long long) (ah * 16 + al) * (kh * 16 + kl), nc));
        return true;
    }
    // W_nc^{a*k2} in the workspace's tile-major order: [tile tA][k2][c], a = tA*CA + c (multipass_kernels.hpp)
    const int ca = tile_width_for (n2);
    for (int ta = 0; ta < n1 / ca; ++ta)
        for (int k2 = 0; k2 < n2; ++k2)
            for (int c = 0; c < ca; ++c)
                out.big_twiddles.push_back (unit_root ((long long) (ta * ca + c) * k2, nc));
    return true;
}
} // namespace cfft::host
