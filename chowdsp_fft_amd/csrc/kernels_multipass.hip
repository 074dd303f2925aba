// Instantiation + launch of the two-pass (N > 4096) kernels and the stand-alone real pair stage.
#include "launch_util.hpp"
#include "multipass_kernels.hpp"
#include "fused_twopass_kernels.hpp"
#include "group_twopass_kernels.hpp"
#include "group_fast_kernels.hpp"
#include "tile_fast_kernels.hpp"
#include "generic_kernels.hpp"
#include "tables.hpp"
#include "tile_static.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace cfft::host
{
namespace
{
template <int N>
struct TileGeom
{
    using P = typename CorePlan<N>::type;
    static constexpr int C = tile_width (N, P::T);
    static constexpr size_t lds_bytes = (size_t) C * TileFrame<N>::stride * sizeof (float2);
    // work-items per workgroup: enough teams for every frame of the tile at once where that needs more than 256
    // (CFFT_TILE_BLOCK_MIN_N, plans.hpp): one round of per-frame FFTs instead of C * T / 256 rounds with barriers between
    static constexpr int BLOCK = (N >= CFFT_TILE_BLOCK_MIN_N && C * P::T > 256) ? (C * P::T > 1024 ? 1024 : C * P::T) : 256;
};

// (the dynamic-LDS cap of a kernel is raised once per kernel, device and size by persistent_grid -> blocks_per_cu_cached)
// pass A: n2-point column FFTs (+ inter-pass twiddle), in -> work (tile-major)
template <int N2, bool INV>
hipError_t pass_a (int n1, const float2* in, float2* work, const float2* tw2, const float2* big_tw, size_t batch, int div, hipStream_t stream, size_t in_stride)
{
    using TG = TileGeom<N2>;
    auto kernel = pass_a_kernel<typename TG::P, TG::C, INV, TG::BLOCK>;
    const size_t tiles = (size_t) (n1 / TG::C);
    unsigned grid = persistent_grid (kernel, TG::BLOCK, TG::lds_bytes, batch * tiles);
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (TG::BLOCK), TG::lds_bytes, stream, in, work, tw2, big_tw, (unsigned) n1, (unsigned) batch, in_stride);
    return hipGetLastError();
}

// pass B: n1-point row FFTs, transposed store, work -> out
template <int N1, bool INV>
hipError_t pass_b (int n2, const float2* work, float2* out, const float2* tw1, size_t batch, int div, hipStream_t stream, size_t out_stride, int ca = 0)
{
    using TG = TileGeom<N1>;
    auto kernel = pass_b_kernel<typename TG::P, TG::C, INV, TG::BLOCK>;
    const size_t tiles = (size_t) (n2 / TG::C);
    unsigned grid = persistent_grid (kernel, TG::BLOCK, TG::lds_bytes, batch * tiles);
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    // pass A's tile width: that of the n2 core, or given explicitly (mixed-radix column pass: 16)
    const int ca_width = ca > 0 ? ca : tile_width_for (n2);
    unsigned ca_log2 = 0;
    while ((1 << ca_log2) < ca_width)
        ++ca_log2;
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (TG::BLOCK), TG::lds_bytes, stream, work, out, tw1, (unsigned) n2, ca_log2, (unsigned) batch, out_stride);
    return hipGetLastError();
}

#define CFFT_DISPATCH_BIG(n, CALL)                           \
    switch (n)                                               \
    {                                                        \
        case 64: { constexpr int N_ = 64; CALL; } break;     \
        case 128: { constexpr int N_ = 128; CALL; } break;   \
        case 256: { constexpr int N_ = 256; CALL; } break;   \
        case 512: { constexpr int N_ = 512; CALL; } break;   \
        case 1024: { constexpr int N_ = 1024; CALL; } break; \
        default: return hipErrorInvalidValue;                \
    }

// N = 65536 = 256 x 256 fast path (register-direct, software-pipelined; multipass_kernels.hpp)
template <bool INV>
hipError_t big256 (const float2* in, float2* work, float2* out, const float2* tw256, const float2* big_tw, size_t batch, int div, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    constexpr size_t lds = (size_t) 16 * TileFrame<256>::stride * sizeof (float2);
    auto ka = pass_a256_kernel<INV>;
    auto kb = pass_b256_kernel<INV>;
    unsigned ga = persistent_grid (ka, 256, lds, batch * 16);
    unsigned gb = persistent_grid (kb, 256, lds, batch * 16);
    if (ga == 0 || gb == 0)
        return hipErrorNoDevice;
    ga = std::max (1u, ga / (unsigned) div);
    gb = std::max (1u, gb / (unsigned) div);
    hipLaunchKernelGGL (ka, dim3 (ga), dim3 (256), lds, stream, in, work, tw256, big_tw, (unsigned) batch, in_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL (kb, dim3 (gb), dim3 (256), lds, stream, (const float2*) work, out, tw256, (unsigned) batch, out_stride);
    return hipGetLastError();
}

int stage_table_len (int n)
{
    PlanDesc d;
    return core_plan_desc (n, d) ? d.tw_total : 0;
}
} // namespace

hipError_t launch_big_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    const size_t frame_len = (size_t) n1 * (size_t) n2;
    in_stride = in_stride ? in_stride : frame_len;
    out_stride = out_stride ? out_stride : frame_len;
    const float2* tw2 = stage_tw;
    const float2* tw1 = stage_tw + stage_table_len (n2);
    if (n1 == 256 && n2 == 256 && batch * 16 <= 0x7FFFFFFFull)
        return inverse ? big256<true> (in, work, out, tw2, big_tw, batch, grid_divisor, stream, in_stride, out_stride)
                       : big256<false> (in, work, out, tw2, big_tw, batch, grid_divisor, stream, in_stride, out_stride);
    hipError_t e = hipErrorInvalidValue;
    if (inverse)
    {
        CFFT_DISPATCH_BIG (n2, e = (pass_a<N_, true> (n1, in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride)));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_BIG (n1, e = (pass_b<N_, true> (n2, work, out, tw1, batch, grid_divisor, stream, out_stride)));
    }
    else
    {
        CFFT_DISPATCH_BIG (n2, e = (pass_a<N_, false> (n1, in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride)));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_BIG (n1, e = (pass_b<N_, false> (n2, work, out, tw1, batch, grid_divisor, stream, out_stride)));
    }
    return e;
}

// ---- N = 2^17 .. 2^20 complex: register-direct passes (tile_fast_kernels.hpp) ------------------------------------------
namespace
{
template <int N2, bool INV>
hipError_t fast_pass_a (int n1, const float2* in, float2* work, const float2* tw2, const float2* big_tw, size_t batch, int div, hipStream_t stream, size_t in_stride)
{
    using FT = FastTile<N2>;
    auto kernel = pass_a_fast_kernel<N2, INV>;
    unsigned grid = persistent_grid (kernel, FT::threads, FT::lds_bytes, batch * (size_t) (n1 / 16));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (FT::threads), FT::lds_bytes, stream, in, work, tw2, big_tw, (unsigned) n1, (unsigned) batch, in_stride);
    return hipGetLastError();
}
template <int N1, bool INV>
hipError_t fast_pass_b (int n2, const float2* work, float2* out, const float2* tw1, size_t batch, int div, hipStream_t stream, size_t out_stride, const float2* mul_h = nullptr, float mul_scale = 1.f)
{
    using FT = FastTile<N1>;
    // (the product with a filter spectrum rides on the FORWARD transform's stores only: fft_convolve_frames_batch)
    auto kernel = (mul_h != nullptr && ! INV) ? pass_b_fast_kernel<N1, false, true> : pass_b_fast_kernel<N1, INV, false>;
    unsigned grid = persistent_grid (kernel, FT::threads, FT::lds_bytes, batch * (size_t) (n2 / 16));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (FT::threads), FT::lds_bytes, stream, work, out, tw1, (unsigned) n2, (unsigned) batch, out_stride, mul_h, mul_scale);
    return hipGetLastError();
}
#define CFFT_DISPATCH_FAST(n, CALL)                          \
    switch (n)                                               \
    {                                                        \
        case 256: { constexpr int N_ = 256; CALL; } break;   \
        case 512: { constexpr int N_ = 512; CALL; } break;   \
        case 1024: { constexpr int N_ = 1024; CALL; } break; \
        default: return hipErrorInvalidValue;                \
    }
} // namespace

int fast_tile_table_rows (int n) { return n == 1024 ? 10 : (n == 512 ? 6 : (n == 256 ? 4 : 0)); }

size_t real4_fast_workspace_frame_len (int r1, int r2) { return (size_t) r1 * (size_t) (r2 / 2 + 16); }

// (CFFT_REAL4_FAST_LAUNCH names its kernels as templates of the tile length alone)
template <int N_>
constexpr auto pass_b_real_fast_plain = pass_b_real_fast_kernel<N_, false>;
template <int N_>
constexpr auto pass_b_real_fast_mulh = pass_b_real_fast_kernel<N_, true>;


// long REAL frames on the register-direct tiles: N = r1 x r2 samples, r1, r2 in {256, 512, 1024}
hipError_t launch_real4_fast (int r1, int r2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride, const float2* mul_h, float mul_scale)
{
    if (mul_h != nullptr && inverse)
        return hipErrorInvalidValue;
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || fast_tile_table_rows (r1) == 0 || fast_tile_table_rows (r2) == 0)
        return hipErrorInvalidValue;
    const size_t m = (size_t) r1 * (size_t) r2 / 2;
    in_stride = in_stride ? in_stride : m;
    out_stride = out_stride ? out_stride : m;
    const float2* tw2 = stage_tw;
    const float2* tw1 = stage_tw + fast_tile_table_rows (r2) * 32;
    const float2* big_fwd = big_tw;                                              // [a][kb < r2/2 + 16]
    const float2* big_inv = big_tw + real4_fast_workspace_frame_len (r1, r2);   // [kb <= r2/2][a]
    const int n1c = r1 / 2;
    const unsigned div = (unsigned) grid_divisor;
    const size_t tiles_a = (size_t) (n1c / 16), tiles_b = (size_t) ((r2 / 2 + 16) / 16);
#define CFFT_REAL4_FAST_LAUNCH(KERNEL, NN, TILES, ...) /* TILES: work items of the launch */                                                              \
    {                                                                                                               \
        using FT = FastTile<NN>;                                                                                    \
        auto k_ = KERNEL<NN>;                                                                                       \
        unsigned grid_ = persistent_grid (k_, FT::threads, FT::lds_bytes, (size_t) (TILES));                        \
        if (grid_ == 0)                                                                                             \
            return hipErrorNoDevice;                                                                                \
        grid_ = std::max (1u, grid_ / div);                                                                         \
        hipLaunchKernelGGL (k_, dim3 (grid_), dim3 (FT::threads), FT::lds_bytes, stream, __VA_ARGS__);              \
        e = hipGetLastError();                                                                                      \
    }
    hipError_t e = hipErrorInvalidValue;
    if (! inverse)
    {
        CFFT_DISPATCH_FAST (r2, CFFT_REAL4_FAST_LAUNCH (pass_a_real_fast_kernel, N_, batch * tiles_a, in, work, tw2, big_fwd, (unsigned) n1c, (unsigned) batch, in_stride));
        if (e != hipSuccess)
            return e;
        if (mul_h != nullptr) // the product with a packed filter spectrum rides on the stores (fft_convolve_frames_batch)
        {
            CFFT_DISPATCH_FAST (r1, CFFT_REAL4_FAST_LAUNCH (pass_b_real_fast_mulh, N_, batch * tiles_b, (const float2*) work, out, tw1, (unsigned) r2, (unsigned) batch, out_stride, mul_h, mul_scale));
        }
        else
        {
            CFFT_DISPATCH_FAST (r1, CFFT_REAL4_FAST_LAUNCH (pass_b_real_fast_plain, N_, batch * tiles_b, (const float2*) work, out, tw1, (unsigned) r2, (unsigned) batch, out_stride, (const float2*) nullptr, 1.f));
        }
    }
    else
    {
        CFFT_DISPATCH_FAST (r1, CFFT_REAL4_FAST_LAUNCH (pass_b_real_inv_fast_kernel, N_, batch * tiles_b, in, work, tw1, big_inv, (unsigned) r2, (unsigned) batch, in_stride));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_FAST (r2, CFFT_REAL4_FAST_LAUNCH (pass_a_real_inv_fast_kernel, N_, batch * tiles_a, (const float2*) work, out, tw2, (unsigned) n1c, (unsigned) batch, out_stride));
    }
#undef CFFT_REAL4_FAST_LAUNCH
    return e;
}

hipError_t launch_fast_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride, const float2* mul_h, float mul_scale)
{
    if (mul_h != nullptr && inverse)
        return hipErrorInvalidValue;
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull || fast_tile_table_rows (n1) == 0 || fast_tile_table_rows (n2) == 0)
        return hipErrorInvalidValue;
    const size_t frame_len = (size_t) n1 * (size_t) n2;
    in_stride = in_stride ? in_stride : frame_len;
    out_stride = out_stride ? out_stride : frame_len;
    const float2* tw2 = stage_tw;                                   // n2 core (columns)
    const float2* tw1 = stage_tw + fast_tile_table_rows (n2) * 32;  // n1 core (rows)
    hipError_t e = hipErrorInvalidValue;
    if (inverse)
    {
        CFFT_DISPATCH_FAST (n2, e = (fast_pass_a<N_, true> (n1, in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride)));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_FAST (n1, e = (fast_pass_b<N_, true> (n2, work, out, tw1, batch, grid_divisor, stream, out_stride)));
    }
    else
    {
        CFFT_DISPATCH_FAST (n2, e = (fast_pass_a<N_, false> (n1, in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride)));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_FAST (n1, e = (fast_pass_b<N_, false> (n2, work, out, tw1, batch, grid_divisor, stream, out_stride, mul_h, mul_scale)));
    }
    return e;
}

namespace
{
template <int N2, bool INV>
hipError_t fast_r2_pass_a (const float2* in, float2* work, const float2* tw2, const float2* big_tw, size_t batch, int div, hipStream_t stream, size_t in_stride)
{
    using FT = FastTile<N2>;
    auto k = pass_a_fast_r2_kernel<N2, INV>;
    unsigned grid = persistent_grid (k, FT::threads, FT::lds_bytes, batch * (size_t) (1024 / 16));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (FT::threads), FT::lds_bytes, stream, reinterpret_cast<const float4*> (in), work, tw2, big_tw, 1024u, (unsigned) batch, in_stride / 2);
    return hipGetLastError();
}
} // namespace

// n2 = column length of the half-length sub-frames (256 / 512 / 1024); their rows are 1024 points
hipError_t launch_fast_r2_c2c (int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride, const float2* mul_h, float mul_scale)
{
    if (mul_h != nullptr && inverse)
        return hipErrorInvalidValue;
    if (batch == 0)
        return hipSuccess;
    if (fast_tile_table_rows (n2) == 0 || batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    constexpr int kN1 = 1024;
    using FB = FastTile<kN1>;
    const size_t frame_len = (size_t) 2 * kN1 * (size_t) n2;
    in_stride = in_stride ? in_stride : frame_len;
    out_stride = out_stride ? out_stride : frame_len;
    if ((in_stride & 1u) != 0)
        return hipErrorInvalidValue; // (frames are read in 16-byte pairs; the C API only ever passes whole float2 counts of even length)
    const float2* tw2 = stage_tw;                                     // column core
    const float2* tw1 = stage_tw + fast_tile_table_rows (n2) * 32;    // row core
    const float2* r2_tw = tw1 + fast_tile_table_rows (kN1) * 32;
    hipError_t e = hipErrorInvalidValue;
    if (inverse)
    {
        CFFT_DISPATCH_FAST (n2, e = (fast_r2_pass_a<N_, true> (in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride)));
    }
    else
    {
        CFFT_DISPATCH_FAST (n2, e = (fast_r2_pass_a<N_, false> (in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride)));
    }
    if (e != hipSuccess)
        return e;
    // (the product with a filter spectrum rides on the FORWARD transform's stores only: fft_convolve_frames_batch)
    auto kb = inverse ? pass_b_fast_r2_kernel<kN1, true, false> : (mul_h != nullptr ? pass_b_fast_r2_kernel<kN1, false, true> : pass_b_fast_r2_kernel<kN1, false, false>);
    unsigned grid = persistent_grid (kb, FB::threads, FB::lds_bytes, batch * (size_t) (n2 / 16));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) grid_divisor);
    hipLaunchKernelGGL (kb, dim3 (grid), dim3 (FB::threads), FB::lds_bytes, stream, (const float2*) work, out, tw1, r2_tw, (unsigned) n2, (unsigned) batch, out_stride, mul_h, mul_scale);
    return hipGetLastError();
}

// ---- long REAL frames: the four-step algorithm on the real data (multipass_kernels.hpp), two sweeps instead of three ----
namespace
{
template <int R2>
hipError_t real4_pass_a (int n1c, const float2* in, float2* work, const float2* tw2, const float2* big_tw, size_t batch, int div, hipStream_t stream, size_t in_stride, size_t ws_len)
{
    using TG = TileGeom<R2>;
    static_assert (TG::C == 16, "the real four-step tiles are 16 complex = 32 real columns wide");
    auto kernel = pass_a_real_kernel<typename TG::P, TG::C, TG::BLOCK>;
    unsigned grid = persistent_grid (kernel, TG::BLOCK, TG::lds_bytes, batch * (size_t) (n1c / TG::C));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (TG::BLOCK), TG::lds_bytes, stream, in, work, tw2, big_tw, (unsigned) n1c, (unsigned) batch, in_stride, ws_len);
    return hipGetLastError();
}
template <int R1>
hipError_t real4_pass_b (int r2, const float2* work, float2* out, const float2* tw1, size_t batch, int div, hipStream_t stream, size_t out_stride, size_t ws_len)
{
    using TG = TileGeom<R1>;
    static_assert (TG::C == 16, "row tiles of 16");
    auto kernel = pass_b_real_kernel<typename TG::P, TG::C, TG::BLOCK>;
    const size_t tiles = ((size_t) r2 / 2 + 1 + TG::C - 1) / TG::C;
    unsigned grid = persistent_grid (kernel, TG::BLOCK, TG::lds_bytes, batch * tiles);
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (TG::BLOCK), TG::lds_bytes, stream, work, out, tw1, (unsigned) r2, (unsigned) batch, out_stride, ws_len);
    return hipGetLastError();
}
template <int R1>
hipError_t real4_pass_b_inv (int r2, const float2* in, float2* work, const float2* tw1, const float2* big_tw, size_t batch, int div, hipStream_t stream, size_t in_stride, size_t ws_len)
{
    using TG = TileGeom<R1>;
    auto kernel = pass_b_real_inv_kernel<typename TG::P, TG::C, TG::BLOCK>;
    const size_t tiles = ((size_t) r2 / 2 + 1 + TG::C - 1) / TG::C;
    unsigned grid = persistent_grid (kernel, TG::BLOCK, TG::lds_bytes, batch * tiles);
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (TG::BLOCK), TG::lds_bytes, stream, in, work, tw1, big_tw, (unsigned) r2, (unsigned) batch, in_stride, ws_len);
    return hipGetLastError();
}
template <int R2>
hipError_t real4_pass_a_inv (int n1c, const float2* work, float2* out, const float2* tw2, size_t batch, int div, hipStream_t stream, size_t out_stride, size_t ws_len)
{
    using TG = TileGeom<R2>;
    auto kernel = pass_a_real_inv_kernel<typename TG::P, TG::C, TG::BLOCK>;
    unsigned grid = persistent_grid (kernel, TG::BLOCK, TG::lds_bytes, batch * (size_t) (n1c / TG::C));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) div);
    hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (TG::BLOCK), TG::lds_bytes, stream, work, out, tw2, (unsigned) n1c, (unsigned) batch, out_stride, ws_len);
    return hipGetLastError();
}
#define CFFT_DISPATCH_REAL4(n, CALL)                         \
    switch (n)                                               \
    {                                                        \
        case 256: { constexpr int N_ = 256; CALL; } break;   \
        case 512: { constexpr int N_ = 512; CALL; } break;   \
        case 1024: { constexpr int N_ = 1024; CALL; } break; \
        default: return hipErrorInvalidValue;                \
    }
} // namespace

size_t real4_workspace_frame_len (int r1, int r2) { return (size_t) r1 * (size_t) (r2 / 2 + 1); }

hipError_t launch_real4 (int r1, int r2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (batch > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    const size_t m = (size_t) r1 * (size_t) r2 / 2; // float2 per frame on the caller's side
    in_stride = in_stride ? in_stride : m;
    out_stride = out_stride ? out_stride : m;
    const size_t ws_len = real4_workspace_frame_len (r1, r2);
    const float2* tw2 = stage_tw;                          // R2-point column core
    const float2* tw1 = stage_tw + stage_table_len (r2);   // R1-point row core
    const int n1c = r1 / 2;
    hipError_t e = hipErrorInvalidValue;
    if (! inverse)
    {
        CFFT_DISPATCH_REAL4 (r2, e = (real4_pass_a<N_> (n1c, in, work, tw2, big_tw, batch, grid_divisor, stream, in_stride, ws_len)));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_REAL4 (r1, e = (real4_pass_b<N_> (r2, work, out, tw1, batch, grid_divisor, stream, out_stride, ws_len)));
    }
    else
    {
        CFFT_DISPATCH_REAL4 (r1, e = (real4_pass_b_inv<N_> (r2, in, work, tw1, big_tw, batch, grid_divisor, stream, in_stride, ws_len)));
        if (e != hipSuccess)
            return e;
        CFFT_DISPATCH_REAL4 (r2, e = (real4_pass_a_inv<N_> (n1c, work, out, tw2, batch, grid_divisor, stream, out_stride, ws_len)));
    }
    return e;
}

// ---- both passes in one launch, hand-off inside an XCD's L2 (fused_twopass_kernels.hpp) -------------------------------
size_t fused_state_bytes() { return sizeof (FusedState); }
size_t fused_workspace_bytes (int nc, int slots) { return (size_t) kFusedMaxXcc * (size_t) slots * (size_t) nc * sizeof (float2); }
bool fused_supported (int n1, int n2) { return n1 == 256 && n2 == 256; }

hipError_t launch_fused_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, int lag, int slots, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (! fused_supported (n1, n2) || batch >= 0x7FFFFFFFull || lag < 1 || slots <= lag || slots > kFusedMaxSlots)
        return hipErrorInvalidValue;
#if ! CFFT_EXPERIMENTS
    (void) in; (void) work; (void) out; (void) stage_tw; (void) big_tw; (void) state; (void) host_err; (void) seq; (void) stream; (void) in_stride; (void) out_stride;
    return hipErrorNotSupported; // the ticket-queue form is an EXPERIMENTS=1 build's (launch.hpp)
#else
    const size_t frame_len = (size_t) n1 * (size_t) n2;
    in_stride = in_stride ? in_stride : frame_len;
    out_stride = out_stride ? out_stride : frame_len;
    // queue state: tickets, claims and counters start from zero in every launch
    hipError_t e = hipMemsetAsync (state, 0, sizeof (FusedState), stream);
    if (e != hipSuccess)
        return e;
    constexpr size_t lds = (size_t) 16 * TileFrame<256>::stride * sizeof (float2);
    auto k = inverse ? fused256_kernel<true> : fused256_kernel<false>;
    // every tile of both passes is a work item; the grid never exceeds what is resident, so every ticket holder runs
    unsigned grid = persistent_grid (k, 256, lds, batch * 32);
    if (grid == 0)
        return hipErrorNoDevice;
    // two workgroups per CU: more of them lengthen the queue (tickets in flight) and with it the time an intermediate has
    // to survive in L2, for no gain in rate (measured; CHOWDSP_FFT_FUSED_WG_PER_CU overrides)
    static const int grid_cap = [] {
        const char* v = std::getenv ("CHOWDSP_FFT_FUSED_WG_PER_CU");
        return v != nullptr ? std::atoi (v) : 2;
    }();
    static const int dbg_mode = [] {
        const char* v = std::getenv ("CHOWDSP_FFT_FUSED_EXPERIMENT");
        return v != nullptr ? std::atoi (v) : 0;
    }();
    if (grid_cap > 0)
    {
        DeviceInfo dev;
        if (current_device_info (dev))
            grid = std::min (grid, (unsigned) (dev.num_cus * grid_cap));
    }
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (256), lds, stream, in, work, out, stage_tw, big_tw, (FusedState*) state, host_err, seq, (unsigned) batch, in_stride, out_stride, (unsigned) lag, (unsigned) slots, (unsigned) dbg_mode);
    e = hipGetLastError();
    static const bool debug = std::getenv ("CHOWDSP_FFT_FUSED_DEBUG") != nullptr;
    if (debug && e == hipSuccess)
    {
        // developer aid: wait for the launch and show how the work spread over the XCD queues
        FusedState h;
        if (hipStreamSynchronize (stream) == hipSuccess && hipMemcpy (&h, state, sizeof (h), hipMemcpyDeviceToHost) == hipSuccess)
        {
            fprintf (stderr, "[fused] grid %u batch %zu lag %d slots %d error %u claimed %u stats B-prefetched %u B-late %u unresolved %u polls %u A-slot-checks %u tickets:",
                     grid, batch, lag, slots, h.error, h.next_frame, h.stat[0], h.stat[1], h.stat[2], h.stat[3], h.stat[4]);
            for (int x = 0; x < kFusedMaxXcc; ++x)
                if (h.q[x].ticket != 0)
                    fprintf (stderr, " xcc%d=%u", x, h.q[x].ticket);
            fprintf (stderr, "\n");
        }
    }
    return e;
#endif
}

// ---- the same, by static groups of workgroups per XCD (group_twopass_kernels.hpp) ---------------------------------------
namespace
{
// workgroups per CU in tenths (CHOWDSP_FFT_GROUP_WG_PER_CU, e.g. 2 or 2.5): 16 of them per XCD are one group, and every group
// needs 512 KiB of that XCD's L2 for its slot
int group_wg_per_cu_x10()
{
    static const int v = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_GROUP_WG_PER_CU");
        const int n = e != nullptr ? (int) (std::atof (e) * 10.0 + 0.5) : 20;
        return (n >= 5 && n <= 40) ? n : 20;
    }();
    return v;
}
constexpr size_t kGroupLds = (size_t) 16 * TileFrame<256>::stride * sizeof (float2);
// options "group_force_solo" (tests: every launch takes the fall-back mode) and "group_reg_polls" (how long a workgroup waits
// for the grid to be complete before it decides for the fall-back mode; one poll is about a microsecond)
std::atomic<int> g_group_force_solo { 0 }, g_group_reg_polls { 200 };
// developer option "group_spin_limit": polls after which a hand-off wait gives up (0 = the built-in limit, a fraction of a
// second).  Tests set 1: the waits DO run out, the launch reports it, the output is invalid -- the failure path end to end.
std::atomic<int> g_group_spin_limit { 0 };
// option "group_handoff": 0 (default) the fast hand-off (plain payload and flag stores, two workgroups per CU), 1 the form the CDNA4
// guide lists (sc1 payload and flag stores, ONE workgroup per CU: fused_twopass_kernels.hpp flag_store_as).  One workgroup per CU is
// enforced, not hoped for: the launch asks for more than half of the CU's LDS.
std::atomic<int> g_group_handoff { 0 };
unsigned group_grid (bool inverse)
{
    auto k = inverse ? group256_kernel<true> : group256_kernel<false>;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return 0;
    const bool listed = g_group_handoff.load() != 0;
    const int per_cu_x10 = std::min (listed ? 10 : group_wg_per_cu_x10(), 10 * blocks_per_cu_cached ((const void*) k, 256, group_lds_bytes (kGroupLds)));
    return (unsigned) (dev.num_cus * std::max (per_cu_x10, 0) / 10);
}
} // namespace
void set_group_handoff (int listed) { g_group_handoff = listed != 0 ? 1 : 0; }
int get_group_handoff() { return g_group_handoff.load(); }
size_t group_lds_bytes (size_t kernel_lds)
{
    // the listed form: more than half of the CU's 160 KiB, so that a second workgroup cannot become resident on the CU
    return g_group_handoff.load() != 0 ? std::max<size_t> (kernel_lds, (size_t) 84 * 1024) : kernel_lds;
}
void set_group_options (int force_solo, int reg_polls)
{
    if (force_solo >= 0)
        g_group_force_solo = force_solo != 0 ? 1 : 0;
    if (reg_polls >= 0)
        g_group_reg_polls = reg_polls;
}
void set_group_spin_limit (int polls) { g_group_spin_limit = (polls > 0 && polls < 65536) ? polls : 0; }
int get_group_spin_limit() { return g_group_spin_limit.load(); }
void get_group_options (int& force_solo, int& reg_polls)
{
    // (the kernels' `force_solo` argument: bit 0 the mode, bit 1 the listed hand-off form, bits 16.. the spin limit -- group_twopass_kernels.hpp group_spin_limit)
    force_solo = g_group_force_solo.load() | (g_group_handoff.load() != 0 ? 2 : 0) | (g_group_spin_limit.load() << 16);
    reg_polls = g_group_reg_polls.load();
}
int group_wg_per_cu_tenths() { return group_wg_per_cu_x10(); }
size_t group_state_bytes() { return sizeof (GroupState); }
size_t group_workspace_bytes()
{
    // GROUP mode addresses kGroupSlotsTotal slots; SOLO mode one per workgroup of the grid (the fused convolution: two)
    const unsigned grid = std::max (std::max (std::max (group_grid (false), group_grid (true)), 2u * group_conv_grid()), std::max (group_real_grid (false), group_real_grid (true)));
    return (size_t) std::max<unsigned> (kGroupSlotsTotal, grid) * 65536u * sizeof (float2);
}
hipError_t launch_group_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (n1 != 256 || n2 != 256 || batch >= 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    in_stride = in_stride ? in_stride : 65536;
    out_stride = out_stride ? out_stride : 65536;
    hipError_t e = hipMemsetAsync (state, 0, sizeof (GroupState), stream); // ranks, mode and flags start from zero in every launch
    if (e != hipSuccess)
        return e;
    auto k = inverse ? group256_kernel<true> : group256_kernel<false>;
    // the WHOLE resident grid, whatever the batch: groups form only when every workgroup of the grid has registered
    const unsigned grid = group_grid (inverse);
    if (grid == 0)
        return hipErrorNoDevice;
    // A/B switch: two slots per group used alternately (no second flag, twice the L2 footprint) instead of one
    // ... and where the next frame's operands are requested (group_twopass_kernels.hpp): CHOWDSP_FFT_GROUP_PREFETCH = 0 behind the
    // pass-B loads, 1 in front of the hand-off, 2 the same with wave 0's share behind the pass-B loads, 3 a quarter at a time
    // between the steps of pass B.  Unset: 3 for the forward transform (2.25 against 2.31 ms per 8192 frames), 0 for the
    // backward one (whose re <-> im swaps schedule differently: 2.46 against 2.69 ms with 3; profiles/r03_group_two_pass.txt)
    static const int pf_env = std::getenv ("CHOWDSP_FFT_GROUP_PREFETCH") != nullptr ? std::atoi (std::getenv ("CHOWDSP_FFT_GROUP_PREFETCH")) : -1;
    static const unsigned slots_flag = (std::getenv ("CHOWDSP_FFT_GROUP_SLOTS") != nullptr && std::atoi (std::getenv ("CHOWDSP_FFT_GROUP_SLOTS")) == 2) ? 1u : 0u;
    const int pf = pf_env >= 0 ? pf_env : 3; // (round 4: the backward instantiation no longer waits for its prefetch at the request, so it takes the spread form too)
    // round 5 A/B: CHOWDSP_FFT_GROUP_INCR = 1: the hand-off member by member (every wave requests a producer's block as that producer's flag comes up)
    static const unsigned incr_flag = (std::getenv ("CHOWDSP_FFT_GROUP_INCR") != nullptr && std::atoi (std::getenv ("CHOWDSP_FFT_GROUP_INCR")) != 0) ? 16u : 0u;
    const unsigned two_slots = slots_flag | incr_flag | (pf == 1 ? 2u : (pf == 2 ? 6u : (pf == 3 ? 8u : 0u)));
    unsigned force_solo = (unsigned) g_group_force_solo.load() | (g_group_handoff.load() != 0 ? 2u : 0u) | ((unsigned) g_group_spin_limit.load() << 16);
    const unsigned reg_polls = (unsigned) g_group_reg_polls.load();
#if CFFT_GROUP_EXPERIMENT
    static const unsigned dbg = std::getenv ("CHOWDSP_FFT_GROUP_DBG") != nullptr ? (unsigned) std::atoi (std::getenv ("CHOWDSP_FFT_GROUP_DBG")) : 0u;
    force_solo |= dbg << 8;
#endif
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (256), group_lds_bytes (kGroupLds), stream, in, work, out, stage_tw, big_tw, (GroupState*) state, host_err, (unsigned) batch, in_stride, out_stride, seq, reg_polls, force_solo, two_slots);
    e = hipGetLastError();
    static const bool debug = std::getenv ("CHOWDSP_FFT_FUSED_DEBUG") != nullptr;
    if (debug && e == hipSuccess)
    {
        GroupState h;
        if (hipStreamSynchronize (stream) == hipSuccess && hipMemcpy (&h, state, sizeof (h), hipMemcpyDeviceToHost) == hipSuccess)
        {
            fprintf (stderr, "[group] grid %u batch %zu registered %u mode %u error %u solo-claimed %u ranks:", grid, batch, h.registered, h.mode, h.error, h.next_frame);
            for (int x = 0; x < kGroupMaxXcc; ++x)
                if (h.xcc[x].count != 0)
                    fprintf (stderr, " xcc%d=%u", x, h.xcc[x].count);
            fprintf (stderr, "\n");
#if CFFT_GROUP_EXPERIMENT
            // phase boundaries of eight iterations of group 0's member 0, in shader cycles since the iteration's start:
            // 1 operands here, 2 pass A done, 3 slot stores issued, 4 ... acknowledged, 5 flag stored, 6 group complete (wave 0),
            // 7 barrier, 8 pass-B loads + prefetch issued, 9 pass-B operands here, 10 pass B done, 11 output stores issued
            for (int it = 0; it < 8; ++it)
            {
                if (h.stamps[it][0] == 0)
                    continue;
                fprintf (stderr, "[group] stamps it %d:", it);
                for (int i = 1; i < 12; ++i)
                    fprintf (stderr, " %lld", (long long) (h.stamps[it][i] - h.stamps[it][0]));
                if (it > 0)
                    fprintf (stderr, "   (iteration %lld)", (long long) (h.stamps[it][0] - h.stamps[it - 1][0]));
                fprintf (stderr, "\n");
            }
#endif
        }
    }
    return e;
}

// ---- ... and for the register-direct tiles: N = 512 x 256 (2^17) and 512 x 512 (2^18), group_fast_kernels.hpp ----------------
bool groupfast_supported (int n1, int n2) { return n1 == 512 && (n2 == 256 || n2 == 512); }
namespace
{
template <int N1, int N2>
hipError_t groupfast_launch (bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    using GF = GroupFast<N1, N2>;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    auto k = inverse ? groupfast_kernel<N1, N2, true> : groupfast_kernel<N1, N2, false>;
    if (blocks_per_cu_cached ((const void*) k, 256, GF::lds_bytes) < 1)
        return hipErrorNoDevice;
    const unsigned grid = (unsigned) dev.num_cus; // ONE workgroup per CU: 32 per XCD = two groups of 16 / one of 32, whose slots fit that XCD's L2
    const float2* tw_a = stage_tw;                                  // n2 core (columns)
    const float2* tw_b = stage_tw + fast_tile_table_rows (N2) * 32; // n1 core (rows)
    int force_solo = 0, reg_polls = 200;
    get_group_options (force_solo, reg_polls);
    // where the next frame's operands are requested (group_fast_kernels.hpp `form`).  Round 4, after the backward kernel stopped
    // waiting for its prefetch at the request: 2^17 (two groups of 16 per XCD) 0.398 / 0.390 / 0.414 of 8 TB/s with 0 / 1 / 2,
    // 2^18 (one group of 32) 0.385 / 0.371 / 0.371 -- in front of the hand-off for 2^17, behind the pass-B loads for 2^18
    static const int form_env = std::getenv ("CHOWDSP_FFT_GROUPFAST_PREFETCH") != nullptr ? std::atoi (std::getenv ("CHOWDSP_FFT_GROUPFAST_PREFETCH")) : -1;
    const unsigned form = form_env >= 0 ? (unsigned) form_env : (N2 == 256 ? 2u : 0u);
    hipLaunchKernelGGL (k, dim3 (grid), dim3 (256), GF::lds_bytes, stream, in, work, out, tw_a, tw_b, big_tw, (GroupState*) state, host_err, (unsigned) batch, in_stride, out_stride, seq, (unsigned) reg_polls, (unsigned) force_solo, form);
    return hipGetLastError();
}
} // namespace
size_t groupfast_workspace_bytes (int n1, int n2)
{
    DeviceInfo dev;
    const unsigned grid = current_device_info (dev) ? (unsigned) dev.num_cus : 256u;
    // GROUP mode: one slot per group, addressed [xcc][group]; SOLO mode: one per workgroup
    return (size_t) std::max<unsigned> ((unsigned) (kGroupMaxXcc * kGroupMaxPerXcc), grid) * (size_t) n1 * (size_t) n2 * sizeof (float2);
}
hipError_t launch_groupfast_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    if (! groupfast_supported (n1, n2) || batch >= 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    const size_t frame_len = (size_t) n1 * (size_t) n2;
    in_stride = in_stride ? in_stride : frame_len;
    out_stride = out_stride ? out_stride : frame_len;
    const hipError_t e = hipMemsetAsync (state, 0, sizeof (GroupState), stream);
    if (e != hipSuccess)
        return e;
    return n2 == 256 ? groupfast_launch<512, 256> (inverse, in, work, out, stage_tw, big_tw, state, host_err, seq, batch, stream, in_stride, out_stride)
                     : groupfast_launch<512, 512> (inverse, in, work, out, stage_tw, big_tw, state, host_err, seq, batch, stream, in_stride, out_stride);
}

// long mixed-radix frames: generic n2-point column pass (generic_kernels.hpp), then the ordinary n1-point row pass
namespace
{
template <int C, int BLOCK>
PassAGenericFn pick_pass_a_cb (bool inverse, bool pipe)
{
    return inverse ? (pipe ? pass_a_generic_kernel<true, C, BLOCK, true> : pass_a_generic_kernel<true, C, BLOCK, false>)
                   : (pipe ? pass_a_generic_kernel<false, C, BLOCK, true> : pass_a_generic_kernel<false, C, BLOCK, false>);
}
template <int C, int BLOCK>
PassBGenericFn pick_pass_b_cb (bool inverse, bool pipe)
{
    return inverse ? (pipe ? pass_b_generic_kernel<true, C, BLOCK, true> : pass_b_generic_kernel<true, C, BLOCK, false>)
                   : (pipe ? pass_b_generic_kernel<false, C, BLOCK, true> : pass_b_generic_kernel<false, C, BLOCK, false>);
}
PassAGenericFn pick_pass_a (int c, int block, bool inverse, bool pipe)
{
    if (block == 1024)
        return c == 16 ? pick_pass_a_cb<16, 1024> (inverse, pipe) : (c == 8 ? pick_pass_a_cb<8, 1024> (inverse, pipe) : pick_pass_a_cb<4, 1024> (inverse, pipe));
    return c == 16 ? pick_pass_a_cb<16, 256> (inverse, pipe) : (c == 8 ? pick_pass_a_cb<8, 256> (inverse, pipe) : pick_pass_a_cb<4, 256> (inverse, pipe));
}
PassBGenericFn pick_pass_b (int c, int block, bool inverse, bool pipe)
{
    if (block == 1024)
        return c == 16 ? pick_pass_b_cb<16, 1024> (inverse, pipe) : (c == 8 ? pick_pass_b_cb<8, 1024> (inverse, pipe) : pick_pass_b_cb<4, 1024> (inverse, pipe));
    return c == 16 ? pick_pass_b_cb<16, 256> (inverse, pipe) : (c == 8 ? pick_pass_b_cb<8, 256> (inverse, pipe) : pick_pass_b_cb<4, 256> (inverse, pipe));
}
} // namespace

hipError_t launch_mixed_big_c2c (const DeviceTables& t, bool inverse, const float2* in, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    const int n1 = t.mixed_n1, n2 = t.mixed_n2;
    if (batch > 0x7FFFFFFFull || n1 < 2 || n2 < 2 || t.gen_ns < 1 || t.gen_ns > kGenericMaxStages)
        return hipErrorInvalidValue;
    const size_t frame_len = (size_t) n1 * (size_t) n2;
    in_stride = in_stride ? in_stride : frame_len;
    out_stride = out_stride ? out_stride : frame_len;
    GenericPlan plan {};
    plan.n = n2;
    plan.ns = t.gen_ns;
    const int ca = t.mixed_ca;
    if (ca != 16 && ca != 8 && ca != 4)
        return hipErrorInvalidValue;
    plan.frames = ca;
    plan.in_stride = plan.out_stride = (unsigned) n2;
    for (int i = 0; i < t.gen_ns; ++i)
    {
        plan.radix[i] = t.gen_radix[i];
        plan.tw_offset[i] = t.gen_tw_offset[i];
    }
    // two images of ca frames; a frame is padded by one point (bank spread while the tile is loaded / stored) unless that
    // pushes the images past the CU's LDS (n2 = 1280 with 8 columns: 163968 B > 160 KiB)
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const size_t ws_frame_len = (size_t) ((n1 + ca - 1) / ca) * (size_t) ca * (size_t) n2; // = n1 * n2 unless the last tile is ragged
    // a compile-time stage list for this column length (kernels_tile_static_*.hip) brings its own geometry; else the run-time list
    unsigned lfs = 0;
    size_t lds = 0;
    int block_a = 0;
    StaticTileLaunch sl {};
    PassAGenericFn ka = pick_static_pass_a (plan, ca, dev.lds_per_cu, inverse, sl);
    if (ka != nullptr)
    {
        lfs = sl.lfs;
        lds = sl.lds_bytes;
        block_a = sl.block;
    }
    else
    {
        lfs = (unsigned) n2 + 1u;
        if ((size_t) 2 * ca * lfs * sizeof (float2) > dev.lds_per_cu)
            lfs = (unsigned) n2;
        lds = (size_t) 2 * ca * lfs * sizeof (float2);
        if (lds > dev.lds_per_cu)
            return hipErrorInvalidValue;
        block_a = generic_tile_block (lds, dev.lds_per_cu);
        const bool pipe_a = tile_pipe_fits (lds, (size_t) n2, dev.lds_per_cu, 1, block_a);
        if (pipe_a)
            lds += (size_t) (n2 - 1) * sizeof (float2);
        ka = pick_pass_a (ca, block_a, inverse, pipe_a);
    }
    unsigned grid = persistent_grid (ka, block_a, lds, batch * (size_t) ((n1 + ca - 1) / ca));
    if (grid == 0)
        return hipErrorNoDevice;
    grid = std::max (1u, grid / (unsigned) grid_divisor);
    hipLaunchKernelGGL (ka, dim3 (grid), dim3 (block_a), lds, stream, in, work, (const float2*) t.stage_twiddles, (const float2*) t.big_twiddles, plan, (unsigned) n1, (unsigned) batch, in_stride, lfs, ws_frame_len);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    if (t.mixed_cb > 0)
    {
        // general form: the n1-point rows by the generic kernel as well
        const int cb = t.mixed_cb;
        if (t.gen2_ns < 1 || t.gen2_ns > kGenericMaxStages || (cb != 16 && cb != 8 && cb != 4))
            return hipErrorInvalidValue;
        GenericPlan pb {};
        pb.n = n1;
        pb.ns = t.gen2_ns;
        pb.frames = cb;
        pb.in_stride = pb.out_stride = (unsigned) n1;
        for (int i = 0; i < t.gen2_ns; ++i)
        {
            pb.radix[i] = t.gen2_radix[i];
            pb.tw_offset[i] = t.gen2_tw_offset[i];
        }
        unsigned lfsb = 0;
        size_t ldsb = 0;
        int block_b = 0;
        StaticTileLaunch slb {};
        PassBGenericFn kb = pick_static_pass_b (pb, cb, dev.lds_per_cu, inverse, slb);
        if (kb != nullptr)
        {
            lfsb = slb.lfs;
            ldsb = slb.lds_bytes;
            block_b = slb.block;
        }
        else
        {
            lfsb = (unsigned) n1 + 1u;
            if ((size_t) 2 * cb * lfsb * sizeof (float2) > dev.lds_per_cu)
                lfsb = (unsigned) n1;
            ldsb = (size_t) 2 * cb * lfsb * sizeof (float2);
            if (ldsb > dev.lds_per_cu)
                return hipErrorInvalidValue;
            block_b = generic_tile_block (ldsb, dev.lds_per_cu);
            const bool pipe_b = tile_pipe_fits (ldsb, (size_t) n1, dev.lds_per_cu, 2, block_b);
            if (pipe_b)
                ldsb += (size_t) (n1 - 1) * sizeof (float2);
            kb = pick_pass_b (cb, block_b, inverse, pipe_b);
        }
        unsigned gridb = persistent_grid (kb, block_b, ldsb, batch * (size_t) ((n2 + cb - 1) / cb));
        if (gridb == 0)
            return hipErrorNoDevice;
        gridb = std::max (1u, gridb / (unsigned) grid_divisor);
        hipLaunchKernelGGL (kb, dim3 (gridb), dim3 (block_b), ldsb, stream, (const float2*) work, out, (const float2*) t.stage_twiddles, pb, (unsigned) n2, (unsigned) ca, (unsigned) batch, out_stride, lfsb, ws_frame_len);
        return hipGetLastError();
    }
    const float2* tw1 = t.stage_twiddles + t.mixed_tw1_offset;
    e = hipErrorInvalidValue;
    if (inverse)
    {
        CFFT_DISPATCH_BIG (n1, e = (pass_b<N_, true> (n2, work, out, tw1, batch, grid_divisor, stream, out_stride, ca)));
    }
    else
    {
        CFFT_DISPATCH_BIG (n1, e = (pass_b<N_, false> (n2, work, out, tw1, batch, grid_divisor, stream, out_stride, ca)));
    }
    return e;
}

hipError_t launch_real_pair (bool forward, const float2* in, float2* out, const float2* real_tw, unsigned m, size_t batch, hipStream_t stream, size_t in_stride, size_t out_stride)
{
    if (batch == 0)
        return hipSuccess;
    in_stride = in_stride ? in_stride : (size_t) m;
    out_stride = out_stride ? out_stride : (size_t) m;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return hipErrorNoDevice;
    const size_t total = batch * ((size_t) (m / 4 + 255) / 256); // (two adjacent pairs per work-item)
    const unsigned grid = (unsigned) std::max<size_t> (1, std::min<size_t> (total, (size_t) dev.num_cus * 16));
    if (forward)
        hipLaunchKernelGGL (real_pair_kernel<true>, dim3 (grid), dim3 (256), 0, stream, in, out, real_tw, m, (unsigned) batch, in_stride, out_stride);
    else
        hipLaunchKernelGGL (real_pair_kernel<false>, dim3 (grid), dim3 (256), 0, stream, in, out, real_tw, m, (unsigned) batch, in_stride, out_stride);
    return hipGetLastError();
}
} // namespace cfft::host
