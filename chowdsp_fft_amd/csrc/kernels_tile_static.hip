// Instantiation of the tile kernels with compile-time stage lists (tile_static.hpp).
#include "tile_static.hpp"
#include "launch_util.hpp"

#include <cstdlib>

namespace cfft::host
{
namespace
{
// compile-time stage lists for the tile kernels (generic_kernels.hpp: SP): one (tile width, workgroup size) per length, the pair
// the launchers below choose on MI355X; anything else (another LDS size, a forced block, a planner list that differs from the
// template's) runs the run-time list
template <int N_>
struct StaticTile
{
    static constexpr int c = N_ <= 639 ? 16 : 8;
    static constexpr size_t lds = (size_t) 2 * c * (N_ + 1) * sizeof (float2);
    static constexpr int block = 4 * lds > 160 * 1024 ? 1024 : 256;
};
template <int N_>
bool static_tile_matches (const GenericPlan& plan, int c, int block, unsigned lfs)
{
    using SP = typename StaticPlanFor<N_>::type;
    bool same = plan.n == N_ && plan.ns == SP::ns && c == StaticTile<N_>::c && block == StaticTile<N_>::block && lfs == (unsigned) N_ + 1u;
    for (int i = 0; same && i < SP::ns; ++i)
        same = plan.radix[i] == SP::radix (i);
    return same;
}
// every length of StaticPlanFor whose padded images fit the CU's LDS (n = 1280 runs unpadded: run-time list)
#define CFFT_STATIC_TILE_SIZES(X) \
    X (48) X (80) X (96) X (144) X (160) X (192) X (240) X (288) X (320) X (384) X (400) X (432) X (480) X (576) X (640) X (720) \
    X (768) X (800) X (864) X (960) X (1152) X (1200)
} // namespace

PassAGenericFn pick_static_pass_a (const GenericPlan& plan, int c, int block, unsigned lfs, bool inverse)
{
    static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
    if (off)
        return nullptr;
#define CFFT_TRY(N_)                                                                                                                  \
    if (static_tile_matches<N_> (plan, c, block, lfs))                                                                                \
        return inverse ? pass_a_generic_kernel<true, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticPlanFor<N_>::type> \
                       : pass_a_generic_kernel<false, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticPlanFor<N_>::type>;
    CFFT_STATIC_TILE_SIZES (CFFT_TRY)
#undef CFFT_TRY
    return nullptr;
}
PassBGenericFn pick_static_pass_b (const GenericPlan& plan, int c, int block, unsigned lfs, bool inverse)
{
    static const bool off = [] { const char* e = std::getenv ("CHOWDSP_FFT_TILE_STATIC"); return e != nullptr && std::atoi (e) == 0; }();
    if (off)
        return nullptr;
#define CFFT_TRY(N_)                                                                                                                  \
    if (static_tile_matches<N_> (plan, c, block, lfs))                                                                                \
        return inverse ? pass_b_generic_kernel<true, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticPlanFor<N_>::type> \
                       : pass_b_generic_kernel<false, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticPlanFor<N_>::type>;
    CFFT_STATIC_TILE_SIZES (CFFT_TRY)
#undef CFFT_TRY
    return nullptr;
}
} // namespace cfft::host
