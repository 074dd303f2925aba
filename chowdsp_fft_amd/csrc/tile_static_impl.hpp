// Body of kernels_tile_static_{a,b,c}.hip: the tile kernels with compile-time stage lists for ONE group of lengths.
// The including file defines CFFT_TILE_GROUP (the list macro of tables.hpp) and CFFT_TILE_PICK_A / CFFT_TILE_PICK_B / CFFT_TILE_PICK_COL (the
// names of the three functions it provides; tile_static.hpp declares them).
#include "tile_static.hpp"
#include "launch_util.hpp"
#include "tables.hpp"

namespace cfft::host
{
namespace
{
// one (tile width, workgroup size) per length: the pair the launchers choose on MI355X; anything else (another LDS size, a planner
// list that differs from the template's) runs the run-time list
template <int N_>
struct StaticTile
{
    // from kTileInplaceMinN on: one LDS image, stages in place (generic_kernels.hpp: tile_inplace_stage), always 16 columns
    static constexpr bool inplace = N_ >= kTileInplaceMinN;
    static constexpr int c = inplace ? (N_ <= kTileInplaceWideMaxN ? 16 : 8) : (N_ <= 639 ? 16 : 8);
    static constexpr size_t images = (size_t) (inplace ? 1 : 2) * c * (N_ + 1) * sizeof (float2);
    static constexpr size_t lds = images + (size_t) (N_ - 1) * sizeof (float2); // + the stage twiddles' copy
    // work-items per workgroup: 256 where four workgroups fit a CU, 1024 where only one does, and 512 between (two or three
    // workgroups of 8 waves each work through their phases independently; measured against 1024 / 256 on MI355X with two
    // images, fraction of 8 TB/s: c23040 0.355 / 0.305 / 0.353, c57600 0.336 / 0.305 / 0.284, c82944 0.317 / 0.287 / 0.220,
    // c61440 0.321 / 0.298 / 0.283, c245760 0.285 / 0.276 / 0.257)
#ifndef CFFT_STATIC_TILE_MID_BLOCK
#define CFFT_STATIC_TILE_MID_BLOCK 512
#endif
    static constexpr int block = 4 * images <= 160 * 1024 ? 256 : (2 * lds <= 160 * 1024 ? CFFT_STATIC_TILE_MID_BLOCK : 1024);
};
template <int N_>
bool static_tile_matches (const GenericPlan& plan, int c, size_t lds_per_cu)
{
    using SP = typename StaticTilePlanFor<N_>::type;
    bool same = plan.n == N_ && plan.ns == SP::ns && c == StaticTile<N_>::c && StaticTile<N_>::lds <= lds_per_cu;
    for (int i = 0; same && i < SP::ns; ++i)
        same = plan.radix[i] == SP::radix (i);
    return same;
}
// (the lengths: CFFT_STATIC_TILE_SIZES in tables.hpp -- every length of StaticPlanFor up to 1200; n = 1280 runs unpadded: run-time list)
} // namespace

PassAGenericFn CFFT_TILE_PICK_A (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch)
{
#define CFFT_TRY(N_)                                                                                                                  \
    if (static_tile_matches<N_> (plan, c, lds_per_cu))                                                                                \
    {                                                                                                                                 \
        launch = StaticTileLaunch { StaticTile<N_>::block, (unsigned) N_ + 1u, StaticTile<N_>::lds };                                 \
        return inverse ? pass_a_generic_kernel<true, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticTilePlanFor<N_>::type, StaticTile<N_>::inplace> \
                       : pass_a_generic_kernel<false, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticTilePlanFor<N_>::type, StaticTile<N_>::inplace>; \
    }
    CFFT_TILE_GROUP (CFFT_TRY)
#undef CFFT_TRY
    return nullptr;
}
PassBGenericFn CFFT_TILE_PICK_B (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch)
{
#define CFFT_TRY(N_)                                                                                                                  \
    if (static_tile_matches<N_> (plan, c, lds_per_cu))                                                                                \
    {                                                                                                                                 \
        launch = StaticTileLaunch { StaticTile<N_>::block, (unsigned) N_ + 1u, StaticTile<N_>::lds };                                 \
        return inverse ? pass_b_generic_kernel<true, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticTilePlanFor<N_>::type, StaticTile<N_>::inplace> \
                       : pass_b_generic_kernel<false, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticTilePlanFor<N_>::type, StaticTile<N_>::inplace>; \
    }
    CFFT_TILE_GROUP (CFFT_TRY)
#undef CFFT_TRY
    return nullptr;
}
ColumnFftFn CFFT_TILE_PICK_COL (const GenericPlan& plan, int c, size_t lds_per_cu, bool inverse, StaticTileLaunch& launch)
{
#define CFFT_TRY(N_)                                                                                                                  \
    if (static_tile_matches<N_> (plan, c, lds_per_cu))                                                                                \
    {                                                                                                                                 \
        launch = StaticTileLaunch { StaticTile<N_>::block, (unsigned) N_ + 1u, StaticTile<N_>::lds };                                 \
        return inverse ? column_fft_kernel<true, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticTilePlanFor<N_>::type, StaticTile<N_>::inplace> \
                       : column_fft_kernel<false, StaticTile<N_>::c, StaticTile<N_>::block, true, typename StaticTilePlanFor<N_>::type, StaticTile<N_>::inplace>; \
    }
    CFFT_TILE_GROUP (CFFT_TRY)
#undef CFFT_TRY
    return nullptr;
}
} // namespace cfft::host
