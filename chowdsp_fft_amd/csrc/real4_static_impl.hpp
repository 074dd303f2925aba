// Body of kernels_real4_static_{a,b,c}.hip: the real four-step's tile kernels with compile-time stage lists for ONE group of lengths.
// The including file defines CFFT_R4S_GROUP (a list macro of tables.hpp) and CFFT_R4S_SUFFIX (a / b / c).
#include "real4_static.hpp"
#include "launch_util.hpp"
#include "tables.hpp"

namespace cfft::host
{
namespace
{
template <int N_>
struct Real4Tile
{
    static constexpr bool cols = N_ % 2 == 0 && N_ >= 16 && N_ <= 639;  // column lengths R2: even
    static constexpr bool rows = N_ % 32 == 0 && N_ <= 639;             // row lengths R1: multiples of 32
    static constexpr bool cplx = N_ % 16 == 0 && N_ <= 639;             // either factor of the complex three-sweep convolution: multiples of 16
    static constexpr size_t images = (size_t) 2 * kReal4GTile * (N_ + 1) * sizeof (float2);
    static constexpr size_t lds = images + (size_t) (N_ - 1) * sizeof (float2);
    // work-items per workgroup as for the complex tiles (tile_static_impl.hpp): 256 where four workgroups fit a CU, 512 where two do, else 1024
    static constexpr int block = 4 * images <= 160 * 1024 ? 256 : (2 * lds <= 160 * 1024 ? 512 : 1024);
    static constexpr int slots_cols = (kReal4GTile * N_ + 32 + block - 1) / block; // (+ 32: the 16 (R2 / 2 + 1) float4 of the rows kept)
    static constexpr int slots_rows = (kReal4GTile * N_ + block - 1) / block;
    static constexpr int even (int v) { return v + (v & 1); }
};
template <int N_>
bool real4_tile_matches (const GenericPlan& plan, size_t lds_per_cu)
{
    using SP = typename StaticTilePlanFor<N_>::type;
    bool same = plan.n == N_ && plan.ns == SP::ns && Real4Tile<N_>::lds <= lds_per_cu;
    for (int i = 0; same && i < SP::ns; ++i)
        same = plan.radix[i] == SP::radix (i);
    return same;
}
} // namespace

#define CFFT_R4S_CAT2(A, B) A##B
#define CFFT_R4S_CAT(A, B) CFFT_R4S_CAT2 (A, B)
#define CFFT_R4S_PICK(KIND, TYPE, WHICH, KERNEL, SLOTS)                                                                               \
    TYPE CFFT_R4S_CAT (KIND, CFFT_R4S_SUFFIX) (const GenericPlan& plan, size_t lds_per_cu, Real4GLaunch& launch)                      \
    {                                                                                                                                 \
        (void) plan; (void) lds_per_cu; (void) launch;                                                                                \
        CFFT_R4S_GROUP (CFFT_R4S_TRY_##WHICH##_##KERNEL##_##SLOTS)                                                                    \
        return nullptr;                                                                                                               \
    }
#define CFFT_R4S_TRY_BODY(N_, WHICH, KERNEL, SLOTS)                                                                                   \
    if constexpr (Real4Tile<N_>::WHICH)                                                                                               \
    {                                                                                                                                 \
        if (real4_tile_matches<N_> (plan, lds_per_cu))                                                                                \
        {                                                                                                                             \
            launch = Real4GLaunch { Real4Tile<N_>::block, (unsigned) N_ + 1u, Real4Tile<N_>::lds };                                   \
            return KERNEL<Real4Tile<N_>::block, Real4Tile<N_>::even (Real4Tile<N_>::SLOTS), typename StaticTilePlanFor<N_>::type>;   \
        }                                                                                                                             \
    }
#define CFFT_R4S_TRY_cols_real4g_pass_a_kernel_slots_cols(N_) CFFT_R4S_TRY_BODY (N_, cols, real4g_pass_a_kernel, slots_cols)
#define CFFT_R4S_TRY_cols_real4g_pass_a_inv_kernel_slots_cols(N_) CFFT_R4S_TRY_BODY (N_, cols, real4g_pass_a_inv_kernel, slots_cols)
#define CFFT_R4S_TRY_rows_real4g_pass_b_kernel_slots_rows(N_) CFFT_R4S_TRY_BODY (N_, rows, real4g_pass_b_kernel, slots_rows)
#define CFFT_R4S_TRY_rows_real4g_pass_b_inv_kernel_slots_rows(N_) CFFT_R4S_TRY_BODY (N_, rows, real4g_pass_b_inv_kernel, slots_rows)
#define CFFT_R4S_TRY_rows_real4g_rows_conv_kernel_slots_rows(N_) CFFT_R4S_TRY_BODY (N_, rows, real4g_rows_conv_kernel, slots_rows)
#define CFFT_R4S_TRY_cplx_c4g_cols_fwd_kernel_slots_rows(N_) CFFT_R4S_TRY_BODY (N_, cplx, c4g_cols_fwd_kernel, slots_rows)
#define CFFT_R4S_TRY_cplx_c4g_rows_conv_kernel_slots_rows(N_) CFFT_R4S_TRY_BODY (N_, cplx, c4g_rows_conv_kernel, slots_rows)
#define CFFT_R4S_TRY_cplx_c4g_cols_inv_kernel_slots_rows(N_) CFFT_R4S_TRY_BODY (N_, cplx, c4g_cols_inv_kernel, slots_rows)

CFFT_R4S_PICK (pick_real4_cols_fwd_, Real4GTwFn, cols, real4g_pass_a_kernel, slots_cols)
CFFT_R4S_PICK (pick_real4_cols_inv_, Real4GFn, cols, real4g_pass_a_inv_kernel, slots_cols)
CFFT_R4S_PICK (pick_real4_rows_fwd_, Real4GFn, rows, real4g_pass_b_kernel, slots_rows)
CFFT_R4S_PICK (pick_real4_rows_inv_, Real4GTwFn, rows, real4g_pass_b_inv_kernel, slots_rows)
CFFT_R4S_PICK (pick_real4_rows_conv_, Real4GConvFn, rows, real4g_rows_conv_kernel, slots_rows)
CFFT_R4S_PICK (pick_conv4_cols_fwd_, Real4GTwFn, cplx, c4g_cols_fwd_kernel, slots_rows)
CFFT_R4S_PICK (pick_conv4_rows_conv_, Real4GConvFn, cplx, c4g_rows_conv_kernel, slots_rows)
CFFT_R4S_PICK (pick_conv4_cols_inv_, Real4GFn, cplx, c4g_cols_inv_kernel, slots_rows)
} // namespace cfft::host
