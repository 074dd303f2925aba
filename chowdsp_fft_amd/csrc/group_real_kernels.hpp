// REAL frames of N = 131072 samples in ONE launch on the static XCD groups of group_twopass_kernels.hpp: the 65536-point
// complex core (256 x 256, hand-off inside the XCD's L2) with the real pair stage done IN REGISTERS by the members of the pass
// that touches the packed spectrum -- pass B of the forward transform, pass A of the backward one.
//
// The pair stage combines bins k and M - k (M = 65536).  With k = kb + 256 ka (row kb, column ka of the spectrum as pass B
// produces it) the partner is row 256 - kb, column 255 - ka: a 16-row tile's partners straddle two other tiles.  So the rows
// are dealt differently here: member t of a group owns the EIGHT rows kb = 8 t + 1 + j and their eight mirrors 255 - 8 t - j
// (j < 8) -- 16 row transforms per member as before, every pair inside one workgroup.  The work-items of a mirrored row take
// their stage-1 butterflies in reversed order (index 15 - i), which puts bin M - k into the SAME register slot pattern,
// mirrored (slot 15 - b), of the lane 8 further on in the same 16-lane row: the partner arrives by one DPP row rotation per
// component, no LDS traffic, no barrier.  Rows 0 and 128 pair with themselves (columns (256 - ka) mod 256 and 255 - ka): they
// take the place of the one duplicate the dealing above would produce (row 128 is its own mirror) in member 15, which
// exchanges these two rows through 4 KiB of LDS behind one extra barrier.
// The price: the packed spectrum moves in runs of 64 bytes (8 rows) instead of 128, so a line is completed by two or three
// members of the same group -- they run in step on one XCD, the halves meet in its L2 (default cache policy on that side).
// The backward transform is the mirror image: pass A owns columns 8 t + 1 + j and 255 - 8 t - j of the packed spectrum, forms
// Z[k] from X[k] and X[M - k] at the point where the prefetched operands are consumed, and goes on as the complex kernel does.
// Pair twiddles: W_N^k = W_N^(kb + 256 i) * W_32^b for k = kb + 256 (i + 16 b): one launch-resident value per work-item
// (the plan's table, rounded once from double) times a compile-time constant.
//
// Replaces, per frame, rfftf1_ps / rfftb1_ps + pffft_real_finalize / pffft_real_preprocess
// (simd/chowdsp_fft_impl_sse.cpp:1080-1253, 1255-1384) of the reference for N = 131072 (test/test.cpp:288-295 runs real 2^17).
#pragma once
#include "group_twopass_kernels.hpp"

// the packed side's accesses (runs of 64 bytes, a line shared with a neighbour in the group): default cache policy (0) or nontemporal (1:
// measured 2.6 % slower, profiles/r04_group_real.txt)
#ifndef CFFT_GR_NT
#define CFFT_GR_NT 0
#endif
// scheduling fence after every CFFT_GR_FENCE pairs of the pair stage (0 = none): keeps the scheduler from forming all 16 twiddles
// and partners first
#ifndef CFFT_GR_FENCE
#define CFFT_GR_FENCE 4
#endif
namespace cfft
{
__device__ __forceinline__ void gr_packed_store (float2* p, float2 v)
{
    if (CFFT_GR_NT)
        gstore (p, v);
    else
        *p = v;
}
__device__ __forceinline__ float2 gr_packed_load (const float2* p) { return CFFT_GR_NT ? gload (p) : *p; }
__device__ __forceinline__ float dpp_row_ror8 (float x)
{
    return __int_as_float (__builtin_amdgcn_mov_dpp (__float_as_int (x), 0x128 /* row_ror:8 */, 0xF, 0xF, true));
}

// W_32^b = exp (-2 pi i b / 32), b < 16 (rounded from double)
static __device__ const float kW32re[16] = { 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f,
                                             0.0f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f };
static __device__ const float kW32im[16] = { -0.0f, -0.19509032201612825f, -0.38268343236508978f, -0.55557023301960218f, -0.70710678118654746f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f,
                                             -1.0f, -0.98078528040323043f, -0.92387953251128674f, -0.83146961230254535f, -0.70710678118654757f, -0.55557023301960218f, -0.38268343236508989f, -0.19509032201612861f };

// the strided index (row of pass B / column of pass A) that lane l (< 16) of member t owns, and whether it is one of the two
// self-paired ones (member 15: lane 7 = index 128, lane 15 = index 0)
__device__ __forceinline__ unsigned gr_index (unsigned t, int l)
{
    if (t == 15u && l == 15)
        return 0u;
    return l < 8 ? 8u * t + 1u + (unsigned) l : 263u - 8u * t - (unsigned) l;
}
// ... and whether its work-items take their butterfly / row index reversed (15 - m)
__device__ __forceinline__ bool gr_reversed (unsigned t, int l) { return l >= 8 && ! (t == 15u && l == 15); }

// X[k] from Z[k], Z[M - k] and W_N^k (forward); Z[k] from X[k], X[M - k] (backward, twice the true value like the reference)
__device__ __forceinline__ float2 gr_split (float2 A, float2 Zp, float2 wh) // wh = W_N^k / 2
{
    const float sx = A.x + Zp.x, sy = A.y - Zp.y; // A + conj Zp
    const float dx = A.x - Zp.x, dy = A.y + Zp.y; // A - conj Zp
    // X = S / 2 - i wh D
    return make_float2 (fmaf (0.5f, sx, fmaf (dx, wh.y, dy * wh.x)), fmaf (0.5f, sy, fmaf (dy, wh.y, -dx * wh.x)));
}
__device__ __forceinline__ float2 gr_merge (float2 A, float2 Xp, float2 w)
{
    const float sx = A.x + Xp.x, sy = A.y - Xp.y; // A + conj Xp
    const float dx = A.x - Xp.x, dy = A.y + Xp.y; // A - conj Xp
    // Z = S + i conj (w) D
    return make_float2 (sx - fmaf (dy, w.x, -dx * w.y), sy + fmaf (dx, w.x, dy * w.y));
}

// ---- the building blocks, for member / tile t of one frame (l = tid & 15: lane along the tile's 16 strided indices, m = tid >> 4) ----
// the pass-A results (times the inter-pass twiddles) into the slot, 16 x 16 block order
__device__ __forceinline__ void gr_slot_store (const float2 (&v)[16], const float2 (&bw)[16], float2* __restrict__ slot, size_t base, bool wt = false)
{
    if (wt)
    {
        // option "group_handoff" = 1: sc1 stores (written through), the listed hand-off form (fused_twopass_kernels.hpp flag_store_as)
#pragma unroll
        for (int b = 0; b < 16; ++b)
            slot_store_wt (&slot[base + b * 256], cmul (v[b], bw[b]));
        return;
    }
#pragma unroll
    for (int b = 0; b < 16; ++b)
        slot[base + b * 256] = cmul (v[b], bw[b]); // plain store: stays in this XCD's L2
}
// packed-side role of this work-item in tile t: W_N^(idx + 256 mr), idx + 256 mr < 4096
__device__ __forceinline__ float2 gr_pair_twiddle (const float2* __restrict__ rtw, unsigned t, int l, int m)
{
    const unsigned idx = gr_index (t, l);
    const unsigned mr = gr_reversed (t, l) ? 15u - (unsigned) m : (unsigned) m;
    return rtw[idx + 256u * mr];
}

// FORWARD pass B of tile t: rows gr_index (t, l) of the slot (already in v) -> Z -> pair stage -> packed spectrum of frame `dstf`
// MULH (fft_convolve_frames_batch): every packed bin times H[bin] * scaling on its way out (bin 0 = DC / Nyquist: two real products,
// reference simd/chowdsp_fft_impl_sse.cpp:1640-1644); the 16 filter bins of a work-item are requested behind the exchange, ahead of the
// last butterflies
#ifndef CFFT_GR_HGROUP
#define CFFT_GR_HGROUP 4
#endif
constexpr int kGrHGroup = CFFT_GR_HGROUP;
template <bool MULH, class Hook>
__device__ __forceinline__ void gr_fwd_pass_b (float2 (&v)[16], unsigned t, float2 wthr, float2* __restrict__ dstf, unsigned* bflag, unsigned bval, Hook hook,
                                               float2* __restrict__ smem, const float2* __restrict__ tw1, float2 (*spec)[256], int tid,
                                               const float2* __restrict__ H = nullptr, float scaling = 1.f, bool wt = false)
{
    const int l = tid & 15, m = tid >> 4;
    const unsigned kb = gr_index (t, l);
    const int mr = gr_reversed (t, l) ? 15 - m : m;
    float2 wloc = make_float2 (0.5f * wthr.x, 0.5f * wthr.y); // (opaque: the 16 products with W_32^b are formed per frame, not kept in 32 registers across the loop)
    asm volatile ("" : "+v"(wloc.x), "+v"(wloc.y));
    fft_reg<16> (v);
    hook (1);
    __syncthreads(); // the previous exchange's reads are finished; every wave's pass-B loads have returned
    if (tid == 0 && bflag != nullptr)
        flag_store_as (bflag, bval, wt);
#pragma unroll
    for (int b = 0; b < 16; ++b)
        smem[l * kFS256 + 17 * m + b] = v[b];
    hook (2);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
        v[a] = smem[l * kFS256 + mr + 17 * a];
    hook (3);
    // (a few filter bins at a time, one group ahead of its products: all 16 at once cost 156 bytes of scratch)
    float2 h[MULH ? 16 : 1];
    const float2* __restrict__ hp = MULH ? H + kb + 256u * (unsigned) mr : nullptr;
    if constexpr (MULH)
    {
#pragma unroll
        for (int b = 0; b < kGrHGroup; ++b)
            h[b] = hp[4096 * b];
    }
#pragma unroll
    for (int a = 1; a < 16; ++a)
        v[a] = cmul (v[a], tw1[(a - 1) * 16 + mr]);
    fft_reg<16> (v);
    // v[b] = Z[kb + 256 (mr + 16 b)]
    float2* __restrict__ dst = dstf + kb + 256u * (unsigned) mr;
    if (t != 15u)
    {
#pragma unroll
        for (int b = 0; b < 16; ++b)
        {
            if constexpr (MULH)
            {
                if (b % kGrHGroup == 0 && b + kGrHGroup < 16)
                {
                    // (an opaque copy of the pointer: the loads are requested HERE, not lifted to the top of the pass and held)
                    const float2* hq = hp;
                    asm volatile ("" : "+v"(hq));
#pragma unroll
                    for (int c = kGrHGroup; c < 2 * kGrHGroup; ++c)
                        h[b + c] = hq[4096 * (b + c)];
                    CFFT_PHASE_FENCE();
                }
            }
            const float2 zp = make_float2 (dpp_row_ror8 (v[15 - b].x), dpp_row_ror8 (v[15 - b].y));
            const float2 w = cmul (wloc, make_float2 (kW32re[b], kW32im[b]));
            float2 x = gr_split (v[b], zp, w);
            if constexpr (MULH)
                x = cmul (x, make_float2 (h[b].x * scaling, h[b].y * scaling));
            gr_packed_store (&dst[4096 * b], x);
            if (CFFT_GR_FENCE > 0 && (b % (CFFT_GR_FENCE > 0 ? CFFT_GR_FENCE : 1)) == CFFT_GR_FENCE - 1)
                CFFT_PHASE_FENCE();
        }
    }
    else
    {
        // (an opaque copy of the member index: the 2 x 16 LDS addresses of this branch are then formed here, per frame, by the one member that
        // needs them -- lifted out of the frame loop they were held in 20-odd registers by every member for the whole launch)
        int mrl = mr;
        CFFT_LAUNDER_INDEX (mrl);
        const bool self = (l == 7 || l == 15);
        if (self)
        {
#pragma unroll
            for (int b = 0; b < 16; ++b)
                spec[l == 15 ? 1 : 0][mrl + 16 * b] = v[b];
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; ++b)
        {
            if constexpr (MULH)
            {
                if (b % kGrHGroup == 0 && b + kGrHGroup < 16)
                {
                    // (an opaque copy of the pointer: the loads are requested HERE, not lifted to the top of the pass and held)
                    const float2* hq = hp;
                    asm volatile ("" : "+v"(hq));
#pragma unroll
                    for (int c = kGrHGroup; c < 2 * kGrHGroup; ++c)
                        h[b + c] = hq[4096 * (b + c)];
                    CFFT_PHASE_FENCE();
                }
            }
            float2 zp = make_float2 (dpp_row_ror8 (v[15 - b].x), dpp_row_ror8 (v[15 - b].y));
            const int ka = mrl + 16 * b;
            if (self)
                zp = (l == 7) ? spec[0][255 - ka] : spec[1][(256 - ka) & 255];
            const float2 w = cmul (wloc, make_float2 (kW32re[b], kW32im[b]));
            float2 x = gr_split (v[b], zp, w);
            if constexpr (MULH)
                x = cmul (x, make_float2 (h[b].x * scaling, h[b].y * scaling));
            if (b == 0 && l == 15 && m == 0)
            {
                x = make_float2 (v[0].x + v[0].y, v[0].x - v[0].y); // packed slot 0: X[0] and X[N/2]
                if constexpr (MULH)
                    x = make_float2 (x.x * h[0].x * scaling, x.y * h[0].y * scaling); // ... two real products
            }
            gr_packed_store (&dst[4096 * b], x);
        }
    }
}
// the slot address of W[a = 16 s + m][kb]: block (a_hi = s, kb_hi), element (a_lo = m, kb_lo)
__device__ __forceinline__ void gr_fwd_load_b (float2 (&v)[16], const float2* __restrict__ slot, unsigned t, int tid)
{
    const int l = tid & 15, m = tid >> 4;
    const unsigned kb = gr_index (t, l);
    const float2* __restrict__ src = slot + (kb >> 4) * 256u + (unsigned) m * 16u + (kb & 15u);
#pragma unroll
    for (int s = 0; s < 16; ++s)
        v[s] = gload_l2 (&src[s * 16 * 256]);
}

// BACKWARD: the packed spectrum's operands of tile t (columns gr_index (t, l), rows mr + 16 s), quarter G (4 = all of them)
template <int G>
__device__ __forceinline__ void gr_inv_load_a (float (&re)[16], float (&im)[16], const float2* __restrict__ srcf, unsigned t, int tid)
{
    const int l = tid & 15, m = tid >> 4;
    const unsigned col = gr_index (t, l);
    const unsigned mr = gr_reversed (t, l) ? 15u - (unsigned) m : (unsigned) m;
    const float2* __restrict__ src = srcf + col + 256u * mr;
#pragma unroll
    for (int s = (G == 4 ? 0 : 4 * G); s < (G == 4 ? 16 : 4 * G + 4); ++s)
    {
        const float2 x = gr_packed_load (&src[4096 * s]); // (default cache policy: the other half of the line belongs to a neighbour in this group)
        re[s] = x.x;
        im[s] = x.y;
    }
}
// ... pair stage (merge) + swap for the backward transform + complex pass A into the slot
__device__ __forceinline__ void gr_inv_pass_a (const float (&re)[16], const float (&im)[16], unsigned t, float2 wthr, const float2 (&bw)[16], float2* __restrict__ slot, size_t base,
                                               float2* __restrict__ smem, const float2* __restrict__ tw1, float2 (*spec)[256], int tid, bool wt = false)
{
    const int l = tid & 15, m = tid >> 4;
    const int mr = gr_reversed (t, l) ? 15 - m : m;
    float2 wloc = wthr; // (opaque: see gr_fwd_pass_b)
    asm volatile ("" : "+v"(wloc.x), "+v"(wloc.y));
    float2 v[16];
    if (t != 15u)
    {
#pragma unroll
        for (int s = 0; s < 16; ++s)
        {
            float a = re[s], c = im[s], pa = re[15 - s], pc = im[15 - s];
            asm volatile ("" : "+v"(a), "+v"(c), "+v"(pa), "+v"(pc)); // the operands are consumed HERE (group_twopass_kernels.hpp)
            const float2 xp = make_float2 (dpp_row_ror8 (pa), dpp_row_ror8 (pc));
            const float2 w = cmul (wloc, make_float2 (kW32re[s], kW32im[s]));
            const float2 z = gr_merge (make_float2 (a, c), xp, w);
            v[s] = make_float2 (z.y, z.x);
            if (CFFT_GR_FENCE > 0 && (s % (CFFT_GR_FENCE > 0 ? CFFT_GR_FENCE : 1)) == CFFT_GR_FENCE - 1)
                CFFT_PHASE_FENCE();
        }
    }
    else
    {
        int mrl = mr; // (opaque: see gr_fwd_pass_b)
        CFFT_LAUNDER_INDEX (mrl);
        const bool self = (l == 7 || l == 15);
        if (self)
        {
#pragma unroll
            for (int s = 0; s < 16; ++s)
                spec[l == 15 ? 1 : 0][mrl + 16 * s] = make_float2 (re[s], im[s]);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s)
        {
            float2 xp = make_float2 (dpp_row_ror8 (re[15 - s]), dpp_row_ror8 (im[15 - s]));
            const int row = mrl + 16 * s;
            if (self)
                xp = (l == 7) ? spec[0][255 - row] : spec[1][(256 - row) & 255];
            const float2 w = cmul (wloc, make_float2 (kW32re[s], kW32im[s]));
            float2 z = gr_merge (make_float2 (re[s], im[s]), xp, w);
            if (s == 0 && l == 15 && m == 0)
                z = make_float2 (re[0] + im[0], re[0] - im[0]); // packed slot 0: X[0] and X[N/2]
            v[s] = make_float2 (z.y, z.x);
        }
    }
    // complex pass A of the 16 columns; this work-item is member mr of column l's team
    fft_reg<16> (v);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b)
        smem[l * kFS256 + 17 * mr + b] = v[b];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
        v[a] = smem[m * kFS256 + l + 17 * a]; // stage-1 role: butterfly l of local column m
#pragma unroll
    for (int a = 1; a < 16; ++a)
        v[a] = cmul (v[a], tw1[(a - 1) * 16 + l]);
    fft_reg<16> (v);
    gr_slot_store (v, bw, slot, base, wt);
}
// slot position of the pass-A results of stage-1 role (column gr_index (t, m), butterfly l): base + 256 b
__device__ __forceinline__ size_t gr_inv_base_a (unsigned t, int tid)
{
    const int l = tid & 15, m = tid >> 4;
    const unsigned col = gr_index (t, m);
    return (size_t) ((col >> 4) * 16u) * 256u + (col & 15u) * 16u + (unsigned) l;
}
// BACKWARD pass B (complex, rows [16 t, 16 t + 16) of the slot, already in v) with the re <-> im swap on the way out
template <class Hook>
__device__ __forceinline__ void gr_inv_pass_b (float2 (&v)[16], unsigned t, float2* __restrict__ dstf, unsigned* bflag, unsigned bval, Hook hook,
                                               float2* __restrict__ smem, const float2* __restrict__ tw1, int tid, bool wt = false)
{
    const int l = tid & 15, m = tid >> 4;
    fft_reg<16> (v);
    hook (1);
    __syncthreads();
    if (tid == 0 && bflag != nullptr)
        flag_store_as (bflag, bval, wt);
#pragma unroll
    for (int b = 0; b < 16; ++b)
        smem[l * kFS256 + 17 * m + b] = v[b];
    hook (2);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
        v[a] = smem[l * kFS256 + m + 17 * a];
    hook (3);
#pragma unroll
    for (int a = 1; a < 16; ++a)
        v[a] = cmul (v[a], tw1[(a - 1) * 16 + m]);
    fft_reg<16> (v);
    float2* __restrict__ dst = dstf + t * 16u + (unsigned) l;
#pragma unroll
    for (int b = 0; b < 16; ++b)
    {
        // (opaque: the optimiser otherwise reads (v[b].y, v[b + 1].x) as one pair out of the array while it still lives in
        // memory, and those straddling accesses keep the whole array there for good: 128 bytes of scratch per work-item)
        float sx = v[b].y, sy = v[b].x;
        asm volatile ("" : "+v"(sx), "+v"(sy));
        gstore (&dst[(unsigned) (m + 16 * b) * 256u], make_float2 (sx, sy));
    }
}
struct GrNoHook
{
    __device__ __forceinline__ void operator() (int) const {}
};

template <bool INV, bool MULH = false>
__global__ __launch_bounds__ (256, 2) void group256_real_kernel (const float2* __restrict__ in,  // forward: the real frame as 65536 float2; backward: the packed spectrum
                                                             float2* __restrict__ work, // max (kGroupSlotsTotal, gridDim.x) slots of 65536 float2
                                                             float2* __restrict__ out,
                                                             const float2* __restrict__ stage_tw, // Plan256
                                                             const float2* __restrict__ big_tw,   // W_65536^(a kb), 16 x 16 block order
                                                             const float2* __restrict__ rtw,      // W_131072^k, k <= 32768
                                                             GroupState* __restrict__ st,
                                                             unsigned* __restrict__ host_err,
                                                             unsigned batch,
                                                             size_t in_stride,
                                                             size_t out_stride,
                                                             unsigned seq,
                                                             unsigned reg_polls,
                                                             unsigned force_solo,
                                                             unsigned form, // bit 3: the next frame's operands a quarter at a time between the steps of pass B (else: one burst behind the pass-B loads)
                                                             const float2* __restrict__ H = nullptr, // MULH (forward only): packed filter spectrum
                                                             float scaling = 1.f)
{
    static_assert (! (INV && MULH), "the product rides on the forward transform's stores");
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    __shared__ unsigned info[8];
    __shared__ float2 tw1[15 * 16];  // stage-1 twiddles of Plan256: [a - 1][k]
    __shared__ float2 spec[2][256];  // member 15: the two self-paired rows / columns (128, 0) by their natural index
    constexpr unsigned N = 65536;
    const int tid = threadIdx.x, tid_k = tid;
    const int l = tid & 15, m = tid >> 4; // lanes run along the 16 strided indices of a tile; m = team member / butterfly
    const bool spread = (form & 8u) != 0;
    const unsigned spin_limit = group_spin_limit (force_solo);
    const bool wt = (force_solo & 2u) != 0; // option "group_handoff" = 1: the listed hand-off form (fused_twopass_kernels.hpp flag_store_as)
    force_solo &= 1u;

    if (tid < 240)
        tw1[tid] = stage_tw[Plan256::tw_offset (1) + tid];
    if (tid == 0)
        group_register (st, info, reg_polls, force_solo);
    __syncthreads();
    const unsigned mode = info[0];

    if (mode == 1)
    {
        if (info[5] == 0)
            return;
        const unsigned xcc = info[1], r = info[2], gg = info[3], total = info[4];
        const unsigned g = r / kGroupMembers, tile = r % kGroupMembers;
        float2* __restrict__ slot = work + (size_t) ((xcc * kGroupMaxPerXcc + g) * 2u) * N;
        unsigned* __restrict__ af = &st->aflag[xcc][g][0];
        unsigned* __restrict__ bf = &st->bflag[xcc][g][0];
        // stage-1 role of pass A: forward column 16 tile + m, backward column gr_index (tile, m); butterfly l
        const size_t base_a = INV ? gr_inv_base_a (tile, tid) : (size_t) (tile * 16) * 256 + (unsigned) m * 16u + (unsigned) l;
        float2 bw[16];
#pragma unroll
        for (int b = 0; b < 16; ++b)
            bw[b] = big_tw[base_a + b * 256];
        const float2 wthr = gr_pair_twiddle (rtw, tile, l, m);

        float nre[16], nim[16];
        unsigned f = gg;
        if (f < batch)
        {
            if constexpr (INV)
                gr_inv_load_a<4> (nre, nim, in + (size_t) f * in_stride, tile, tid);
            else
                g256_load_a<false> (nre, nim, in + (size_t) f * in_stride + tile * 16, m, l);
        }
        for (unsigned k = 0; f < batch; f += total, ++k)
        {
            // ---- pass A
            float2 va[16];
            if constexpr (! INV)
            {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    va[s] = make_float2 (nre[s], nim[s]);
                fft256_two_stage_lds (va, tw1, l, smem, l, m, m, l);
            }
            if (k != 0)
            {
                // every member has read frame k - 1 out of the slot (each wave looks for itself: no barrier needed; the backward
                // transform waits in front of its whole pass A, whose last step stores -- the flags are normally up long before)
                unsigned n = 0;
                while (! flags_reached (bf, k, tid & 63))
                {
                    __builtin_amdgcn_s_sleep (1);
                    if (++n > spin_limit)
                    {
                        if ((tid & 63) == 0)
                            group_fail (st, host_err, seq, 2u);
                        break;
                    }
                }
            }
            if constexpr (INV)
                gr_inv_pass_a (nre, nim, tile, wthr, bw, slot, base_a, smem, tw1, spec, tid, wt);
            else
                gr_slot_store (va, bw, slot, base_a, wt);
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores are in L2
            __syncthreads();
            if (tid == 0)
                flag_store_as (&af[tile], k + 1u, wt);
            if (tid < 64)
            {
                unsigned n = 0;
                while (! flags_reached (af, k + 1u, tid))
                {
                    __builtin_amdgcn_s_sleep (2);
                    if (++n > spin_limit)
                    {
                        if (tid == 0)
                            group_fail (st, host_err, seq, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            // ---- pass B
            float2 v[16];
            if constexpr (INV)
            {
                const float2* __restrict__ srcb = slot + tile * 256 + tid;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    v[s] = gload_l2 (&srcb[s * 16 * 256]);
            }
            else
                gr_fwd_load_b (v, slot, tile, tid);
            // the next frame's operands: behind the pass-B loads (vector memory operations retire in issue order), in one burst or
            // (`spread`) a quarter here and the others between the steps of pass B
            const unsigned fn = f + total;
            const bool pre = fn < batch;
            const float2* __restrict__ nsrc = in + (size_t) fn * in_stride;
#define CFFT_GR_PART(G)                                                                 \
    {                                                                                   \
        if constexpr (INV)                                                              \
            gr_inv_load_a<G> (nre, nim, nsrc, tile, tid);                               \
        else                                                                            \
            g256_load_a_part<false, G> (nre, nim, nsrc + tile * 16, m, l);              \
    }
            if (pre && ! spread)
            {
                CFFT_GR_PART (0)
                CFFT_GR_PART (1)
                CFFT_GR_PART (2)
                CFFT_GR_PART (3)
            }
            if (pre && spread)
                CFFT_GR_PART (0)
            CFFT_PHASE_FENCE();
            auto hook = [&] (int step) __attribute__ ((always_inline)) {
                CFFT_PHASE_FENCE();
                if (pre && spread)
                {
                    if (step == 1)
                        CFFT_GR_PART (1)
                    else if (step == 2)
                        CFFT_GR_PART (2)
                    else
                        CFFT_GR_PART (3)
                }
                CFFT_PHASE_FENCE();
            };
            if constexpr (INV)
                gr_inv_pass_b (v, tile, out + (size_t) f * out_stride, &bf[tile], k + 1u, hook, smem, tw1, tid, wt);
            else
                gr_fwd_pass_b<MULH> (v, tile, wthr, out + (size_t) f * out_stride, &bf[tile], k + 1u, hook, smem, tw1, spec, tid, H, scaling, wt);
#undef CFFT_GR_PART
        }
        return;
    }

#ifndef CFFT_GR_NO_SOLO
    // ---- SOLO mode: whole frames by one workgroup, its own slot, no dependency on anybody
    float2* __restrict__ slot = work + (size_t) blockIdx.x * N;
    for (;;)
    {
        __syncthreads(); // everybody has read info[6] of the previous round (and the slot of the previous frame)
        if (tid == 0)
            info[6] = __hip_atomic_fetch_add (&st->next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned f = info[6];
        if (f >= batch)
            break;
#pragma unroll 1
        for (unsigned t = 0; t < 16; ++t)
        {
            float re[16], im[16];
            const size_t base_a = INV ? gr_inv_base_a (t, tid) : (size_t) (t * 16) * 256 + (unsigned) m * 16u + (unsigned) l;
            float2 bw[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                bw[b] = big_tw[base_a + b * 256];
            if constexpr (INV)
            {
                gr_inv_load_a<4> (re, im, in + (size_t) f * in_stride, t, tid);
                gr_inv_pass_a (re, im, t, gr_pair_twiddle (rtw, t, l, m), bw, slot, base_a, smem, tw1, spec, tid);
            }
            else
            {
                g256_load_a<false> (re, im, in + (size_t) f * in_stride + t * 16, m, l);
                float2 v[16];
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    v[s] = make_float2 (re[s], im[s]);
                fft256_two_stage_lds (v, tw1, l, smem, l, m, m, l);
                gr_slot_store (v, bw, slot, base_a);
            }
        }
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); // every wave's stores are in L2 (or beyond) ...
        __syncthreads();                                   // ... before any wave reads the slot back, past its L1
#pragma unroll 1
        for (unsigned t = 0; t < 16; ++t)
        {
            // (per tile: nothing derived from the work-item index is lifted out of this loop and held across it -- the fall-back mode has no
            // registers to spare next to the product's filter bins)
            int tid = tid_k;
            CFFT_LAUNDER_INDEX (tid);
            const int l = tid & 15, m = tid >> 4;
            float2 v[16];
            if constexpr (INV)
            {
                const float2* __restrict__ srcb = slot + t * 256 + tid;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    v[s] = gload_l2 (&srcb[s * 16 * 256]);
                gr_inv_pass_b (v, t, out + (size_t) f * out_stride, nullptr, 0u, GrNoHook(), smem, tw1, tid);
            }
            else
            {
                gr_fwd_load_b (v, slot, t, tid);
                gr_fwd_pass_b<MULH> (v, t, gr_pair_twiddle (rtw, t, l, m), out + (size_t) f * out_stride, nullptr, 0u, GrNoHook(), smem, tw1, spec, tid, H, scaling);
            }
        }
    }
#endif
}
} // namespace cfft
