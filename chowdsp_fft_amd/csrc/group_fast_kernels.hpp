// The one-launch, XCD-local two-pass form (group_twopass_kernels.hpp: static groups of workgroups, the inter-pass hand-off
// inside an XCD's L2) for the register-direct tiles of tile_fast_kernels.hpp: N = n1 x n2 = 512 x 256 (2^17) and 512 x 512 (2^18).
//
// What changes against N = 65536: a frame is 1 / 2 MiB, and a group's slot has to stay resident in a 4 MiB L2 next to the
// streamed frames (about 2.5 MiB of room, profiles/r03_group_two_pass.txt) -- so an XCD holds TWO groups of 16 members (2^17)
// or ONE group of 32 (2^18): one 256-thread workgroup per CU, which therefore gets the whole register file (512 per
// work-item): 32 points, the 32 prefetched operands of the next frame, the member's 32 inter-pass twiddles and both cores'
// stage twiddles all stay in registers.  A member owns SA adjacent column tiles of pass A (two 128-thread halves for
// n2 = 256) and SB row tiles of pass B; the tile transforms, the exchange through the LDS plane and the plain [a][kb] layout
// of the intermediate are those of pass_a_fast_kernel / pass_b_fast_kernel.  Registration, mode decision (GROUP / SOLO),
// flags and the fall-back mode: as in group256_kernel.
// 2^19 (4 MiB per frame) and 2^20 do not fit an XCD's L2 at all: they keep the two-launch form.
#pragma once
#include "group_twopass_kernels.hpp"
#include "tile_fast_kernels.hpp"

namespace cfft
{
template <int N1, int N2>
struct GroupFast
{
    using FA = FastTile<N2>; // pass A: N2-point column transforms, tiles of 16 columns
    using FB = FastTile<N1>; // pass B: N1-point row transforms, tiles of 16 rows
    static constexpr int T = 256;
    static constexpr int SA = T / FA::threads, SB = T / FB::threads; // tiles a member transforms at once in each pass
    static_assert (SA >= 1 && SB >= 1 && SA * FA::threads == T && SB * FB::threads == T, "tile teams must fill the workgroup");
    static constexpr int M = (N1 / 16) / SA; // members per group
    static_assert (M == (N2 / 16) / SB && (M == 16 || M == 32), "both passes must split over the same number of members");
    static constexpr size_t plane_a = FA::lds_bytes / sizeof (float), plane_b = FB::lds_bytes / sizeof (float);
    static constexpr size_t lds_bytes = (SA * FA::lds_bytes > SB * FB::lds_bytes) ? SA * FA::lds_bytes : SB * FB::lds_bytes;
    static constexpr size_t N = (size_t) N1 * N2;
};

// wave-uniform: have the group's M flags reached `need`?
template <int M>
__device__ __forceinline__ bool flags_reached_m (const unsigned* flags, unsigned need, int lane)
{
    const unsigned f = ld_relaxed (flags + (lane & (M - 1)));
    return __all (f >= need) != 0;
}
template <int M, int SLEEP>
__device__ __forceinline__ void group_wait_m (const unsigned* flags, unsigned need, int lane, GroupState* st, unsigned* host_err, unsigned seq, unsigned code, unsigned spin_limit)
{
    unsigned n = 0;
    while (! flags_reached_m<M> (flags, need, lane))
    {
        __builtin_amdgcn_s_sleep (SLEEP);
        if (++n > spin_limit)
        {
            if (lane == 0)
                group_fail (st, host_err, seq, code);
            break;
        }
    }
}

template <int N1, int N2, bool INV>
__global__ __launch_bounds__ (256, 1) void groupfast_kernel (const float2* __restrict__ in,
                                                            float2* __restrict__ work, // max (kGroupMaxXcc * kGroupMaxPerXcc, gridDim.x) slots of N float2
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ tw_a, // factorised stage-1 rows of the n2 core
                                                            const float2* __restrict__ tw_b, // ... of the n1 core
                                                            const float2* __restrict__ big_tw, // [a][kb] = W_N^(a kb)
                                                            GroupState* __restrict__ st,
                                                            unsigned* __restrict__ host_err,
                                                            unsigned batch,
                                                            size_t in_stride,
                                                            size_t out_stride,
                                                            unsigned seq,
                                                            unsigned reg_polls,
                                                            unsigned force_solo,
                                                            unsigned form) // where the next frame's operands are requested: 0 behind the pass-B loads; 1 in front of the hand-off (waves 1-3; wave 0, which polls, behind the pass-B loads); 2 in front of it, every wave
{
    using GF = GroupFast<N1, N2>;
    using FA = typename GF::FA;
    using FB = typename GF::FB;
    extern __shared__ __attribute__ ((aligned (16))) float gf_plane[];
    __shared__ unsigned info[8];
    constexpr size_t N = GF::N;
    const int tid0 = threadIdx.x;
    // pass A: `sub_a` of the member's SA tiles, work-item ta of that tile's team; pass B the same with sub_b / tb
    const int sub_a = tid0 / FA::threads, ta0 = tid0 % FA::threads;
    const int sub_b = tid0 / FB::threads, tb0 = tid0 % FB::threads;
    float* __restrict__ plane_a = gf_plane + (size_t) sub_a * GF::plane_a;
    float* __restrict__ plane_b = gf_plane + (size_t) sub_b * GF::plane_b;
    const unsigned spin_limit = group_spin_limit (force_solo);
    force_solo &= 1u; // (bit 1, the listed hand-off form, is not offered by this kernel: capi.cpp runs two launches instead)
    if (tid0 == 0)
        group_register (st, info, reg_polls, force_solo, (unsigned) GF::M);
    __syncthreads();
    const unsigned mode = info[0];

    float2 wa[10];
    ft_load_twiddles<FA> (wa, tw_a, ta0 & 31); // stage-1 role of pass A: butterfly index i = ta & 31
    float2 wb[FB::Q][10];
#pragma unroll
    for (int q = 0; q < FB::Q; ++q)
        ft_load_twiddles<FB> (wb[q], tw_b, (tb0 >> 4) + FB::Tn * q); // pass B: butterflies i_q = member + Tn q

// pass A of one tile (columns [A0, A0 + 16) of a frame whose operands are in V): transform, times BW, into SLOT[a][kb]
#define CFFT_GF_PASS_A_STORE(V, A0, SLOT, BW, TA)                                                   \
    {                                                                                               \
        const unsigned i_ = (unsigned) (TA) & 31u;                                                  \
        _Pragma ("unroll") for (int q = 0; q < FA::Q; ++q)                                          \
        {                                                                                           \
            const unsigned a_ = (A0) + ((unsigned) (TA) >> 5) + (unsigned) (FA::Tn / 2) * q;        \
            float2* __restrict__ dst_ = (SLOT) + (size_t) a_ * N2 + i_;                             \
            _Pragma ("unroll") for (int b = 0; b < FA::R1; ++b)                                     \
                dst_[32 * b] = cmul (V[q * FA::R1 + b], BW[q * FA::R1 + b]);                        \
        }                                                                                           \
    }
#define CFFT_GF_LOAD_BW(BW, A0, TA)                                                                 \
    {                                                                                               \
        const unsigned i_ = (unsigned) (TA) & 31u;                                                  \
        _Pragma ("unroll") for (int q = 0; q < FA::Q; ++q)                                          \
        {                                                                                           \
            const unsigned a_ = (A0) + ((unsigned) (TA) >> 5) + (unsigned) (FA::Tn / 2) * q;        \
            _Pragma ("unroll") for (int b = 0; b < FA::R1; ++b)                                     \
                BW[q * FA::R1 + b] = big_tw[(size_t) a_ * N2 + i_ + 32 * b];                        \
        }                                                                                           \
    }
#define CFFT_GF_LOAD_A(V, SRC, TA, SWAP)                                                                  \
    {                                                                                               \
        const float2* __restrict__ s_ = (SRC) + ((unsigned) (TA) & 15u);                            \
        const unsigned m_ = (unsigned) (TA) >> 4;                                                   \
        _Pragma ("unroll") for (int a = 0; a < 32; ++a)                                             \
        {                                                                                           \
            const float2 x_ = gload (&s_[(size_t) (m_ + FA::Tn * a) * N1]);                         \
            V[a] = (INV && (SWAP)) ? make_float2 (x_.y, x_.x) : x_;                                 \
        }                                                                                           \
    }
#define CFFT_GF_STORE_OUT(V, DST, TB)                                                               \
    {                                                                                               \
        float2* __restrict__ d_ = (DST) + ((unsigned) (TB) & 15u);                                  \
        const unsigned ms_ = (unsigned) (TB) >> 4;                                                  \
        _Pragma ("unroll") for (int q = 0; q < FB::Q; ++q)                                          \
            _Pragma ("unroll") for (int b = 0; b < FB::R1; ++b)                                     \
            {                                                                                       \
                const float2 y_ = V[q * FB::R1 + b];                                                \
                gstore (&d_[(size_t) (ms_ + FB::Tn * q + 32 * b) * N2], INV ? make_float2 (y_.y, y_.x) : y_); \
            }                                                                                       \
    }

    if (mode == 1)
    {
        if (info[5] == 0)
            return;
        const unsigned xcc = info[1], r = info[2], gg = info[3], total = info[4];
        const unsigned g = r / (unsigned) GF::M, member = r % (unsigned) GF::M;
        float2* __restrict__ slot = work + (size_t) (xcc * kGroupMaxPerXcc + g) * N;
        unsigned* __restrict__ af = &st->aflag[xcc][g][0];
        unsigned* __restrict__ bf = &st->bflag[xcc][g][0];
        const unsigned a0 = (member * (unsigned) GF::SA + (unsigned) sub_a) * 16u;  // this team's columns of pass A
        const unsigned kb0 = (member * (unsigned) GF::SB + (unsigned) sub_b) * 16u; // ... rows of pass B
        float2 bw[32];
        CFFT_GF_LOAD_BW (bw, a0, ta0)
        float2 nv[32];
        unsigned f = gg;
        if (f < batch)
            CFFT_GF_LOAD_A (nv, in + (size_t) f * in_stride + a0, ta0, false)
        for (unsigned k = 0; f < batch; f += total, ++k)
        {
            int ta = ta0, tb = tb0;
            CFFT_LAUNDER_INDEX (ta);
            CFFT_LAUNDER_INDEX (tb);
            float2 v[32];
#pragma unroll
            for (int a = 0; a < 32; ++a)
            {
                if constexpr (INV)
                {
                    // the backward transform's re <-> im swap where the operands are CONSUMED, behind an opaque statement (at the
                    // request it is an instruction that waits for the prefetch where it was issued: group_twopass_kernels.hpp)
                    float2 x = nv[a];
                    asm volatile ("" : "+v"(x.x), "+v"(x.y));
                    v[a] = make_float2 (x.y, x.x);
                }
                else
                    v[a] = nv[a];
            }
            // ---- pass A
            fft_reg32<false> (v, wa);
            CFFT_PHASE_FENCE();
            ft_exchange<FA, true> (v, plane_a, ta);
            CFFT_PHASE_FENCE();
            ft_stage1_all<FA> (v, wa, wa, wa, wa);
            CFFT_PHASE_FENCE();
            if (k != 0)
                group_wait_m<GF::M, 1> (bf, k, tid0 & 63, st, host_err, seq, 2u, spin_limit); // every member has read frame k - 1 out of the slot
            CFFT_GF_PASS_A_STORE (v, a0, slot, bw, ta)
            // With one workgroup per CU the wait for the group is idle time for the whole CU: the next frame's operands can be in
            // flight during it.  Vector memory operations retire in issue order, so the polling wave's flag loads would come back
            // behind its own prefetch: form 1 leaves wave 0's share behind the pass-B loads.
            const unsigned fn = f + total;
            const bool pre_here = fn < batch && (form == 2u || (form == 1u && tid0 >= 64));
            if (pre_here)
            {
                CFFT_GF_LOAD_A (nv, in + (size_t) fn * in_stride + a0, ta, false)
                asm volatile ("s_waitcnt vmcnt(32)" ::: "memory"); // all but the 32 prefetch loads: this wave's stores are in L2
            }
            else
                asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
#if CFFT_GROUP_WAVE_FLAGS
            // the hand-off by wave (group_twopass_kernels.hpp): word 4 member + wave, every wave polls all 4 M words itself,
            // no workgroup barrier on either side
            if ((tid0 & 63) == 0)
                flag_store (&af[4u * member + ((unsigned) tid0 >> 6)], k + 1u);
            {
                unsigned n = 0;
                for (;;)
                {
                    bool up = ld_relaxed (af + (tid0 & 63)) >= k + 1u;
                    if (GF::M == 32)
                        up = up && ld_relaxed (af + 64 + (tid0 & 63)) >= k + 1u;
                    if (__all (up) != 0)
                        break;
                    __builtin_amdgcn_s_sleep (2);
                    if (++n > spin_limit)
                    {
                        if ((tid0 & 63) == 0)
                            group_fail (st, host_err, seq, 1u);
                        break;
                    }
                }
            }
#else
            __syncthreads();
            if (tid0 == 0)
                flag_store (&af[member], k + 1u);
            if (tid0 < 64)
                group_wait_m<GF::M, 2> (af, k + 1u, tid0, st, host_err, seq, 1u, spin_limit);
            __syncthreads();
#endif
            // ---- pass B: rows [kb0, kb0 + 16) of the slot
            {
                const float2* __restrict__ src = slot + kb0 + ((unsigned) tb & 15u);
                const unsigned mb = (unsigned) tb >> 4;
#pragma unroll
                for (int a = 0; a < 32; ++a)
                    v[a] = gload_l2 (&src[(size_t) (mb + FB::Tn * a) * N2]);
            }
            if (fn < batch && ! pre_here)
                CFFT_GF_LOAD_A (nv, in + (size_t) fn * in_stride + a0, ta, false)
            CFFT_PHASE_FENCE();
            fft_reg32<false> (v, wb[0]);
            CFFT_PHASE_FENCE();
            ft_exchange<FB, false> (v, plane_b, tb);
            if (tid0 == 0)
                flag_store (&bf[member], k + 1u); // (every wave has been through a barrier since its loads returned)
            CFFT_PHASE_FENCE();
            ft_stage1_all<FB> (v, wb[0], wb[FB::Q > 1 ? 1 : 0], wb[FB::Q > 2 ? 2 : 0], wb[FB::Q > 2 ? 3 : 0]);
            CFFT_PHASE_FENCE();
            CFFT_GF_STORE_OUT (v, out + (size_t) f * out_stride + kb0, tb)
        }
        return;
    }

    // ---- SOLO mode: whole frames by one workgroup through a slot of its own
    float2* __restrict__ slot = work + (size_t) blockIdx.x * N;
    for (;;)
    {
        __syncthreads();
        if (tid0 == 0)
            info[6] = __hip_atomic_fetch_add (&st->next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned f = info[6];
        if (f >= batch)
            break;
#pragma unroll 1
        for (unsigned round = 0; round < (unsigned) (N1 / 16 / GF::SA); ++round)
        {
            const unsigned a0 = (round * (unsigned) GF::SA + (unsigned) sub_a) * 16u;
            float2 v[32], bw[32];
            CFFT_GF_LOAD_A (v, in + (size_t) f * in_stride + a0, ta0, true)
            CFFT_GF_LOAD_BW (bw, a0, ta0)
            fft_reg32<false> (v, wa);
            ft_exchange<FA, true> (v, plane_a, ta0);
            ft_stage1_all<FA> (v, wa, wa, wa, wa);
            CFFT_GF_PASS_A_STORE (v, a0, slot, bw, ta0)
        }
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (unsigned round = 0; round < (unsigned) (N2 / 16 / GF::SB); ++round)
        {
            const unsigned kb0 = (round * (unsigned) GF::SB + (unsigned) sub_b) * 16u;
            float2 v[32];
            const float2* __restrict__ src = slot + kb0 + ((unsigned) tb0 & 15u);
            const unsigned mb = (unsigned) tb0 >> 4;
#pragma unroll
            for (int a = 0; a < 32; ++a)
                v[a] = gload_l2 (&src[(size_t) (mb + FB::Tn * a) * N2]);
            fft_reg32<false> (v, wb[0]);
            ft_exchange<FB, false> (v, plane_b, tb0);
            ft_stage1_all<FB> (v, wb[0], wb[FB::Q > 1 ? 1 : 0], wb[FB::Q > 2 ? 2 : 0], wb[FB::Q > 2 ? 3 : 0]);
            CFFT_GF_STORE_OUT (v, out + (size_t) f * out_stride + kb0, tb0)
        }
    }
#undef CFFT_GF_PASS_A_STORE
#undef CFFT_GF_LOAD_BW
#undef CFFT_GF_LOAD_A
#undef CFFT_GF_STORE_OUT
}
} // namespace cfft
