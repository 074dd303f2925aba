// Two-pass (four-step) transform of N = 65536 in ONE launch, the inter-pass hand-off inside an XCD's L2, with STATIC
// groups of workgroups instead of the ticket queues of fused_twopass_kernels.hpp.
//
// What the ticket form measured: its fabric traffic is 1.5 x the algorithmic bytes (the two-launch form:
// 2 x), but it takes longer than two launches, because an item's way from its ticket to its completion flag (5-6 us) is
// longer than the step the queue advances by (2.8 us): items wait for each other.  Here nothing is queued:
//
//   * At its start every workgroup reads the XCD it runs on (HW_REG_XCC_ID) and draws a rank on that XCD.  16 consecutive
//     ranks form a GROUP; member m of a group owns tile m of both passes of every frame the group transforms -- frames
//     g, g + G, g + 2 G ... for global group number g of G.  A member therefore needs the same 16 inter-pass twiddles per
//     work-item for every frame (they live in VGPRs: no twiddle traffic at all), and the group's 16 tiles of one pass
//     cover whole 2 KiB rows of the frame at about the same time.
//   * Per frame one hand-off: a member stores its pass-A tile into the group's slot (plain stores: the lines stay in this
//     XCD's L2 and are written through), waits for the stores' acknowledgements, sets its flag; when all 16 flags of the
//     frame are up it reads its pass-B tile back from L2 (L1-bypassing loads).  ONE slot per group: the footprint that
//     has to survive in an XCD's 4 MiB L2 next to the streamed frames is then 512 KiB per group (2 MiB at two workgroups
//     per CU).  With two slots used alternately no second flag would be needed (a member sees the 16 flags of frame k only
//     after every member has finished reading frame k - 1), but the cyclic 4 MiB footprint thrashes L2: measured, three
//     quarters of the pass-B loads then come back over the fabric (FETCH_SIZE 7.0 instead of 4 GiB per 8192 frames,
//     profiles/r03_group_two_pass.txt; `two_slots` keeps that form for A/B runs).  So a second flag per member says "my
//     pass-B loads of frame k have returned", and every wave looks at the group's 16 of them before it overwrites the
//     slot with frame k + 1 -- two steps of arithmetic later, when they are normally all up.
//   * The next frame's pass-A operands are requested right behind the pass-B loads, i.e. before this frame's pass-B
//     arithmetic and stores (vector memory operations retire in issue order: behind the pass-B loads they delay nothing).
//
// Placement is still NOT assumed.  Groups exist only if every workgroup of the grid is resident at the same time; that is
// established, not presumed: every workgroup registers, and the mode of the launch is decided ONCE by a compare-and-swap
// on one word -- GROUP by the workgroup whose registration completes the grid, SOLO by any workgroup that has waited
// `reg_polls` polls for that to happen (a grid that is not fully resident, e.g. beside another stream's kernel).  In SOLO
// mode every workgroup transforms whole frames by itself (16 pass-A tiles into its own slot, then 16 pass-B tiles), with
// no dependency on any other workgroup: slower (the slots no longer fit L2), always correct, for any placement and any
// number of resident workgroups.  Workgroups that start after the decision find it made.
//
// Reference: the same passes for any N, chowdsp_fft.h:131-134; cfftf1_ps simd/chowdsp_fft_impl_sse.cpp:372-432.
#pragma once
#include "fused_twopass_kernels.hpp"

// CFFT_GROUP_EXPERIMENT builds (tools/build_variants.sh; never the shipped library) read timing-only switches from bits 8.. of
// `force_solo` -- results are WRONG with any of them set: 1 no waiting for the group, 2 no slot stores, 4 no slot loads,
// 8 no input loads after the first frame, 16 no output stores
#ifndef CFFT_GROUP_EXPERIMENT
#define CFFT_GROUP_EXPERIMENT 0
#endif
#if CFFT_GROUP_EXPERIMENT
#define CFFT_GROUP_DBG(BIT) ((dbg & (BIT)) != 0)
#define CFFT_GROUP_STAMP(I)                                                                \
    if (stamping && tid == 0 && k >= 100u && k < 108u)                                     \
        st->stamps[k - 100u][I] = __builtin_amdgcn_s_memtime();
#else
#define CFFT_GROUP_DBG(BIT) false
#define CFFT_GROUP_STAMP(I)
#endif

// The backward transform as conj (DFT (conj x)) (1) instead of swap (DFT (swap x)) (0, default).  An A/B switch kept from the
// search for why the backward instantiation runs 5 % behind the forward one (2.43 against 2.27 ms per 8192 frames): with the
// packed arithmetic a swap is two register moves per value where a conjugation rides on operand modifiers -- measured equal
// (2.428 ms both ways), so the moves are not it.
#ifndef CFFT_GROUP_INV_CONJ
#define CFFT_GROUP_INV_CONJ 0
#endif
// Hand-off flags per WAVE and no workgroup barrier around the hand-off (1) or per member behind two barriers (0, default).
// Round-4 A/B (profiles/r04_group_two_pass.txt): the per-wave form is no faster at 8192 frames (0.483-0.485 against 0.488 of
// 8 TB/s) and 5 % slower at 2048 (0.461 against 0.484); 2^17 / 2^18 0.393 / 0.377 against 0.395 / 0.385 -- four waves polling
// 64 words each cost more than the two barriers they replace.
#ifndef CFFT_GROUP_WAVE_FLAGS
#define CFFT_GROUP_WAVE_FLAGS 0
#endif

// workgroups per CU the compiler has to leave room for (2: 256 VGPRs; 3: 168)
#ifndef CFFT_GROUP_MIN_WG
#define CFFT_GROUP_MIN_WG 2
#endif
namespace cfft
{
constexpr int kGroupMaxXcc = 16;      // HW_REG_XCC_ID is a 4-bit field
constexpr int kGroupMaxPerXcc = 8;    // groups per XCD that have flags and slots (more workgroups than 128 per XCD stay idle)
constexpr unsigned kGroupMembers = 16;
constexpr unsigned kGroupSlotsTotal = kGroupMaxXcc * kGroupMaxPerXcc * 2; // slots of N float2 the group mode addresses

struct GroupState
{
    unsigned registered; // workgroups that have drawn their rank
    unsigned mode;       // 0 undecided, 1 GROUP, 2 SOLO
    unsigned next_frame; // SOLO mode: frames are claimed one by one
    unsigned error;      // a bounded wait ran out (also stored to the host word, tagged with the launch number)
    unsigned pad0[28];
    struct
    {
        unsigned count; // ranks drawn on this XCD
        unsigned pad[31];
    } xcc[kGroupMaxXcc];
    // [xcc][group][member]: number of frames whose pass-A tile the member has stored (one 64-byte poll per group).
    // group256_kernel (round 4): [4 member + wave] -- every WAVE publishes its own stores and polls all 64 words with one load
    unsigned aflag[kGroupMaxXcc][kGroupMaxPerXcc][128]; // (groupfast_kernel with 32 members: 128 waves)
    // ... and whose pass-B tile it has loaded (lines of their own: they are written at another time than the aflags)
    unsigned bflag[kGroupMaxXcc][kGroupMaxPerXcc][32];
#if CFFT_GROUP_EXPERIMENT
    unsigned long long stamps[8][16]; // s_memtime at the phase boundaries of eight iterations of group 0's member 0 (dbg & 32)
#endif
};

__device__ __forceinline__ void group_fail (GroupState* st, unsigned* host_err, unsigned seq, unsigned code)
{
    __hip_atomic_store (&st->error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (host_err != nullptr)
        __hip_atomic_store (host_err, (seq << 8) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ... rows [4 G, 4 G + 4) only (the operands requested a quarter at a time, between the steps of the transform before)
template <bool INV, int G>
__device__ __forceinline__ void g256_load_a_part (float (&re)[16], float (&im)[16], const float2* __restrict__ src, int t0, int c0)
{
#pragma unroll
    for (int s = 4 * G; s < 4 * G + 4; ++s)
    {
        const float2 x = gload (&src[(unsigned) (t0 + 16 * s) * 256u + c0]);
        re[s] = (INV && ! CFFT_GROUP_INV_CONJ) ? x.y : x.x;
        im[s] = INV ? (CFFT_GROUP_INV_CONJ ? -x.y : x.x) : x.y;
    }
}

// pass A operands of one tile: 16 rows per work-item, lanes along the tile's 16 columns (128 contiguous bytes per row)
template <bool INV>
__device__ __forceinline__ void g256_load_a (float (&re)[16], float (&im)[16], const float2* __restrict__ src, int t0, int c0)
{
#pragma unroll
    for (int s = 0; s < 16; ++s)
    {
        const float2 x = gload (&src[(unsigned) (t0 + 16 * s) * 256u + c0]);
        re[s] = (INV && ! CFFT_GROUP_INV_CONJ) ? x.y : x.x;
        im[s] = INV ? (CFFT_GROUP_INV_CONJ ? -x.y : x.x) : x.y;
    }
}

// Lane 0 of every workgroup, first thing: draw a rank on this XCD, register, learn (or make) the launch's mode.
// info[0] = mode (1 GROUP, 2 SOLO), [1] = XCD, [2] = rank on it, [3] = global group number, [4] = groups in all, [5] = 0 for a
// rank beyond its XCD's last complete group (nothing to do in GROUP mode).
__device__ __forceinline__ void group_register (GroupState* __restrict__ st, unsigned* info, unsigned reg_polls, unsigned force_solo, unsigned members = kGroupMembers)
{
    const unsigned xcc = hw_xcc_id();
    const unsigned r = __hip_atomic_fetch_add (&st->xcc[xcc].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); // the rank is drawn before the registration below is counted
    const unsigned tot = __hip_atomic_fetch_add (&st->registered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (tot == gridDim.x || force_solo != 0)
    {
        unsigned expect = 0;
        (void) __hip_atomic_compare_exchange_strong (&st->mode, &expect, force_solo != 0 ? 2u : 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned mode, n = 0;
    while ((mode = ld_relaxed (&st->mode)) == 0)
    {
        __builtin_amdgcn_s_sleep (4);
        if (++n > reg_polls)
        {
            unsigned expect = 0; // not every workgroup of the grid is running: nobody may wait for another one
            (void) __hip_atomic_compare_exchange_strong (&st->mode, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    unsigned gg = 0, total = 0, valid = 1;
    if (mode == 1)
    {
        // every rank has been drawn: the XCD populations are final.  Global group number = groups of the XCDs before this one + r / 16
        unsigned mine = 0;
        for (unsigned x = 0; x < (unsigned) kGroupMaxXcc; ++x)
        {
            unsigned g = ld_relaxed (&st->xcc[x].count) / members;
            g = g > (unsigned) kGroupMaxPerXcc ? (unsigned) kGroupMaxPerXcc : g;
            if (x == xcc)
            {
                gg = total + r / members;
                mine = g;
            }
            total += g;
        }
        valid = (r / members < mine) ? 1u : 0u; // a rank beyond the last complete group has nothing to do
        // No XCD has a complete group (a small grid: CU mask, an occupancy drop, a small partition): nobody would transform
        // anything.  The counts are final and the same for every workgroup, so every workgroup takes this decision alike:
        // the launch runs in the fall-back mode.
        if (total == 0)
        {
            mode = 2;
            valid = 1;
        }
    }
    info[0] = mode;
    info[1] = xcc;
    info[2] = r;
    info[3] = gg;
    info[4] = total;
    info[5] = valid;
}

// The `force_solo` kernel argument of the group kernels carries two developer fields besides bit 0 (force the fall-back mode):
// bits 8..15 the timing-only switches of CFFT_GROUP_EXPERIMENT builds, bits 16..31 a spin limit that replaces kFusedSpinLimit
// (option "group_spin_limit"; tests set 1 so that a bounded wait DOES run out and the failure path can be exercised end to end).
__device__ __forceinline__ unsigned group_spin_limit (unsigned force_solo_arg)
{
    const unsigned v = force_solo_arg >> 16;
    return v != 0 ? v : kFusedSpinLimit;
}

// one wave: wait until the group's 16 flags at `flags` have reached `need` (bounded; uniform)
template <int SLEEP>
__device__ __forceinline__ void group_wait (const unsigned* flags, unsigned need, int lane, GroupState* st, unsigned* host_err, unsigned seq, unsigned code, unsigned spin_limit)
{
    unsigned n = 0;
    while (! flags_reached (flags, need, lane))
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

// fft256_two_stage (multipass_kernels.hpp) with the stage-1 twiddles read from LDS, tw[(a - 1) * 16 + k]: pass A and pass B
// index them by different roles, and both sets in VGPRs next to the resident inter-pass twiddles do not fit 256 registers
__device__ __forceinline__ void fft256_two_stage_lds (float2 (&v)[16], const float2* __restrict__ tw, int k, float2* __restrict__ smem,
                                                      int c0, int t0, int c1, int i1)
{
    fft_reg<16> (v);
    __syncthreads(); // the previous tile's stage-1 reads are finished
#pragma unroll
    for (int b = 0; b < 16; ++b)
        smem[c0 * kFS256 + 17 * t0 + b] = v[b];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
        v[a] = smem[c1 * kFS256 + i1 + 17 * a];
#pragma unroll
    for (int a = 1; a < 16; ++a)
        v[a] = cmul (v[a], tw[(a - 1) * 16 + k]);
    fft_reg<16> (v);
}

template <bool INV>
__global__ __launch_bounds__ (256, CFFT_GROUP_MIN_WG) void group256_kernel (const float2* __restrict__ in,
                                                        float2* __restrict__ work, // max (kGroupSlotsTotal, gridDim.x) slots of 65536 float2
                                                        float2* __restrict__ out,
                                                        const float2* __restrict__ stage_tw,
                                                        const float2* __restrict__ big_tw,
                                                        GroupState* __restrict__ st,
                                                        unsigned* __restrict__ host_err,
                                                        unsigned batch,
                                                        size_t in_stride,
                                                        size_t out_stride,
                                                        unsigned seq,
                                                        unsigned reg_polls,
                                                        unsigned force_solo,
                                                        unsigned form) // bit 0: two slots per group; bit 1: early prefetch; bit 2: ... wave 0's share late; bit 3: prefetch spread over pass B
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    __shared__ unsigned info[8];
    __shared__ float2 tw1[15 * 16]; // stage-1 twiddles of Plan256: [a - 1][k]
    constexpr unsigned N = 65536;
    const int tid = threadIdx.x;
    // pass A roles (multipass_kernels.hpp): stage 0 lanes run along the tile's 16 columns, stage 1 along the butterfly index
    const int c0 = tid & 15, t0 = tid >> 4;
    const int i1 = tid & 15, c1 = tid >> 4;
    // pass B roles: lanes run along the tile's 16 rows in both stages
    const int kbl = tid & 15, tb = tid >> 4;
    const unsigned two_slots = form & 1u;
    const bool early = (form & 2u) != 0, w0late = (form & 4u) != 0, spread = (form & 8u) != 0;
#if CFFT_GROUP_WAVE_FLAGS
    const bool incr = false;
#else
    const bool incr = (form & 16u) != 0;
#endif
    const unsigned spin_limit = group_spin_limit (force_solo);
#if CFFT_GROUP_EXPERIMENT
    const unsigned dbg = (force_solo >> 8) & 255u;
#endif
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
        float2* __restrict__ slots = work + (size_t) ((xcc * kGroupMaxPerXcc + g) * 2u) * N;
        unsigned* __restrict__ af = &st->aflag[xcc][g][0];
        unsigned* __restrict__ bf = &st->bflag[xcc][g][0];
        const size_t base_a = (size_t) (tile * 16) * 256 + c1 * 16 + i1; // this work-item's 16 pass-A outputs: base_a + 256 b
        float2 bw[16];
#pragma unroll
        for (int b = 0; b < 16; ++b)
            bw[b] = big_tw[base_a + b * 256];

        float nre[16], nim[16];
        unsigned f = gg;
        if (f < batch)
            g256_load_a<false> (nre, nim, in + (size_t) f * in_stride + tile * 16, t0, c0);
#if CFFT_GROUP_EXPERIMENT
        const bool stamping = (dbg & 32u) != 0 && gg == 0 && tile == 0;
#endif
        for (unsigned k = 0; f < batch; f += total, ++k)
        {
            float2 v[16];
            CFFT_GROUP_STAMP (0)
#if CFFT_GROUP_EXPERIMENT
            if (dbg & 32u)
                asm volatile ("s_waitcnt vmcnt(16)" ::: "memory"); // the operands are here (the previous frame's stores may be out)
#endif
            CFFT_GROUP_STAMP (1)
#pragma unroll
            for (int s = 0; s < 16; ++s)
            {
                if constexpr (INV)
                {
                    // the swap HERE, behind an opaque statement: written at the request (or written here without the statement --
                    // the optimiser turns the 32 loop-carried scalars into 16 loop-carried pairs and builds the pairs at the
                    // end of the previous iteration, i.e. right behind the loads) it is an instruction that reads the load's
                    // result, so every wave waits for its prefetch where it issued it (vmcnt is one in-order counter: and for
                    // everything requested before it).  That cost the backward instantiation 6-12 % in round 3.
                    float a = nre[s], c = nim[s];
                    asm volatile ("" : "+v"(a), "+v"(c));
                    v[s] = CFFT_GROUP_INV_CONJ ? make_float2 (a, -c) : make_float2 (c, a);
                }
                else
                    v[s] = make_float2 (nre[s], nim[s]);
            }
            // ---- pass A: columns [16 tile, 16 tile + 16) of frame f
            fft256_two_stage_lds (v, tw1, i1, smem, c0, t0, c1, i1);
            float2* __restrict__ slot = slots + (size_t) (two_slots != 0 ? (k & 1u) : 0u) * N;
            CFFT_GROUP_STAMP (2)
            if (two_slots == 0 && k != 0 && ! CFFT_GROUP_DBG (1u))
            {
                // every member has read frame k - 1 out of the slot (each wave looks for itself: no barrier needed)
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
            if (! CFFT_GROUP_DBG (2u))
            {
                if (wt)
                {
#pragma unroll
                    for (int b = 0; b < 16; ++b)
                        slot_store_wt (&slot[base_a + b * 256], cmul (v[b], bw[b])); // sc1: written through
                }
                else
                {
#pragma unroll
                    for (int b = 0; b < 16; ++b)
                        slot[base_a + b * 256] = cmul (v[b], bw[b]); // plain store: stays in this XCD's L2
                }
            }
            CFFT_GROUP_STAMP (3)
            // The next frame's pass-A operands.  `early`: requested HERE, in front of the hand-off: the wait for the rest of the
            // group then has this workgroup's share of the input stream in flight (vector memory operations retire in issue
            // order, so wave 0's flag poll returns behind its own prefetch -- with `w0late` wave 0 therefore asks for its share
            // behind the pass-B loads instead).  Otherwise: behind the pass-B loads.
            const unsigned fn = f + total;
            const bool pre_here = early && fn < batch && ! CFFT_GROUP_DBG (8u) && ! (w0late && tid < 64);
            if (pre_here)
            {
                g256_load_a<false> (nre, nim, in + (size_t) fn * in_stride + tile * 16, t0, c0);
                asm volatile ("s_waitcnt vmcnt(16)" ::: "memory"); // all but the 16 prefetch loads: this wave's stores are in L2
            }
            else
                asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores are in L2
            CFFT_GROUP_STAMP (4)
#if CFFT_GROUP_WAVE_FLAGS
            // Round 4: the hand-off by WAVE instead of by workgroup.  A wave's stores are acknowledged (the wait above): it
            // publishes them itself -- word 4 tile + wave -- and polls the group's 64 words with ONE load, lane l word l.  No
            // workgroup barrier on either side of the hand-off (it had two): a wave goes on as soon as the 64 producers of
            // the frame have published, whatever its sister waves are doing.  Same protocol, same scopes, finer grain.
            if ((tid & 63) == 0)
                flag_store (&af[4u * tile + ((unsigned) tid >> 6)], k + 1u);
            CFFT_GROUP_STAMP (5)
            if (! CFFT_GROUP_DBG (1u))
            {
                unsigned n = 0;
                while (__all (ld_relaxed (af + (tid & 63)) >= k + 1u) == 0)
                {
                    __builtin_amdgcn_s_sleep (2);
                    if (++n > spin_limit)
                    {
                        if ((tid & 63) == 0)
                            group_fail (st, host_err, seq, 1u);
                        break;
                    }
                }
            }
            CFFT_GROUP_STAMP (6)
            CFFT_GROUP_STAMP (7)
#else
            __syncthreads();
            if (tid == 0)
                flag_store_as (&af[tile], k + 1u, wt);
            CFFT_GROUP_STAMP (5)
            const float2* __restrict__ srcb_i = slot + tile * 256 + tid;
            if (incr)
            {
                // Round 5 (form bit 4): the hand-off member by member.  Block s of this member's pass-B tile is member s's pass-A
                // tile, so its load depends on member s's flag alone: every wave polls the group's 16 flags for itself (one 64-byte
                // line, lane l word l & 15) and requests the blocks of the members that have published, until it has all 16 -- the
                // slowest member then delays one sixteenth of the tile instead of all of it.  (Vector memory operations return in
                // issue order, so a poll issued behind block loads comes back behind them: the loop cannot spin faster than the L2.)
                unsigned done = 0, n = 0;
                while (done != 0xFFFFu)
                {
                    const unsigned fl = ld_relaxed (af + (tid & 15));
                    const unsigned ready = (unsigned) __ballot (fl >= k + 1u) & 0xFFFFu;
                    const unsigned fresh = ready & ~done;
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        if ((fresh >> s) & 1u)
                            v[s] = gload_l2 (&srcb_i[s * 16 * 256]);
                    done |= fresh;
                    if (fresh == 0)
                    {
                        __builtin_amdgcn_s_sleep (1);
                        if (++n > spin_limit)
                        {
                            if ((tid & 63) == 0)
                                group_fail (st, host_err, seq, 1u);
                            break;
                        }
                    }
                }
            }
            else if (tid < 64 && ! CFFT_GROUP_DBG (1u))
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
            CFFT_GROUP_STAMP (6)
            if (! incr)
                __syncthreads();
            CFFT_GROUP_STAMP (7)
#endif
            // ---- pass B: rows [16 tile, 16 tile + 16) of the slot -> transposed store into frame f
            const float2* __restrict__ srcb = slot + tile * 256 + tid;
            if (! CFFT_GROUP_DBG (4u) && ! incr)
            {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    v[s] = gload_l2 (&srcb[s * 16 * 256]); // block (a_hi = s, kb_hi = tile), element tid
            }
            // ... behind the pass-B loads and ahead of everything else of this frame
            const bool pre_late = fn < batch && ! CFFT_GROUP_DBG (8u) && ! pre_here;
            const float2* __restrict__ nsrc = in + (size_t) fn * in_stride + tile * 16;
            // `spread`: a quarter of them here, the others between the steps of the pass-B transform (its arithmetic and LDS
            // traffic then run while the memory pipeline drains, instead of behind one burst of 32 loads)
            if (pre_late && ! spread)
                g256_load_a<false> (nre, nim, nsrc, t0, c0);
            if (pre_late && spread)
                g256_load_a_part<false, 0> (nre, nim, nsrc, t0, c0);
            CFFT_PHASE_FENCE();
            CFFT_GROUP_STAMP (8)
#if CFFT_GROUP_EXPERIMENT
            if (dbg & 32u)
            {
                if (pre_late && spread)
                    asm volatile ("s_waitcnt vmcnt(4)" ::: "memory");
                else if (pre_late)
                    asm volatile ("s_waitcnt vmcnt(16)" ::: "memory"); // the pass-B operands are here (the prefetch may be out)
                else
                    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            }
#endif
            CFFT_GROUP_STAMP (9)
            // fft256_two_stage_lds, opened up for the flag
            fft_reg<16> (v);
            CFFT_PHASE_FENCE();
            if (pre_late && spread)
                g256_load_a_part<false, 1> (nre, nim, nsrc, t0, c0);
            CFFT_PHASE_FENCE();
            __syncthreads(); // the previous exchange's reads are finished; every wave's pass-B loads have returned
            if (tid == 0 && two_slots == 0)
                flag_store_as (&bf[tile], k + 1u, wt);
#pragma unroll
            for (int b = 0; b < 16; ++b)
                smem[kbl * kFS256 + 17 * tb + b] = v[b];
            CFFT_PHASE_FENCE();
            if (pre_late && spread)
                g256_load_a_part<false, 2> (nre, nim, nsrc, t0, c0);
            CFFT_PHASE_FENCE();
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 16; ++a)
                v[a] = smem[kbl * kFS256 + tb + 17 * a];
            CFFT_PHASE_FENCE();
            if (pre_late && spread)
                g256_load_a_part<false, 3> (nre, nim, nsrc, t0, c0);
            CFFT_PHASE_FENCE();
#pragma unroll
            for (int a = 1; a < 16; ++a)
                v[a] = cmul (v[a], tw1[(a - 1) * 16 + tb]);
            fft_reg<16> (v);
            CFFT_GROUP_STAMP (10)
            float2* __restrict__ dst = out + (size_t) f * out_stride + tile * 16 + kbl;
            if (! CFFT_GROUP_DBG (16u))
            {
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    gstore (&dst[(unsigned) (tb + 16 * b) * 256u], INV ? (CFFT_GROUP_INV_CONJ ? make_float2 (v[b].x, -v[b].y) : make_float2 (v[b].y, v[b].x)) : v[b]);
            }
            else if (v[0].x == 12345.678f) // keeps the arithmetic alive
                gstore (&dst[0], v[1]);
            CFFT_GROUP_STAMP (11)
        }
        return;
    }

    // ---- SOLO mode: whole frames by one workgroup, its own slot, no dependency on anybody
    float2* __restrict__ slot = work + (size_t) blockIdx.x * N;
    for (;;)
    {
        __syncthreads(); // everybody has read info[6] of the previous round
        if (tid == 0)
            info[6] = __hip_atomic_fetch_add (&st->next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned f = info[6];
        if (f >= batch)
            break;
#pragma unroll 1
        for (unsigned tile = 0; tile < 16; ++tile)
        {
            float re[16], im[16];
            g256_load_a<INV> (re, im, in + (size_t) f * in_stride + tile * 16, t0, c0);
            const size_t base_a = (size_t) (tile * 16) * 256 + c1 * 16 + i1;
            float2 bw[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                bw[b] = big_tw[base_a + b * 256];
            float2 v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = make_float2 (re[s], im[s]);
            fft256_two_stage_lds (v, tw1, i1, smem, c0, t0, c1, i1);
#pragma unroll
            for (int b = 0; b < 16; ++b)
                slot[base_a + b * 256] = cmul (v[b], bw[b]);
        }
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); // every wave's stores are in L2 (or beyond) ...
        __syncthreads();                                   // ... before any wave reads the slot back, past its L1
#pragma unroll 1
        for (unsigned tile = 0; tile < 16; ++tile)
        {
            float2 v[16];
            const float2* __restrict__ srcb = slot + tile * 256 + tid;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = gload_l2 (&srcb[s * 16 * 256]);
            fft256_two_stage_lds (v, tw1, tb, smem, kbl, tb, kbl, tb);
            float2* __restrict__ dst = out + (size_t) f * out_stride + tile * 16 + kbl;
#pragma unroll
            for (int b = 0; b < 16; ++b)
                gstore (&dst[(unsigned) (tb + 16 * b) * 256u], INV ? (CFFT_GROUP_INV_CONJ ? make_float2 (v[b].x, -v[b].y) : make_float2 (v[b].y, v[b].x)) : v[b]);
        }
        // the slot is rewritten by the next frame's pass A only after the barrier at the loop top, which every wave reaches
        // with its pass-B loads consumed
    }
}
// =====================================================================================================
// Fused convolution of N = 65536 complex frames in ONE launch on the same groups:
//     y = BACKWARD (FORWARD (x) * H * scaling)          (the call sequence of reference test/test.cpp:161-166)
// The backward transform is the four-step algorithm run the other way round, which lets it start where the forward one
// ends: with n = a + 256 b and k = 256 ka + kb,
//     y[a + 256 b] = sum_kb W_256^(-b kb) [ W_N^(-a kb) sum_ka Y[256 ka + kb] W_256^(-a ka) ]
// and a pass-B work-item that has just produced X[256 ka + kb] for ka = tb + 16 b' of its row kb holds exactly the operands
// -- in exactly the order -- that the inner sum over ka needs from it.  So per frame and member (tile T of every phase):
//     A   columns of x -> slot (block T)                        hand-off 1 (as in group256_kernel)
//     B   rows of the slot -> X;  * H * scaling;  inverse 256-point transform over ka in the same registers;
//         * conj W_N^(a kb)  -> slot (block T again: a member only ever writes its own block)    hand-off 2
//     A'  rows kb of the slot for 16 columns a -> inverse 256-point transform over kb -> y
// The frame is read once and written once; the two intermediates stay in the XCD's L2 (written through, read back from L2).
// H, both sets of inter-pass twiddles and the filter bins a work-item multiplies are the same for every frame: VGPRs.
// Flags per frame: hand-off 1 stored / read, hand-off 2 stored / read (each member's counters; a block is rewritten only
// after every member has read the previous contents out of it).  SOLO mode (fall-back, see above): two slots per workgroup.
template <bool UNUSED = false>
__global__ __launch_bounds__ (256, 2) void group256_conv_kernel (const float2* __restrict__ in,
                                                                 float2* __restrict__ work, // max (kGroupSlotsTotal, 2 * gridDim.x) slots of 65536 float2
                                                                 float2* __restrict__ out,
                                                                 const float2* __restrict__ H,
                                                                 const float2* __restrict__ stage_tw,
                                                                 const float2* __restrict__ big_tw,
                                                                 GroupState* __restrict__ st,
                                                                 unsigned* __restrict__ host_err,
                                                                 unsigned batch,
                                                                 float scaling,
                                                                 unsigned seq,
                                                                 unsigned reg_polls,
                                                                 unsigned force_solo)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    __shared__ unsigned info[8];
    __shared__ float2 tw1[15 * 16];
    constexpr unsigned N = 65536;
    const int tid = threadIdx.x;
    const int lo = tid & 15, hi = tid >> 4;
    const unsigned spin_limit = group_spin_limit (force_solo);
    force_solo &= 1u; // (bit 1, the listed hand-off form, is not offered by this kernel: capi.cpp runs three launches instead)
    if (tid < 240)
        tw1[tid] = stage_tw[Plan256::tw_offset (1) + tid];
    if (tid == 0)
        group_register (st, info, reg_polls, force_solo);
    __syncthreads();
    const unsigned mode = info[0];

// one tile of each phase, shared by both modes.  v in / out as described at the call sites.
#define CFFT_GCONV_PASS_A(V, TILE, SLOT, BWA)                                                       \
    {                                                                                               \
        fft256_two_stage_lds (V, tw1, lo, smem, lo, hi, hi, lo);                                    \
        _Pragma ("unroll") for (int b = 0; b < 16; ++b)                                             \
            (SLOT)[(size_t) (TILE) * 4096u + b * 256 + tid] = cmul (V[b], BWA[b]);                  \
    }
// rows of the slot -> X -> * H -> inverse over ka -> * conj W: V[b'] = J'[kb = 16 TILE + lo][a = hi + 16 b'].  (SOLO mode only: the
// filter bins and twiddles are fetched where they are used, behind a laundered index -- no early requests, no register cost)
#define CFFT_GCONV_PASS_B(V, TILE)                                                                  \
    {                                                                                               \
        fft256_two_stage_lds (V, tw1, hi, smem, lo, hi, lo, hi);                                    \
        int tq_ = tid;                                                                              \
        CFFT_LAUNDER_INDEX (tq_);                                                                   \
        _Pragma ("unroll") for (int b = 0; b < 16; ++b)                                             \
        {                                                                                           \
            const float2 h_ = H[(unsigned) ((tq_ >> 4) + 16 * b) * 256u + (TILE) * 16 + (tq_ & 15)]; \
            const float2 y_ = cmul (V[b], make_float2 (h_.x * scaling, h_.y * scaling));            \
            V[b] = make_float2 (y_.y, y_.x); /* swapped: operand of the inverse transform */        \
        }                                                                                           \
        fft256_two_stage_lds (V, tw1, hi, smem, lo, hi, lo, hi);                                    \
        int tr_ = tid;                                                                              \
        CFFT_LAUNDER_INDEX (tr_);                                                                   \
        _Pragma ("unroll") for (int b = 0; b < 16; ++b)                                             \
            V[b] = cmulc (make_float2 (V[b].y, V[b].x), big_tw[(size_t) b * 4096u + (TILE) * 256 + tr_]); \
    }
// rows kb of the slot for the tile's 16 columns (V[s] = J'[kb = lo + 16 s][a = 16 TILE + hi]) -> y[b = hi + 16 b'][a = 16 TILE + lo]
#define CFFT_GCONV_PASS_A_INV(V, TILE, DST)                                                         \
    {                                                                                               \
        _Pragma ("unroll") for (int s = 0; s < 16; ++s)                                             \
            V[s] = make_float2 (V[s].y, V[s].x);                                                    \
        fft256_two_stage_lds (V, tw1, hi, smem, hi, lo, lo, hi);                                    \
        _Pragma ("unroll") for (int b = 0; b < 16; ++b)                                             \
            gstore (&(DST)[(unsigned) (hi + 16 * b) * 256u + (TILE) * 16 + lo], make_float2 (V[b].y, V[b].x)); \
    }

    if (mode == 1)
    {
        if (info[5] == 0)
            return;
        const unsigned xcc = info[1], r = info[2], gg = info[3], total = info[4];
        const unsigned g = r / kGroupMembers, tile = r % kGroupMembers;
        float2* __restrict__ slot = work + (size_t) ((xcc * kGroupMaxPerXcc + g) * 2u) * N;
        unsigned* __restrict__ af = &st->aflag[xcc][g][0]; // stored: 2 k + 1 after hand-off 1, 2 k + 2 after hand-off 2 of frame k
        unsigned* __restrict__ bf = &st->bflag[xcc][g][0]; // read:   the same counts
        float2 bwa[16];
#pragma unroll
        for (int b = 0; b < 16; ++b)
            bwa[b] = big_tw[(size_t) tile * 4096u + b * 256 + tid];        // W_N^(a kb), a = 16 tile + hi, kb = lo + 16 b
        float nre[16], nim[16];
        unsigned f = gg;
        if (f < batch)
            g256_load_a<false> (nre, nim, in + (size_t) f * N + tile * 16, hi, lo);
        const int tid_outer = tid;
        for (unsigned k = 0; f < batch; f += total, ++k)
        {
            // the work-item index is laundered per frame: otherwise the ~100 loop-invariant LDS / slot offsets of the three
            // phases are lifted out of the frame loop and held (then spilled) for good (stockham.hpp, CFFT_LAUNDER_INDEX)
            int tid = tid_outer;
            CFFT_LAUNDER_INDEX (tid);
            const int lo = tid & 15, hi = tid >> 4;
            float2 v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = make_float2 (nre[s], nim[s]);
            // ---- A: columns [16 tile, 16 tile + 16) of frame f -> block `tile` of the slot
            fft256_two_stage_lds (v, tw1, lo, smem, lo, hi, hi, lo);
            if (k != 0)
                group_wait<1> (bf, 2u * k, tid & 63, st, host_err, seq, 2u, spin_limit); // every member has read frame k - 1's second intermediate
#pragma unroll
            for (int b = 0; b < 16; ++b)
                slot[(size_t) tile * 4096u + b * 256 + tid] = cmul (v[b], bwa[b]);
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                flag_store (&af[tile], 2u * k + 1u);
            if (tid < 64)
                group_wait<2> (af, 2u * k + 1u, tid, st, host_err, seq, 1u, spin_limit);
            __syncthreads();
            // ---- B: rows [16 tile, 16 tile + 16)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = gload_l2 (&slot[(size_t) s * 4096u + tile * 256 + tid]);
            CFFT_PHASE_FENCE();
            fft_reg<16> (v);
            __syncthreads(); // every wave's loads of the first intermediate have returned
            if (tid == 0)
                flag_store (&bf[tile], 2u * k + 1u);
#pragma unroll
            for (int b = 0; b < 16; ++b)
                smem[lo * kFS256 + 17 * hi + b] = v[b];
            __syncthreads();
            // The member's filter bins (16 per work-item, the same for every frame) and its twiddles of the backward direction
            // do not fit the registers next to the forward twiddles and the prefetched operands: they come from L2 per frame,
            // requested HALF A TRANSFORM ahead of their use.  Their addresses hang on an index laundered at this point, so that
            // the requests cannot be moved further up (with the requests at the phase's start the kernel spilled 94 registers).
            int tq = tid;
            CFFT_LAUNDER_INDEX (tq);
            float2 hr[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                hr[b] = H[(unsigned) ((tq >> 4) + 16 * b) * 256u + tile * 16 + (tq & 15)]; // bin 256 ka + kb, ka = hi + 16 b, kb = 16 tile + lo
#pragma unroll
            for (int a = 0; a < 16; ++a)
                v[a] = smem[lo * kFS256 + hi + 17 * a];
#pragma unroll
            for (int a = 1; a < 16; ++a)
                v[a] = cmul (v[a], tw1[(a - 1) * 16 + hi]);
            fft_reg<16> (v);
#pragma unroll
            for (int b = 0; b < 16; ++b)
            {
                const float2 y = cmul (v[b], make_float2 (hr[b].x * scaling, hr[b].y * scaling));
                v[b] = make_float2 (y.y, y.x); // swapped: operand of the inverse transform over ka
            }
            // the inverse 256-point transform over ka, in the same roles (fft256_two_stage_lds opened up for the requests)
            fft_reg<16> (v);
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 16; ++b)
                smem[lo * kFS256 + 17 * hi + b] = v[b];
            __syncthreads();
            int tr = tid;
            CFFT_LAUNDER_INDEX (tr);
            float2 bwb[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                bwb[b] = big_tw[(size_t) b * 4096u + tile * 256 + tr]; // W_N^(a kb), a = hi + 16 b, kb = 16 tile + lo
#pragma unroll
            for (int a = 0; a < 16; ++a)
                v[a] = smem[lo * kFS256 + hi + 17 * a];
#pragma unroll
            for (int a = 1; a < 16; ++a)
                v[a] = cmul (v[a], tw1[(a - 1) * 16 + hi]);
            fft_reg<16> (v);
            group_wait<1> (bf, 2u * k + 1u, tid & 63, st, host_err, seq, 3u, spin_limit); // every member has read the first intermediate
#pragma unroll
            for (int b = 0; b < 16; ++b)
                slot[(size_t) tile * 4096u + b * 256 + tid] = cmulc (make_float2 (v[b].y, v[b].x), bwb[b]);
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                flag_store (&af[tile], 2u * k + 2u);
            if (tid < 64)
                group_wait<2> (af, 2u * k + 2u, tid, st, host_err, seq, 4u, spin_limit);
            __syncthreads();
            // ---- A': rows kb of the second intermediate for columns [16 tile, 16 tile + 16) -> y
#pragma unroll
            for (int s = 0; s < 16; ++s)
            {
                const float2 x = gload_l2 (&slot[(size_t) s * 4096u + tile * 256 + tid]);
                v[s] = make_float2 (x.y, x.x);
            }
            // the next frame's pass-A operands, behind these loads and ahead of this phase's arithmetic and stores (the filter
            // bins and these operands do not fit the registers together: hence here and not behind the pass-B loads)
            const unsigned fn = f + total;
            if (fn < batch)
                g256_load_a<false> (nre, nim, in + (size_t) fn * N + tile * 16, hi, lo);
            CFFT_PHASE_FENCE();
            fft_reg<16> (v);
            __syncthreads(); // every wave's loads of the second intermediate have returned
            if (tid == 0)
                flag_store (&bf[tile], 2u * k + 2u);
#pragma unroll
            for (int b = 0; b < 16; ++b)
                smem[hi * kFS256 + 17 * lo + b] = v[b];
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 16; ++a)
                v[a] = smem[lo * kFS256 + hi + 17 * a];
#pragma unroll
            for (int a = 1; a < 16; ++a)
                v[a] = cmul (v[a], tw1[(a - 1) * 16 + hi]);
            fft_reg<16> (v);
            float2* __restrict__ dst = out + (size_t) f * N;
#pragma unroll
            for (int b = 0; b < 16; ++b)
                gstore (&dst[(unsigned) (hi + 16 * b) * 256u + tile * 16 + lo], make_float2 (v[b].y, v[b].x));
        }
        return;
    }

    // ---- SOLO mode: whole frames by one workgroup, two slots of its own (first and second intermediate)
    float2* __restrict__ s1 = work + (size_t) (2u * blockIdx.x) * N;
    float2* __restrict__ s2 = s1 + N;
    for (;;)
    {
        __syncthreads();
        if (tid == 0)
            info[6] = __hip_atomic_fetch_add (&st->next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned f = info[6];
        if (f >= batch)
            break;
#pragma unroll 1
        for (unsigned tile = 0; tile < 16; ++tile)
        {
            float re[16], im[16];
            g256_load_a<false> (re, im, in + (size_t) f * N + tile * 16, hi, lo);
            float2 bwa[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                bwa[b] = big_tw[(size_t) tile * 4096u + b * 256 + tid];
            float2 v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = make_float2 (re[s], im[s]);
            CFFT_GCONV_PASS_A (v, tile, s1, bwa)
        }
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (unsigned tile = 0; tile < 16; ++tile)
        {
            float2 v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = gload_l2 (&s1[(size_t) s * 4096u + tile * 256 + tid]);
            CFFT_GCONV_PASS_B (v, tile)
#pragma unroll
            for (int b = 0; b < 16; ++b)
                s2[(size_t) tile * 4096u + b * 256 + tid] = v[b];
        }
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (unsigned tile = 0; tile < 16; ++tile)
        {
            float2 v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = gload_l2 (&s2[(size_t) s * 4096u + tile * 256 + tid]);
            float2* __restrict__ dst = out + (size_t) f * N;
            CFFT_GCONV_PASS_A_INV (v, tile, dst)
        }
    }
#undef CFFT_GCONV_PASS_A
#undef CFFT_GCONV_PASS_B
#undef CFFT_GCONV_PASS_A_INV
}
} // namespace cfft
