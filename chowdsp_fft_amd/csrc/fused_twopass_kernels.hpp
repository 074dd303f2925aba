// Two-pass (four-step) transforms in ONE launch, with the inter-pass hand-off kept inside an XCD's L2.
//
// multipass_kernels.hpp runs pass A (columns) and pass B (rows) as two launches; the intermediate then crosses the
// fabric twice (written by pass A, read back by pass B) and the frame moves 4 x 8N bytes for 2 x 8N algorithmic ones.
// Here both passes of a frame are executed by workgroups of the SAME XCD, a few microseconds apart: pass A's plain
// stores leave the intermediate resident in that XCD's 4 MiB L2 (they are written through to memory as well -- all
// stored bytes leave L2 on this chip -- but nobody reads that copy), pass B reads it back from L2 with L1-bypassing
// loads.  Fabric traffic per frame: 8N in + 8N intermediate write + 8N out = 1.5 x algorithmic instead of 2 x.
//
// How "the same XCD" is obtained WITHOUT assuming anything about workgroup placement: every workgroup reads the id of
// the XCD it actually runs on (HW_REG_XCC_ID) and works from that XCD's own queue.  A queue is a ticket counter; ticket
// t means step s = t / 2T, slot r = t % 2T (T = tiles per pass):
//        r <  T :  pass A of tile r      of the frame this queue claimed for step s
//        r >= T :  pass B of tile r - T  of the frame this queue claimed for step s - D       (D = lag, in steps)
// Frames are claimed from ONE global counter, one per step, by the workgroup that draws the first ticket of the step
// before, and published in a small tagged ring; an XCD without workgroups claims nothing, a faster XCD claims more.
// All of a frame's work items therefore run on one XCD by construction, whatever the dispatcher did.
//
// Dependencies, all on STRICTLY EARLIER tickets of the same queue (tickets are handed out in order to running
// workgroups, so every wait terminates): pass B of a frame waits for that frame's T pass-A tiles (aflag); pass A of
// step s waits, before it stores, for pass B of the frame that used its workspace slot S steps earlier (bflag).
// Visibility inside the XCD: the producer's waves have their stores acknowledged by L2 (vmcnt), the workgroup
// synchronises, one lane stores the tile's flag; the consumer polls the flags with L1-bypassing loads and reads the
// intermediate with L1-bypassing (sc1) loads only, so no stale L1 line can be hit.
//
// Reference: the same passes for any N, chowdsp_fft.h:131-134; cfftf1_ps simd/chowdsp_fft_impl_sse.cpp:372-432.
#pragma once
#include "multipass_kernels.hpp"

namespace cfft
{
constexpr int kFusedMaxXcc = 16;   // HW_REG_XCC_ID is a 4-bit field
constexpr int kFusedRing = 256;    // claimed-frame ring entries per queue (steps in flight are a handful)
constexpr int kFusedMaxSlots = 8;  // workspace slots (frames) per queue

struct FusedQueue
{
    unsigned ticket;
    unsigned pad0[31];
    unsigned long long claim_word; // (steps claimed so far) << 1 | (1 while a lane is claiming the next one)
    unsigned pad3[30];
    // Completion flags, one word per tile and slot, each written by exactly ONE workgroup with a plain store (which leaves
    // the line in this XCD's L2, where the pollers' L1-bypassing loads find it): the "use count" of the slot, i.e. 1 for the
    // first frame that goes through it, 2 for the second ...  No read-modify-write: atomics execute at the memory side
    // on this chip and would put a fabric round trip (and a hot line) into every item.
    unsigned aflag[kFusedMaxSlots][32]; // [slot][tile]: pass A of the tile has stored (one 128-byte line per slot)
    unsigned bflag[kFusedMaxSlots][32]; // [slot][tile]: pass B of the tile has loaded
    unsigned long long frame_of[kFusedRing]; // (step + 1) << 32 | frame   (frame = kFusedEnd: batch exhausted)
};
struct FusedState
{
    unsigned next_frame; // global claim counter
    unsigned error;      // set if a bounded spin ran out (never observed; keeps a broken launch from hanging the GPU)
    unsigned stat[6];    // CFFT_FUSED_STATS builds only: prefetched / late pass-B items, unresolved items, poll iterations
    unsigned pad[24];
    FusedQueue q[kFusedMaxXcc];
};
constexpr unsigned kFusedEnd = 0xFFFFFFFFu;
constexpr unsigned kFusedSpinLimit = 1u << 18; // polls of ~1 us each: a broken launch ends after a fraction of a second

__device__ __forceinline__ unsigned hw_xcc_id()
{
    // s_getreg_b32 HW_REG_XCC_ID (id 20), bits [3:0]: simm16 = (size - 1) << 11 | offset << 6 | id
    return (unsigned) __builtin_amdgcn_s_getreg ((3 << 11) | (0 << 6) | 20) & 15u;
}

__device__ __forceinline__ unsigned ld_relaxed (const unsigned* p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_relaxed64 (const unsigned long long* p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// L1-bypassing 8-byte load of intermediate data (global_load_dwordx2 ... sc1): served by the XCD's L2
__device__ __forceinline__ float2 gload_l2 (const float2* p)
{
    const unsigned long long v = __hip_atomic_load (reinterpret_cast<const unsigned long long*> (p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2 (__uint_as_float ((unsigned) v), __uint_as_float ((unsigned) (v >> 32)));
}

// one lane: wait until *p >= target (monotonic counter); bounded
__device__ __forceinline__ void spin_until_ge (const unsigned* p, unsigned target, FusedState* st)
{
    unsigned n = 0;
    while (ld_relaxed (p) < target)
    {
        __builtin_amdgcn_s_sleep (8);
        if (++n > kFusedSpinLimit)
        {
            __hip_atomic_store (&st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// What a ticket resolves to.
struct FusedItem
{
    unsigned kind;   // 0 = pass A, 1 = pass B, 2 = nothing (skip), 3 = queue exhausted (exit), 4 = not resolved yet (ticket only)
    unsigned frame;  // global frame
    unsigned tile;
    unsigned slot;   // workspace slot of the frame inside this queue
    unsigned need;   // pass A: every bflag of the slot must have reached this before the stores; pass B: every aflag before the loads
    unsigned ticket;
    unsigned ready;  // pass B: the dependency was already satisfied when the item was resolved (no wait at its start)
    unsigned pad;
};

// PROGRESS RULE (why no wait in here can deadlock).  A workgroup holds up to three tickets (the item it works on and the
// two after it) and processes them in ticket order.  It may BLOCK only for the item it is currently working on, and only
// on things owed by strictly smaller tickets: pass-A tiles of the frame (adone), pass-B tiles of the slot's previous
// frame (bdone), the claim published by the workgroup that drew the first ticket of the step before.  Whatever a ticket
// owes to others (its counter increment, the claim it has to publish) is reached without waiting on anything owed by a
// larger ticket; in particular the look-up of a FUTURE item's frame never spins -- an unpublished claim leaves the item
// "not resolved" (kind 4) and it is looked up again when it becomes the current item -- and claims are not tied to one
// workgroup's progress: whoever needs a frame that has not been claimed yet claims it (fused_advance_claims).  By induction over ticket numbers
// every ticket completes, for any placement and any number of resident workgroups.

// Lane 0: the first ticket of step s claims the frame of step s + 1 (and that of step 0 at the very beginning) and
// publishes it.
// Claims are made in STEP ORDER, by whoever gets there first: `claim_word` holds the number of steps claimed so far and
// a busy bit.  A lane that finds fewer than `upto + 1` steps claimed sets the busy bit with a compare-and-swap, draws the
// next frame from the global counter, publishes it (tagged with its step) and stores the incremented, un-busy word.
// Nothing in the busy section waits for another workgroup, so the bit is always cleared again; a lane that finds it set
// either comes back later (look-ahead) or keeps trying (the current item needs the frame).  Because the draws happen in
// step order, frames grow with the step, and "the batch ran out at step s" implies that it ran out at every later step
// of this queue: the exit test relies on that.
__device__ __forceinline__ void fused_advance_claims (FusedState* st, FusedQueue* q, unsigned upto, unsigned batch, bool keep_trying)
{
    unsigned n = 0;
    for (;;)
    {
        const unsigned long long w = ld_relaxed64 (&q->claim_word);
        const unsigned claimed = (unsigned) (w >> 1);
        if (claimed > upto)
            return;
        if ((w & 1ull) == 0)
        {
            unsigned long long expect = w;
            if (__hip_atomic_compare_exchange_strong (&q->claim_word, &expect, w | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            {
                const unsigned f = __hip_atomic_fetch_add (&st->next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store (&q->frame_of[claimed % kFusedRing], ((unsigned long long) (claimed + 1) << 32) | (f < batch ? f : kFusedEnd),
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store (&q->claim_word, (unsigned long long) (claimed + 1) << 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
        }
        if (! keep_trying)
            return;
        __builtin_amdgcn_s_sleep (2);
        if (++n > kFusedSpinLimit)
        {
            __hip_atomic_store (&st->error, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

// Lane 0, for the first ticket of a step: make sure the frames up to the step AFTER it are claimed (look-ahead: the claim
// is then out long before anybody needs it; whoever needs one that is not, claims it himself in fused_resolve).  Never waits.
template <unsigned TILES>
__device__ __forceinline__ void fused_publish_if_opening (FusedState* st, FusedQueue* q, unsigned t, unsigned batch)
{
    if (t % (2 * TILES) == 0)
        fused_advance_claims (st, q, t / (2 * TILES) + 1, batch, false);
}

// step whose frame ticket t works on; `fill` = a pass-B ticket of the first `lag` steps (nothing to do yet)
template <unsigned TILES>
__device__ __forceinline__ unsigned fused_frame_step (unsigned t, unsigned lag, bool& is_a, bool& fill)
{
    const unsigned s = t / (2 * TILES), r = t % (2 * TILES);
    is_a = r < TILES;
    fill = ! is_a && s < lag;
    return is_a ? s : s - lag;
}

// Lane 0: turn ticket t and its frame_of entry `e` (loaded by the caller) into an item; kind 4 if the entry is not the
// step's yet.
template <unsigned TILES>
__device__ __forceinline__ FusedItem fused_make_item (unsigned t, unsigned long long e, unsigned lag, unsigned slots)
{
    FusedItem it;
    bool is_a, fill;
    const unsigned fs = fused_frame_step<TILES> (t, lag, is_a, fill);
    const unsigned r = t % (2 * TILES);
    it.ticket = t;
    it.ready = 0;
    it.pad = 0;
    it.tile = is_a ? r : r - TILES;
    it.frame = 0;
    it.slot = 0;
    it.need = 0;
    if (fill)
    {
        it.kind = 2;
        return it;
    }
    if ((e >> 32) != (unsigned long long) (fs + 1))
    {
        it.kind = 4;
        return it;
    }
    it.frame = (unsigned) e;
    it.slot = fs % slots;
    if (it.frame == kFusedEnd)
    {
        // the batch ran out at or before step fs.  A pass-B item of an exhausted step means everything older is done:
        // the queue is finished (frames are claimed in step order).  A pass-A item is just skipped: older pass-B work follows.
        it.kind = is_a ? 2u : 3u;
        return it;
    }
    it.kind = is_a ? 0u : 1u;
    it.need = is_a ? fs / slots : fs / slots + 1; // the slot's use count: this frame is its (fs / slots + 1)-th user
    return it;
}

// Lane 0: look ticket t's frame up; `blocking` only for the item the workgroup is about to work on (progress rule)
template <unsigned TILES>
__device__ __forceinline__ FusedItem fused_resolve (FusedState* st, FusedQueue* q, unsigned t, unsigned lag, unsigned slots, bool blocking, unsigned batch)
{
    bool is_a, fill;
    const unsigned fs = fused_frame_step<TILES> (t, lag, is_a, fill);
    unsigned long long e = fill ? 0ull : ld_relaxed64 (&q->frame_of[fs % kFusedRing]);
    FusedItem it = fused_make_item<TILES> (t, e, lag, slots);
    unsigned n = 0;
    while (blocking && it.kind == 4)
    {
        fused_advance_claims (st, q, fs, batch, true); // help: make the claim ourselves if nobody has yet
        e = ld_relaxed64 (&q->frame_of[fs % kFusedRing]);
        it = fused_make_item<TILES> (t, e, lag, slots);
        if (++n > kFusedSpinLimit)
        {
            __hip_atomic_store (&st->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            it.kind = 3;
            break;
        }
    }
    return it;
}

// =====================================================================================================
// N = 65536 = 256 x 256 (BASELINE configuration 5): pass A / pass B bodies of pass_a256_kernel / pass_b256_kernel
// (multipass_kernels.hpp: register-direct, radix-16, one LDS exchange, radix-16), 16 tiles per pass and frame.
// Intermediate layout inside a slot: the same 16 x 16 block order as there.
//
// What an item costs besides its own loads and stores has been taken off the critical path:
//   * the next ticket (the only read-modify-write per item) is drawn when the current item starts, its frame is asked
//     for after the first exchange barrier, and the resolved item -- with a "ready" flag from one look at its
//     dependency flags -- is parked in LDS before the item's last barrier;
//   * a pass-A item does not wait for its stores to be acknowledged: its completion flag is stored during the NEXT item,
//     behind that item's first wait for operands (vector-memory operations retire in issue order, so operands that have
//     arrived imply that the older stores have been acknowledged) and the barrier that follows it.  Before anything that
//     could block, the pending flag is flushed the slow way (the item being waited for may depend on it);
//   * the stage-1 twiddles of the 256-point core live in LDS (pass A and pass B index them differently; both sets in
//     VGPRs cost a workgroup per CU); pass A's inter-pass twiddles are requested together with its operands.
// =====================================================================================================
struct Fused256Shared
{
    FusedItem item[2];    // current / next item: written by lane 0, read by everybody a barrier later
    float2 tw[15 * 16];   // stage-1 twiddles of Plan256: [a - 1][k]
};

#ifndef CFFT_FUSED_STATS
#define CFFT_FUSED_STATS 0
#endif
#ifndef CFFT_FUSED_EXPERIMENT
#define CFFT_FUSED_EXPERIMENT 0
#endif
#if CFFT_FUSED_STATS
#define CFFT_FUSED_COUNT(IDX, N) __hip_atomic_fetch_add (&st->stat[IDX], (unsigned) (N), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define CFFT_FUSED_COUNT(IDX, N) (void) 0
#endif

// All 64 lanes of ONE wave: have the 16 tile flags of a slot reached `need`?  (one 64-byte L2 request; uniform result)
__device__ __forceinline__ bool flags_reached (const unsigned* flags, unsigned need, int lane)
{
    const unsigned f = ld_relaxed (flags + (lane & 15));
    return __all (f >= need) != 0;
}
// ... wait until they have (bounded)
__device__ __forceinline__ void flags_wait (const unsigned* flags, unsigned need, int lane, FusedState* st, int stat)
{
    unsigned n = 0;
    while (! flags_reached (flags, need, lane))
    {
        __builtin_amdgcn_s_sleep (8);
        if (++n > kFusedSpinLimit)
        {
            __hip_atomic_store (&st->error, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    if (n != 0 && lane == 0)
        CFFT_FUSED_COUNT (stat, n);
}
__device__ __forceinline__ void flag_store (unsigned* p, unsigned v)
{
    __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // a plain store, kept from being merged or reordered by the compiler
}

// ---- the hand-off in the form the CDNA4 guide LISTS (option "group_handoff" = 1; MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in
// place of the acquire", first row): every store of the handed-off bytes `sc1` (write-through: the line leaves this XCD's L2), every storing
// wave's s_waitcnt vmcnt(0), the workgroup's barrier, ONE lane's `sc1` flag store; the consumer's `sc1` poll, a barrier, `sc1` loads; one
// workgroup per CU.  The default (fast) form differs in three cells: plain payload and flag stores (the lines STAY in the L2 the same-XCD
// consumers read them from) and two workgroups per CU.  `wt` is uniform over the launch.
__device__ __forceinline__ void flag_store_as (unsigned* p, unsigned v, bool wt)
{
    if (wt)
        __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_store_dword ... sc1
    else
        flag_store (p, v);
}
__device__ __forceinline__ void slot_store_wt (float2* p, float2 v)
{
    const unsigned long long w = (unsigned long long) __float_as_uint (v.x) | ((unsigned long long) __float_as_uint (v.y) << 32);
    __hip_atomic_store (reinterpret_cast<unsigned long long*> (p), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_store_dwordx2 ... sc1
}

template <bool INV>
__global__ __launch_bounds__ (256) void fused256_kernel (const float2* __restrict__ in,
                                                        float2* __restrict__ work, // [kFusedMaxXcc][slots][65536]
                                                        float2* __restrict__ out,
                                                        const float2* __restrict__ stage_tw,
                                                        const float2* __restrict__ big_tw,
                                                        FusedState* __restrict__ st,
                                                        unsigned* __restrict__ host_err, // pinned, device-visible: (seq << 8) | code if a bounded wait ran out
                                                        unsigned seq,
                                                        unsigned batch,
                                                        size_t in_stride,
                                                        size_t out_stride,
                                                        unsigned lag,
                                                        unsigned slots,
                                                        unsigned dbg)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 smem[];
    __shared__ Fused256Shared sh;
    constexpr unsigned n1 = 256, n2 = 256, N = 65536, TILES = 16;
    const int tid = threadIdx.x;
    const unsigned xcc = hw_xcc_id();
    FusedQueue* q = &st->q[xcc];
    float2* __restrict__ wq = work + (size_t) xcc * slots * N;

    // pass A roles (multipass_kernels.hpp): stage 0 lanes run along the tile's 16 columns, stage 1 along the butterfly index
    const int c0 = tid & 15, t0 = tid >> 4;
    const int i1 = tid & 15, c1 = tid >> 4;
    // pass B roles: lanes run along the tile's 16 rows in both stages
    const int kbl = tid & 15, tb = tid >> 4;

    if (tid < 240)
        sh.tw[tid] = stage_tw[Plan256::tw_offset (1) + tid];
    if (tid == 0)
    {
        const unsigned t = __hip_atomic_fetch_add (&q->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fused_publish_if_opening<TILES> (st, q, t, batch);
        sh.item[0] = fused_resolve<TILES> (st, q, t, lag, slots, true, batch);
    }
    __syncthreads();

    // Look-ahead pipeline of lane 0 / wave 0, one item ahead, every step asynchronous:
    //   end of item i - 1 : draw the ticket of item i + 1
    //   start of item i   : (operands of item i requested)  ticket -> ask for its frame
    //   first barrier     : frame -> item i + 1 known; ask for its dependency flags (wave 0)
    //   end of item i     : flags -> "ready"; item i + 1 parked in LDS; draw the ticket of item i + 2
    unsigned tk = 0;             // lane 0: the next ticket
    unsigned long long fe = 0;   // lane 0: its frame_of entry
    FusedItem ni;                // lane 0: the next item
    unsigned nflag = 0;          // wave 0: one of the next item's 16 dependency flags
    unsigned nkind = 2, nneed = 0;
    if (tid == 0)
        tk = __hip_atomic_fetch_add (&q->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a pass-A item whose completion flag has not been stored yet (workgroup-uniform)
    bool pend = false;
    unsigned pend_slot = 0, pend_tile = 0, pend_val = 0;

// the pending flag, the slow way: every wave waits for all of its stores, the workgroup meets, lane 0 stores the flag
#define CFFT_FUSED_FLUSH_PENDING()                                              \
    if (pend)                                                                   \
    {                                                                           \
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");                       \
        __syncthreads();                                                        \
        if (tid == 0)                                                           \
            flag_store (&q->aflag[pend_slot][pend_tile], pend_val);             \
        pend = false;                                                           \
    }
// ... and the fast way, right after a barrier that every wave reached AFTER its current operands had arrived
#define CFFT_FUSED_SIGNAL_PENDING()                                             \
    if (pend)                                                                   \
    {                                                                           \
        if (tid == 0)                                                           \
            flag_store (&q->aflag[pend_slot][pend_tile], pend_val);             \
        pend = false;                                                           \
    }
// look-ahead at the start of an item (lane 0, after the item's operands have been requested): ticket -> ask for its frame
#define CFFT_FUSED_LOOKAHEAD_TICKET()                                           \
    if (tid == 0 && ! (CFFT_FUSED_EXPERIMENT && (dbg & 0x100u)))                \
    {                                                                           \
        fused_publish_if_opening<TILES> (st, q, tk, batch);                     \
        bool is_a_, fill_;                                                      \
        const unsigned fs_ = fused_frame_step<TILES> (tk, lag, is_a_, fill_);   \
        fe = fill_ ? 0ull : ld_relaxed64 (&q->frame_of[fs_ % kFusedRing]);      \
    }
// look-ahead after the first barrier (wave 0): frame -> the next item; ask for its dependency flags
#define CFFT_FUSED_LOOKAHEAD_FRAME()                                                                             \
    if (tid < 64 && ! (CFFT_FUSED_EXPERIMENT && (dbg & 0x100u)))                                                 \
    {                                                                                                            \
        if (tid == 0)                                                                                            \
        {                                                                                                        \
            ni = fused_make_item<TILES> (tk, fe, lag, slots);                                                    \
            if (ni.kind == 4) /* claims are self-service: this depends on nobody's progress */                   \
                ni = fused_resolve<TILES> (st, q, tk, lag, slots, true, batch);                                  \
        }                                                                                                        \
        nkind = __builtin_amdgcn_readfirstlane (ni.kind);                                                        \
        nneed = __builtin_amdgcn_readfirstlane (ni.need);                                                        \
        const unsigned nslot_ = __builtin_amdgcn_readfirstlane (ni.slot);                                        \
        if (nkind == 1)                                                                                          \
            nflag = ld_relaxed (&q->aflag[nslot_][tid & 15]);                                                    \
    }
// look-ahead at the end of an item (wave 0): flags -> "ready", the next item parked in LDS, the ticket after it drawn
#define CFFT_FUSED_LOOKAHEAD_PARK()                                                                              \
    if (tid < 64 && ! (CFFT_FUSED_EXPERIMENT && (dbg & 0x100u)))                                                 \
    {                                                                                                            \
        bool ready_ = false;                                                                                     \
        if (nkind == 1)                                                                                          \
            ready_ = __all (nflag >= nneed) != 0;                                                                \
        if (tid == 0)                                                                                            \
        {                                                                                                        \
            ni.ready = ready_ ? 1u : 0u;                                                                         \
            sh.item[cur ^ 1u] = ni;                                                                              \
            tk = __hip_atomic_fetch_add (&q->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            \
        }                                                                                                        \
    }

#if CFFT_FUSED_EXPERIMENT
    unsigned static_idx = blockIdx.x;
#endif
    for (unsigned cur = 0;; cur ^= 1u)
    {
        FusedItem it = sh.item[cur];
        if (it.kind == 3)
            break;
#if CFFT_FUSED_EXPERIMENT
        if (dbg & 0x100u)
        {
            // no queue at all: pass-A items handed out statically (what the data path alone can do)
            if (dbg & 0x200u)
            {
                // ... both passes, no dependencies (the mix of traffic of the real thing)
                if (static_idx >= batch * 32u)
                    break;
                it.kind = (static_idx % 32u) < 16u ? 0u : 1u;
                it.frame = static_idx / 32u;
                it.tile = static_idx % 16u;
            }
            else
            {
                if (static_idx >= batch * 16u)
                    break;
                it.kind = 0;
                it.frame = static_idx / 16u;
                it.tile = static_idx % 16u;
            }
            it.slot = it.frame % slots;
            static_idx += gridDim.x;
        }
#endif
#if CFFT_FUSED_EXPERIMENT
        // timing experiments only (results are wrong): dbg 1 = pass A items only, 2 = pass B items only, 3 = both; no waiting
        if (((dbg & 15u) == 1 && it.kind == 1) || ((dbg & 15u) == 2 && it.kind == 0))
            it.kind = 2;
        if (dbg != 0)
        {
            it.need = 0;
            it.ready = 1;
        }
#endif
        float2 v[16];
        if (it.kind == 0)
        {
            // ---- pass A: columns [16 tile, 16 tile + 16) of frame it.frame; the inter-pass twiddles come along
            const float2* __restrict__ src = in + (size_t) it.frame * in_stride + it.tile * 16;
#pragma unroll
            for (int s = 0; s < 16; ++s)
            {
                const float2 x = gload (&src[(unsigned) (t0 + 16 * s) * n1 + c0]);
                v[s] = INV ? make_float2 (x.y, x.x) : x;
            }
            const size_t base = (size_t) (it.tile * 16) * 256 + c1 * 16 + i1;
            float2 bw[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                bw[b] = big_tw[base + b * 256];
            CFFT_FUSED_LOOKAHEAD_TICKET()
            fft_reg<16> (v);
            __syncthreads(); // the previous item's stage-1 reads are finished; every wave's older stores are acknowledged
            CFFT_FUSED_SIGNAL_PENDING()
            CFFT_FUSED_LOOKAHEAD_FRAME()
#pragma unroll
            for (int b = 0; b < 16; ++b)
                smem[c0 * kFS256 + 17 * t0 + b] = v[b];
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 16; ++a)
                v[a] = smem[c1 * kFS256 + i1 + 17 * a];
#pragma unroll
            for (int a = 1; a < 16; ++a)
                v[a] = cmul (v[a], sh.tw[(a - 1) * 16 + i1]);
            fft_reg<16> (v);
#pragma unroll
            for (int b = 0; b < 16; ++b)
                v[b] = cmul (v[b], bw[b]);
            if (it.need != 0)
            {
                // the slot's previous frame must have been read completely by its pass B (drawn slots - lag steps ago)
                if (tid < 64)
                    flags_wait (&q->bflag[it.slot][0], it.need, tid, st, 4);
                __syncthreads();
            }
            float2* __restrict__ dst = wq + (size_t) it.slot * N + base;
#pragma unroll
            for (int b = 0; b < 16; ++b)
                dst[b * 256] = v[b]; // plain store: stays in this XCD's L2
            CFFT_FUSED_LOOKAHEAD_PARK()
            pend = true;
            pend_slot = it.slot;
            pend_tile = it.tile;
            pend_val = it.need + 1;
            __syncthreads(); // the next item is visible to everybody
        }
        else if (it.kind == 1)
        {
            // ---- pass B: rows [16 tile, 16 tile + 16) of the slot -> transposed store into frame it.frame
            if (it.ready == 0)
            {
                CFFT_FUSED_FLUSH_PENDING() // what is waited for below may include this workgroup's own last tile
                if (tid < 64)
                    flags_wait (&q->aflag[it.slot][0], it.need, tid, st, 3);
                __syncthreads();
            }
            const float2* __restrict__ src = wq + (size_t) it.slot * N + it.tile * 256 + tid;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = gload_l2 (&src[s * 16 * 256]); // block (a_hi = s, kb_hi = tile), element tid
            CFFT_FUSED_LOOKAHEAD_TICKET()
            fft_reg<16> (v);
            __syncthreads();
            CFFT_FUSED_SIGNAL_PENDING()
            CFFT_FUSED_LOOKAHEAD_FRAME()
#pragma unroll
            for (int b = 0; b < 16; ++b)
                smem[kbl * kFS256 + 17 * tb + b] = v[b];
            __syncthreads();
            // every load of this workgroup has returned (the values went through the exchange): the slot may be rewritten
            if (tid == 0)
                flag_store (&q->bflag[it.slot][it.tile], it.need);
#pragma unroll
            for (int a = 0; a < 16; ++a)
                v[a] = smem[kbl * kFS256 + tb + 17 * a];
#pragma unroll
            for (int a = 1; a < 16; ++a)
                v[a] = cmul (v[a], sh.tw[(a - 1) * 16 + tb]);
            fft_reg<16> (v);
            float2* __restrict__ dst = out + (size_t) it.frame * out_stride + it.tile * 16 + kbl;
#pragma unroll
            for (int b = 0; b < 16; ++b)
                gstore (&dst[(unsigned) (tb + 16 * b) * n2], INV ? make_float2 (v[b].y, v[b].x) : v[b]);
            CFFT_FUSED_LOOKAHEAD_PARK()
            __syncthreads(); // the next item is visible to everybody
        }
        else
        {
            CFFT_FUSED_FLUSH_PENDING()
            CFFT_FUSED_LOOKAHEAD_TICKET()
            CFFT_FUSED_LOOKAHEAD_FRAME()
            CFFT_FUSED_LOOKAHEAD_PARK()
            __syncthreads();
        }
    }
    CFFT_FUSED_FLUSH_PENDING()
    // a bounded wait that ran out anywhere in this launch (its output is not valid) is reported where the host looks
    // after its next synchronisation; several workgroups may store the same value
    if (tid == 0 && host_err != nullptr)
    {
        const unsigned err = ld_relaxed (&st->error);
        if (err != 0)
            __hip_atomic_store (host_err, (seq << 8) | err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#undef CFFT_FUSED_FLUSH_PENDING
#undef CFFT_FUSED_SIGNAL_PENDING
#undef CFFT_FUSED_LOOKAHEAD_FRAME
#undef CFFT_FUSED_LOOKAHEAD_TICKET
#undef CFFT_FUSED_LOOKAHEAD_PARK
}
} // namespace cfft
