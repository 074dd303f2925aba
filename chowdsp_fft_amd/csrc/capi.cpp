// The C-ABI layer: the eleven legacy chowdsp_fft symbols (include/chowdsp_fft.h) and the batched
// device-pointer extension (include/chowdsp_fft_gpu.h).  Host C++ only; kernels live in kernels_*.hip.
//
// Mirrors the reference dispatcher chowdsp_fft.cpp:232-452, with "which SIMD path" replaced by
// "host or device pointers": host pointers are staged through device memory (one frame per call, PCIe
// bound, kept only so existing callers work), device pointers are used in place.  There is no CPU
// compute path in this library.
#include "../../include/chowdsp_fft_gpu.h"

#include "host_common.hpp"
#include "launch.hpp"
#include "plans.hpp"
#include "tables.hpp"

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

using namespace cfft::host;
namespace cf = chowdsp::fft;

namespace
{
constexpr size_t kHostAlign = 64; // reference chowdsp_fft.cpp:63

size_t frame_floats (const Setup* s) { return s->transform == cf::FFT_COMPLEX ? 2 * (size_t) s->N : (size_t) s->N; }

bool is_device_pointer (const void* p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes (&attr, p) != hipSuccess)
    {
        (void) hipGetLastError(); // plain malloc memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// per-thread staging area for the legacy host-pointer path
struct Staging
{
    float* dev = nullptr;
    size_t floats = 0;
    int device = -1;
    ~Staging()
    {
        if (dev != nullptr)
            (void) hipFree (dev);
    }
    float* get (size_t need, int device_now)
    {
        if (need <= floats && device == device_now)
            return dev;
        if (dev != nullptr)
            (void) hipFree (dev);
        dev = nullptr;
        floats = 0;
        if (hipMalloc ((void**) &dev, need * sizeof (float)) != hipSuccess)
        {
            dev = nullptr;
            return nullptr;
        }
        floats = need;
        device = device_now;
        return dev;
    }
};
thread_local Staging g_staging;

// Per-thread PINNED host buffer for short frames on the legacy host-pointer path: the kernel reads its input from and
// writes its result to host memory directly over PCIe (zero-copy), which replaces two DMA submissions and their
// completions by two host memcpys -- the per-call cost is all latency at these sizes.  Longer frames keep the DMA copies.
struct PinnedStaging
{
    float* host = nullptr;
    size_t floats = 0;
    ~PinnedStaging()
    {
        if (host != nullptr)
            (void) hipHostFree (host);
    }
    float* get (size_t need)
    {
        if (need <= floats)
            return host;
        if (host != nullptr)
            (void) hipHostFree (host);
        host = nullptr;
        floats = 0;
        if (hipHostMalloc ((void**) &host, need * sizeof (float), hipHostMallocPortable) != hipSuccess)
        {
            (void) hipGetLastError();
            host = nullptr;
            return nullptr;
        }
        floats = need;
        return host;
    }
};
thread_local PinnedStaging g_pinned;

// CHOWDSP_FFT_ZEROCOPY_MAX_BYTES: longest frame (bytes) that takes the zero-copy legacy path; 0 disables it
size_t zerocopy_max_bytes()
{
    static const size_t n = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_ZEROCOPY_MAX_BYTES");
        return e != nullptr ? (size_t) std::atoll (e) : (size_t) 262144;
    }();
    return n;
}

// grow-only scratch for the two-pass sizes, one per (device, stream); bounded by workspace_cap() so that the
// intermediate of one chunk of frames stays inside the 256 MiB Infinity Cache between pass A and pass B.
constexpr size_t kWorkspaceCapDefault = 256ull << 20;
// option "workspace_mib" (initial value: CHOWDSP_FFT_WORKSPACE_MIB): the calls made from now on chunk their batches against it; a
// workspace that is already larger stays (grow-only).  Results do not depend on it (tests/test_gpu_chunking.py).
static std::atomic<size_t> g_workspace_cap { 0 };
size_t workspace_cap()
{
    size_t cap = g_workspace_cap.load();
    if (cap == 0)
    {
        const char* e = std::getenv ("CHOWDSP_FFT_WORKSPACE_MIB");
        const long v = e != nullptr ? std::atol (e) : 0;
        cap = v > 0 ? (size_t) v << 20 : kWorkspaceCapDefault;
        g_workspace_cap = cap;
    }
    return cap;
}
// Two-pass plans work through their batch in CHUNKS of at most workspace_cap() bytes of intermediate.  Since round 5 the chunks run
// one after the other on the caller's stream, each launch with the whole resident grid (one lane: the default of every family but
// one, lane_count below).
// CHOWDSP_FFT_LANES=2 is the form of rounds 1-4: a two-lane software pipeline, chunk i on lane i%2 (its own internal stream and
// workspace, every launch with HALF the resident grid), so that pass B of one chunk overlaps pass A of the next; the lanes fork from /
// join back into the caller's stream with events (asynchronous, ordered, capturable).  Measured again with round 5's kernels
// (profiles/r05_lanes_sweep.txt): one lane is 3-12 % faster on every chunked plan, and two lanes lose a further 20-45 % whenever the
// runtime maps both lane streams onto ONE hardware queue (a process with several other streams: the half grids then run one after
// the other).
struct Lanes
{
    hipStream_t lane[2] = { nullptr, nullptr };
    hipEvent_t start = nullptr, done[2] = { nullptr, nullptr };
    float2* ws[2] = { nullptr, nullptr };
    size_t bytes = 0; // per workspace
    bool ok = false;
    // fused (one-launch) two-pass plans: queue state + the per-XCD slot ring of intermediates
    void* fused_state = nullptr;
    float2* fused_ws = nullptr;
    size_t fused_ws_bytes = 0;
    // pinned, device-visible: 0, or (launch number << 8) | code stored BY THE KERNEL whose bounded waits ran out (its output
    // is not valid).  Sticky until the host takes it: looked at by the next one-launch call on the stream and by every
    // library synchronisation of the stream (fft_gpu_stream_synchronize, fft_gpu_release_stream_resources, the legacy calls)
    unsigned* fused_err_host = nullptr;
    unsigned fused_seq = 0; // launches so far on this (device, stream)
    // frames beyond 2^21 complex points: one frame of workspace for the transposes
    float2* huge_ws = nullptr;
    size_t huge_ws_bytes = 0;
};
std::recursive_mutex g_ws_mutex; // (recursive: a frame beyond 2^21 points runs its factors' two-pass plans while it holds the lock)
std::map<std::pair<int, hipStream_t>, Lanes> g_lanes;

// CHOWDSP_FFT_PAIR_CHUNK_DIV: how much smaller the chunks of long REAL transforms that run complex two-pass + pair stage are than those of
// complex ones.  Round 5, one lane (profiles/r05_lanes_sweep.txt): 1 -- r2097152 0.222 / r491520 0.205 / r1310720 0.187 of 8 TB/s against
// 0.215 / 0.193 / 0.168 with 2 and 0.214 / 0.186 / 0.172 with 4.  (Rounds 1-4, two lanes: 2.)
size_t pair_chunk_divisor()
{
    static const size_t n = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_PAIR_CHUNK_DIV");
        const long v = e != nullptr ? std::atol (e) : 1;
        return (size_t) (v >= 1 && v <= 64 ? v : 1);
    }();
    return n;
}

// CHOWDSP_FFT_REAL4_CHUNK_DIV: the real four-step's chunks are this many times smaller than the complex two-pass plans' (round 4,
// workspace sweep on MI355X: real N = 2^17 / 2^19 0.305 / 0.289 of 8 TB/s with 256 MiB of workspace, 0.314 / 0.297 with 128 -- a
// real chunk's input and output are half the size of its workspace, so the smaller set stays in the Infinity Cache together)
size_t real4_chunk_divisor()
{
    static const size_t n = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_REAL4_CHUNK_DIV");
        const long v = e != nullptr ? std::atol (e) : 2;
        return (size_t) (v >= 1 && v <= 16 ? v : 2);
    }();
    return n;
}

// lanes of a chunked call: CHOWDSP_FFT_LANES = 1 / 2 for every plan; unset: ONE, except for the LDS-image 256 x 256 passes (N = 65536 complex as
// two launches per chunk, option fused_two_pass = 0), the only family that still measures faster on two lanes (0.400 against 0.382 of 8 TB/s:
// its kernels run below the streaming ceiling on their own and gain from the overlap; profiles/r05_lanes_sweep.txt)
int lane_count (bool lds_image_passes = false)
{
    static const int n = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_LANES");
        const int v = e != nullptr ? std::atoi (e) : 0;
        return (v == 1 || v == 2) ? v : 0;
    }();
    return n != 0 ? n : (lds_image_passes ? 2 : 1);
}

// the two-pass plans whose passes are the LDS-image kernels of both factors (launch_big_c2c: N = 65536 complex as two launches per chunk)
static bool lds_image_passes (const DeviceTables* t)
{
    return t != nullptr && t->mixed_n2 == 0 && t->r2_n == 0 && t->fast_n1 == 0 && t->real4_r1 == 0 && t->huge_r == 0;
}

// caller holds g_ws_mutex
Lanes* get_lanes (int device, hipStream_t stream, size_t bytes_each, int lanes = 1)
{
    Lanes& L = g_lanes[std::make_pair (device, stream)];
    if (lanes == 2 && ! L.ok)
    {
        // (the lane streams and their events exist only once a two-lane call has been made: a process has few hardware queues)
        bool good = hipEventCreateWithFlags (&L.start, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 2 && good; ++k)
            good = hipStreamCreateWithFlags (&L.lane[k], hipStreamNonBlocking) == hipSuccess
                   && hipEventCreateWithFlags (&L.done[k], hipEventDisableTiming) == hipSuccess;
        if (! good)
        {
            set_last_error ("could not create the two-pass pipeline streams");
            return nullptr;
        }
        L.ok = true;
    }
    const bool need_second = lanes == 2 && L.ws[1] == nullptr;
    if (L.bytes < bytes_each || need_second)
    {
        // growing means waiting for the stream and freeing: neither is legal while the stream is being captured into a graph
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (L.ws[0] != nullptr && hipStreamIsCapturing (stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        {
            set_last_error ("two-pass workspace would have to grow during stream capture: run the same call once outside the capture first");
            return nullptr;
        }
        const size_t want = std::max (L.bytes, bytes_each);
        const int count = (lanes == 2 || L.ws[1] != nullptr) ? 2 : 1;
        if (L.bytes < bytes_each)
        {
            for (int k = 0; k < 2; ++k)
            {
                if (L.ws[k] != nullptr)
                {
                    if (L.lane[k] != nullptr)
                        (void) hipStreamSynchronize (L.lane[k]);
                    (void) hipStreamSynchronize (stream);
                    (void) hipFree (L.ws[k]);
                    L.ws[k] = nullptr;
                }
            }
            L.bytes = 0;
        }
        for (int k = 0; k < count; ++k)
        {
            if (L.ws[k] == nullptr && hipMalloc ((void**) &L.ws[k], want) != hipSuccess)
            {
                L.ws[k] = nullptr;
                set_last_error ("workspace allocation failed");
                return nullptr;
            }
        }
        L.bytes = want;
    }
    return &L;
}

// Run-time options of the one-launch two-pass forms (fft_gpu_set_option; initial values from the environment):
//   fused_two_pass   CHOWDSP_FFT_FUSED            which form runs the sizes that have a one-launch form (N = 65536 complex):
//                                                 2 (default, round 3) = ONE launch by static groups of 16 workgroups per XCD
//                                                 (group_twopass_kernels.hpp): 2.24-2.40 ms per 8192 frames, fabric traffic
//                                                 1.38 x the algorithmic bytes (profiles/r03_group_two_pass.txt);
//                                                 1 = one launch by per-XCD ticket queues (fused_twopass_kernels.hpp, round 2):
//                                                 2.9 ms, 1.5-1.6 x;  0 = two launches per chunk (round 1): 2.63-2.76 ms, 2.0 x
//   fused_min_batch  CHOWDSP_FFT_FUSED_MIN_BATCH  smallest batch that takes a one-launch form: its launch registers the whole
//                                                 grid first (about 10 us), so short batches are faster as two launches
//                                                 (group form, MI355X: batch 1: 20 vs 12 us, 64: 34 vs 30, 128: 56 vs 51,
//                                                 256: 90 vs 93, 512: 163 vs 204, 8192: 2243 vs 2628;
//                                                 profiles/r03_group_batch_sweep.txt)
//   fused_lag        CHOWDSP_FFT_FUSED_LAG        ticket form: steps between a frame's pass A and its pass B
//   fused_slots      CHOWDSP_FFT_FUSED_SLOTS      ticket form: workspace slots (frames) per XCD queue, > lag (default lag + 2)
//   group_force_solo / group_reg_polls            group form: force its fall-back mode (tests) / how many polls (about a
//                                                 microsecond each) a workgroup waits for the grid to be complete
// Ticket form, measured (N = 65536 x 8192, profiles/r02_fused_two_pass.txt): lag 3 / 5 slots keeps every intermediate in L2
// (fabric traffic 1.50 x algorithmic) at 3.10 ms per transform; lag 4 / 6 slots 1.58 x at 2.90 ms; lag 5 / 7 slots no longer
// fits L2 (1.9 x).
struct FusedOptions
{
    std::atomic<long> enabled, min_batch, lag, slots;
    FusedOptions()
    {
        const auto env = [] (const char* name, long dflt) {
            const char* e = std::getenv (name);
            return e != nullptr ? std::atol (e) : dflt;
        };
        const long form = env ("CHOWDSP_FFT_FUSED", 2);
        enabled = (form >= 0 && form <= 2) ? form : 2;
        min_batch = std::max (1l, env ("CHOWDSP_FFT_FUSED_MIN_BATCH", 256));
        long l = env ("CHOWDSP_FFT_FUSED_LAG", 4);
        l = (l >= 1 && l <= 6) ? l : 4;
        long sl = env ("CHOWDSP_FFT_FUSED_SLOTS", l + 2);
        sl = (sl > l && sl <= 8) ? sl : l + 2;
        lag = l;
        slots = sl;
    }
};
FusedOptions& fused_options()
{
    static FusedOptions o;
    return o;
}
// "split16384" / CHOWDSP_FFT_SPLIT16384: N = 16384 complex / real N = 32768 through the split-plane kernels (frame in registers,
// exchanges one component at a time; 512 work-items) instead of the interleaved-image plan with streamed twiddles
std::atomic<int> g_split16384 { -1 };
// "split8192": N = 8192 complex / real N = 16384 through the split-plane kernels with 256 work-items (32 x 32 x 8) instead of
// the 512-thread interleaved-image plan; an A/B switch, off by default
std::atomic<int> g_split8192 { 0 };
// "conv_wave2048": fused convolution of 2048-point cores (real N = 4096, complex N = 2048) by ONE WAVEFRONT per frame on the
// split-plane kernels (64 work-items x 32 points, no workgroup barrier) instead of the 128-thread LDS-image plan
std::atomic<int> g_conv_wave2048 { 0 };
// "fused_conv_split": fft_convolve_frames_batch on the split-plane sizes (8192 / 16384 / 32768 complex points, real N = 16384 / 32768 / 65536)
// as ONE launch (default) or as the three-launch composition (0), for A/B timing
std::atomic<int> g_fused_conv_split { 1 };
// "fused_conv_generic": the same switch for mixed-radix frames (one launch, generic_conv_kernels.hpp) -- A/B against the composition
std::atomic<int> g_fused_conv_generic { 1 };
// "fused_conv_group": ... and for N = 65536 complex (one launch on the static groups of the one-launch transform)
std::atomic<int> g_fused_conv_group { 1 };
// "group_fast": N = 2^17 / 2^18 complex through the one-launch group form on the register-direct tiles (A/B against two launches)
std::atomic<int> g_group_fast { 1 };
// "group_real": real N = 131072 in one launch on the static groups, pair stage in registers (group_real_kernels.hpp; A/B against the
// real four-step's two launches per chunk)
std::atomic<int> g_group_real { 1 };
// "huge_columns": frames beyond 2^21 complex points as strided columns (2-D column kernel, twiddle at its stores) x contiguous rows + ONE
// transpose (round 4) instead of the plans of two factors between three transposes (0)
std::atomic<int> g_huge_columns { 1 };
bool split16384_enabled()
{
    int v = g_split16384.load();
    if (v < 0)
    {
        const char* e = std::getenv ("CHOWDSP_FFT_SPLIT16384");
        v = (e != nullptr && e[0] == '0') ? 0 : 1;
        g_split16384 = v;
    }
    return v != 0;
}
// CHOWDSP_FFT_DEVELOPER=1 unlocks the options that can make a launch produce invalid output on purpose ("group_spin_limit"): they are
// process-wide, and a production host must not be able to reach them through the public fft_gpu_set_option by accident
bool developer_options()
{
    static const bool on = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_DEVELOPER");
        return e != nullptr && e[0] == '1';
    }();
    return on;
}
bool fused_enabled() { return fused_options().enabled.load() != 0; }
int fused_form() { return (int) fused_options().enabled.load(); } // 1 = ticket queues, 2 = static groups
size_t fused_min_batch() { return (size_t) fused_options().min_batch.load(); }
int fused_lag() { return (int) fused_options().lag.load(); }
int fused_slots() { return (int) fused_options().slots.load(); }

std::string fused_error_text (unsigned word)
{
    return "one-launch two-pass transform number " + std::to_string (word >> 8) + " on this stream did not complete its hand-offs (code "
           + std::to_string (word & 255u) + "): its output is not valid";
}
// After `stream` has been synchronised: did a one-launch two-pass transform on it report that its bounded waits ran out?
// Takes the report (the word is cleared).  0 = no.
unsigned take_fused_error (hipStream_t stream)
{
    // the report of THIS device's lane only: the same stream handle (the null stream above all) names a different queue on every
    // device, and a synchronisation on device B must neither take nor clear what a launch on device A has reported
    int device = -1;
    if (hipGetDevice (&device) != hipSuccess)
        device = -1;
    std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
    unsigned word = 0;
    for (auto& kv : g_lanes)
        if (kv.first.second == stream && (device < 0 || kv.first.first == device) && kv.second.fused_err_host != nullptr && *kv.second.fused_err_host != 0)
        {
            if (word == 0)
                word = *kv.second.fused_err_host;
            *kv.second.fused_err_host = 0;
        }
    return word;
}

} // namespace
namespace cfft::host
{
// THE way a synchronous library entry point waits for a stream: hipStreamSynchronize + the hand-off report of the one-launch
// two-pass forms (which validate their results only at a library synchronisation).  A launch whose bounded waits ran out
// turns into FFT_GPU_ELAUNCH here, with its number in fft_gpu_last_error(); the word is cleared, the next call starts clean.
int sync_stream_checked (hipStream_t stream, const char* what)
{
    const hipError_t e = hipStreamSynchronize (stream);
    if (e != hipSuccess)
    {
        set_last_error (what, e);
        return cf::FFT_GPU_ELAUNCH;
    }
    if (const unsigned w = take_fused_error (stream))
    {
        set_last_error (fused_error_text (w).c_str());
        return cf::FFT_GPU_ELAUNCH;
    }
    return cf::FFT_GPU_OK;
}
} // namespace cfft::host
namespace
{
// caller holds g_ws_mutex
Lanes* get_fused (int device, hipStream_t stream, size_t ws_bytes)
{
    Lanes& L = g_lanes[std::make_pair (device, stream)];
    if (L.fused_state == nullptr && hipMalloc (&L.fused_state, std::max (fused_state_bytes(), group_state_bytes())) != hipSuccess)
    {
        L.fused_state = nullptr;
        set_last_error ("fused two-pass state allocation failed");
        return nullptr;
    }
    if (L.fused_err_host == nullptr)
    {
        if (hipHostMalloc ((void**) &L.fused_err_host, sizeof (unsigned), hipHostMallocDefault) != hipSuccess)
        {
            L.fused_err_host = nullptr;
            set_last_error ("fused two-pass status word allocation failed");
            return nullptr;
        }
        *L.fused_err_host = 0;
    }
    if (L.fused_ws_bytes < ws_bytes)
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (L.fused_ws != nullptr && hipStreamIsCapturing (stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        {
            set_last_error ("two-pass workspace would have to grow during stream capture: run the same call once outside the capture first");
            return nullptr;
        }
        if (L.fused_ws != nullptr)
        {
            (void) hipStreamSynchronize (stream);
            (void) hipFree (L.fused_ws);
            L.fused_ws = nullptr;
            L.fused_ws_bytes = 0;
        }
        if (hipMalloc ((void**) &L.fused_ws, ws_bytes) != hipSuccess)
        {
            L.fused_ws = nullptr;
            set_last_error ("fused two-pass workspace allocation failed");
            return nullptr;
        }
        L.fused_ws_bytes = ws_bytes;
    }
    return &L;
}

// complex length the plan's core transforms
int complex_len (const Setup* s) { return s->transform == cf::FFT_COMPLEX ? s->N : s->N / 2; }

// bytes of ONE frame in a two-pass workspace: what big_c2c_chunked / big_real4_chunked allocate per frame of a chunk and what
// fft_gpu_workspace_bytes reports (one helper, so that the two cannot drift apart: a ragged last column tile of the
// generic-column plans pads the frame, the real four-step keeps r2 / 2 + 1 or + 16 rows)
size_t workspace_frame_bytes (const DeviceTables* t, int nc)
{
    if (t->real4_r1 > 0)
        return (t->real4_fast ? real4_fast_workspace_frame_len (t->real4_r1, t->real4_r2) : real4_workspace_frame_len (t->real4_r1, t->real4_r2)) * sizeof (float2);
    if (t->mixed_n2 > 0)
    {
        const size_t ca = (size_t) t->mixed_ca, m1 = (size_t) t->mixed_n1, m2 = (size_t) t->mixed_n2;
        return ((m1 + ca - 1) / ca) * ca * m2 * sizeof (float2);
    }
    return (size_t) nc * sizeof (float2);
}

// pair: 0 = complex transform; +1 = REAL FORWARD: the pair (split) stage runs on each chunk's output right after its
// pass B, in place; -1 = REAL BACKWARD: the pair (merge) stage in -> out runs on each chunk right before its pass A,
// which then works in place on `out`.  Doing it per chunk (and with smaller chunks) lets the stage find its operands in
// the Infinity Cache instead of re-reading the whole batch from HBM afterwards.
// mul_h (register-direct tile plans, forward only): every output bin times mul_h[bin] * mul_scale (fft_convolve_frames_batch)
int big_c2c_chunked (Setup* s, int nc, bool inverse, const float2* in, float2* out, size_t batch, hipStream_t stream, size_t is2, size_t os2, int pair = 0, const float2* mul_h = nullptr, float mul_scale = 1.f)
{
    const bool mixed = s->tables->mixed_n2 > 0; // long mixed-radix frame: generic column pass + power-of-two row pass
    int n1 = 0, n2 = 0;
    if (! mixed)
        split_large (nc, n1, n2);
    if (! mixed && pair == 0 && fused_enabled() && fused_supported (n1, n2) && batch >= fused_min_batch())
    {
        // both passes in one launch, hand-off inside an XCD's L2: no chunking, no lanes, 1.5 x instead of 2 x the traffic
        std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
        const bool groups = fused_form() == 2;
        const int lag = fused_lag(), slots = fused_slots();
        Lanes* L = get_fused (s->device, stream, groups ? group_workspace_bytes() : fused_workspace_bytes (nc, slots));
        if (L == nullptr)
            return cf::FFT_GPU_ENOMEM;
        if (*L->fused_err_host != 0)
        {
            // a bounded wait inside an EARLIER launch on this stream ran out (its output is not valid): say so now
            const unsigned w = *L->fused_err_host;
            *L->fused_err_host = 0;
            set_last_error (fused_error_text (w).c_str());
            return cf::FFT_GPU_ELAUNCH;
        }
        const unsigned seq = ++L->fused_seq;
        const hipError_t e = groups ? launch_group_c2c (n1, n2, inverse, in, L->fused_ws, out, s->tables->stage_twiddles, s->tables->big_twiddles, L->fused_state, L->fused_err_host, seq, batch, stream, is2, os2)
                                    : launch_fused_c2c (n1, n2, inverse, in, L->fused_ws, out, s->tables->stage_twiddles, s->tables->big_twiddles, L->fused_state, L->fused_err_host, seq, batch, lag, slots, stream, is2, os2);
        if (e != hipSuccess)
        {
            set_last_error ("fused two-pass kernel launch", e);
            return cf::FFT_GPU_ELAUNCH;
        }
        return cf::FFT_GPU_OK;
    }
    if (! mixed && pair == 0 && mul_h == nullptr && fused_form() == 2 && s->tables->fast_n1 > 0 && groupfast_supported (n1, n2) && g_group_fast.load() != 0 && get_group_handoff() == 0
        && batch >= std::max<size_t> (1, fused_min_batch() / 4))
    {
        // N = 2^17 / 2^18 on the register-direct tiles: one launch, static groups, one workgroup per CU (group_fast_kernels.hpp)
        std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
        Lanes* L = get_fused (s->device, stream, groupfast_workspace_bytes (n1, n2));
        if (L == nullptr)
            return cf::FFT_GPU_ENOMEM;
        if (*L->fused_err_host != 0)
        {
            const unsigned w = *L->fused_err_host;
            *L->fused_err_host = 0;
            set_last_error (fused_error_text (w).c_str());
            return cf::FFT_GPU_ELAUNCH;
        }
        const unsigned seq = ++L->fused_seq;
        const hipError_t e = launch_groupfast_c2c (n1, n2, inverse, in, L->fused_ws, out, s->tables->stage_twiddles, s->tables->big_twiddles, L->fused_state, L->fused_err_host, seq, batch, stream, is2, os2);
        if (e != hipSuccess)
        {
            set_last_error ("fused two-pass kernel launch (register-direct tiles)", e);
            return cf::FFT_GPU_ELAUNCH;
        }
        return cf::FFT_GPU_OK;
    }
    const size_t frame_bytes = workspace_frame_bytes (s->tables, nc);
    const int want_lanes = lane_count (lds_image_passes (s->tables));
    // the cap covers ALL the workspaces of the call (one; two in the two-lane form), so that they fit the Infinity Cache together
    size_t chunk = workspace_cap() / (size_t) want_lanes / frame_bytes;
    if (pair != 0)
        chunk /= pair_chunk_divisor(); // the chunk's output has to stay cached next to the two workspaces
    if (chunk < 1)
        chunk = 1;
    if (chunk > batch)
        chunk = batch;
    // One enqueue sequence at a time per process: concurrent host threads sharing a stream (e.g. legacy calls on the
    // default stream) must not interleave "pass A -> workspace -> pass B" of different calls on the same workspace.
    // Only the (microseconds of) enqueueing is serialised; execution order then follows from stream order.
    std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
    Lanes* L = get_lanes (s->device, stream, chunk * frame_bytes, want_lanes);
    if (L == nullptr)
        return cf::FFT_GPU_ENOMEM;
    const size_t nchunks = (batch + chunk - 1) / chunk;
    chunk = (batch + nchunks - 1) / nchunks; // equal chunks (the workspace was sized for the largest): no short last launch
    const int lanes = (nchunks >= 2) ? want_lanes : 1;
    if (lanes == 2)
    {
        CFFT_HIP_TRY (hipEventRecord (L->start, stream), cf::FFT_GPU_ELAUNCH);
        for (int k = 0; k < 2; ++k)
            CFFT_HIP_TRY (hipStreamWaitEvent (L->lane[k], L->start, 0), cf::FFT_GPU_ELAUNCH);
    }
    size_t i = 0;
    for (size_t f0 = 0; f0 < batch; f0 += chunk, ++i)
    {
        const size_t nb = (batch - f0 < chunk) ? batch - f0 : chunk;
        const int k = (lanes == 2) ? (int) (i & 1) : 0;
        const hipStream_t st = (lanes == 2) ? L->lane[k] : stream;
        hipError_t e = hipSuccess;
        const float2* cin = in + f0 * is2;
        size_t cis = is2;
        if (pair < 0)
        {
            e = launch_real_pair (false, cin, out + f0 * os2, s->tables->real_twiddles, (unsigned) nc, nb, st, is2, os2);
            cin = out + f0 * os2;
            cis = os2;
        }
        if (e == hipSuccess)
            e = (mixed && mul_h != nullptr) ? launch_conv4_generic (*s->tables, cin, mul_h, mul_scale, L->ws[k], out + f0 * os2, nb, lanes, st, cis, os2) // (the whole fused convolution: three sweeps)
                : mixed ? launch_mixed_big_c2c (*s->tables, inverse, cin, L->ws[k], out + f0 * os2, nb, lanes, st, cis, os2)
                : s->tables->r2_n > 0
                    ? launch_fast_r2_c2c (s->tables->r2_n, inverse, cin, L->ws[k], out + f0 * os2, s->tables->stage_twiddles, s->tables->big_twiddles, nb, lanes, st, cis, os2, mul_h, mul_scale)
                : s->tables->fast_n1 > 0
                    ? launch_fast_c2c (n1, n2, inverse, cin, L->ws[k], out + f0 * os2, s->tables->stage_twiddles, s->tables->big_twiddles, nb, lanes, st, cis, os2, mul_h, mul_scale)
                    : launch_big_c2c (n1, n2, inverse, cin, L->ws[k], out + f0 * os2, s->tables->stage_twiddles, s->tables->big_twiddles, nb, lanes, st, cis, os2);
        if (e == hipSuccess && pair > 0)
            e = launch_real_pair (true, out + f0 * os2, out + f0 * os2, s->tables->real_twiddles, (unsigned) nc, nb, st, os2, os2);
        if (e != hipSuccess)
        {
            set_last_error ("two-pass kernel launch", e);
            return cf::FFT_GPU_ELAUNCH;
        }
    }
    if (lanes == 2)
    {
        for (int k = 0; k < 2; ++k)
        {
            CFFT_HIP_TRY (hipEventRecord (L->done[k], L->lane[k]), cf::FFT_GPU_ELAUNCH);
            CFFT_HIP_TRY (hipStreamWaitEvent (stream, L->done[k], 0), cf::FFT_GPU_ELAUNCH);
        }
    }
    return cf::FFT_GPU_OK;
}

// real N = 131072 in ONE launch on the static XCD groups (group_real_kernels.hpp): same conditions as the complex N = 65536 form
bool group_real_applies (const Setup* s, size_t batch)
{
    return s->transform == cf::FFT_REAL && s->tables->group_core != nullptr && fused_form() == 2 && g_group_real.load() != 0 && batch >= fused_min_batch();
}
int group_real_transform (Setup* s, bool inverse, const float2* in, float2* out, size_t batch, hipStream_t stream, size_t is2, size_t os2, const float2* mul_h = nullptr, float mul_scale = 1.f)
{
    std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
    Lanes* L = get_fused (s->device, stream, group_workspace_bytes());
    if (L == nullptr)
        return cf::FFT_GPU_ENOMEM;
    if (*L->fused_err_host != 0)
    {
        // a bounded wait inside an EARLIER launch on this stream ran out (its output is not valid): say so now
        const unsigned w = *L->fused_err_host;
        *L->fused_err_host = 0;
        set_last_error (fused_error_text (w).c_str());
        return cf::FFT_GPU_ELAUNCH;
    }
    const unsigned seq = ++L->fused_seq;
    const DeviceTables* core = s->tables->group_core;
    const hipError_t e = launch_group_real (inverse, in, L->fused_ws, out, core->stage_twiddles, core->big_twiddles, s->tables->real_twiddles, L->fused_state, L->fused_err_host, seq, batch, stream, is2, os2, mul_h, mul_scale);
    if (e != hipSuccess)
    {
        set_last_error ("one-launch real two-pass kernel launch", e);
        return cf::FFT_GPU_ELAUNCH;
    }
    return cf::FFT_GPU_OK;
}

// long real frames by the real four-step: the same chunked two-lane pipeline as big_c2c_chunked, two launches per chunk
int big_real4_chunked (Setup* s, bool inverse, const float2* in, float2* out, size_t batch, hipStream_t stream, size_t is2, size_t os2, const float2* mul_h = nullptr, float mul_scale = 1.f)
{
    const DeviceTables* t = s->tables;
    const int r1 = t->real4_r1, r2 = t->real4_r2;
    const size_t frame_bytes = workspace_frame_bytes (t, complex_len (s));
    const int want_lanes = lane_count();
    size_t chunk = workspace_cap() / (size_t) want_lanes / frame_bytes / real4_chunk_divisor();
    if (chunk < 1)
        chunk = 1;
    if (chunk > batch)
        chunk = batch;
    std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
    Lanes* L = get_lanes (s->device, stream, chunk * frame_bytes, want_lanes);
    if (L == nullptr)
        return cf::FFT_GPU_ENOMEM;
    const size_t nchunks = (batch + chunk - 1) / chunk;
    chunk = (batch + nchunks - 1) / nchunks; // equal chunks (the workspace was sized for the largest): no short last launch
    const int lanes = (nchunks >= 2) ? want_lanes : 1;
    if (lanes == 2)
    {
        CFFT_HIP_TRY (hipEventRecord (L->start, stream), cf::FFT_GPU_ELAUNCH);
        for (int k = 0; k < 2; ++k)
            CFFT_HIP_TRY (hipStreamWaitEvent (L->lane[k], L->start, 0), cf::FFT_GPU_ELAUNCH);
    }
    size_t i = 0;
    for (size_t f0 = 0; f0 < batch; f0 += chunk, ++i)
    {
        const size_t nb = (batch - f0 < chunk) ? batch - f0 : chunk;
        const int k = (lanes == 2) ? (int) (i & 1) : 0;
        const hipStream_t st = (lanes == 2) ? L->lane[k] : stream;
        const hipError_t e = t->real4_generic
                                 ? (mul_h != nullptr ? launch_real4_generic_conv (*t, in + f0 * is2, mul_h, mul_scale, L->ws[k], out + f0 * os2, nb, lanes, st, is2, os2)
                                                     : launch_real4_generic (*t, inverse, in + f0 * is2, L->ws[k], out + f0 * os2, nb, lanes, st, is2, os2))
                             : t->real4_fast
                                 ? launch_real4_fast (r1, r2, inverse, in + f0 * is2, L->ws[k], out + f0 * os2, t->stage_twiddles, t->big_twiddles, nb, lanes, st, is2, os2, mul_h, mul_scale)
                                 : launch_real4 (r1, r2, inverse, in + f0 * is2, L->ws[k], out + f0 * os2, t->stage_twiddles, t->big_twiddles, nb, lanes, st, is2, os2);
        if (e != hipSuccess)
        {
            set_last_error ("real four-step kernel launch", e);
            return cf::FFT_GPU_ELAUNCH;
        }
    }
    if (lanes == 2)
    {
        for (int k = 0; k < 2; ++k)
        {
            CFFT_HIP_TRY (hipEventRecord (L->done[k], L->lane[k]), cf::FFT_GPU_ELAUNCH);
            CFFT_HIP_TRY (hipStreamWaitEvent (stream, L->done[k], 0), cf::FFT_GPU_ELAUNCH);
        }
    }
    return cf::FFT_GPU_OK;
}

int run_transform (Setup* s, const float* in, float* out, size_t batch, cf::fft_direction_t dir, hipStream_t stream, size_t in_stride, size_t out_stride);

// Frames beyond 2^21 complex points (tables.hpp split_huge; the reference has no upper bound, simd/chowdsp_fft_impl_common.hpp:168-177):
// N = R x C, the frame as an [R][C] matrix (n = r C + c, k = kr + R kc):
//     X[kr + R kc] = sum_c W_C^(c kc) [ W_N^(c kr) sum_r x[r C + c] W_R^(r kr) ]
// as  transpose -> C batched R-point transforms (the factor's own plan, whatever form it takes) -> transpose with the twiddle ->
// R batched C-point transforms -> transpose.  Five sweeps over a frame of at least 32 MiB: completeness, not speed (0.05-0.08 of
// the HBM roofline).  One frame of workspace per (device, stream); real frames add the stand-alone pair stage.
int huge_transform (Setup* s, bool inverse, const float2* in, float2* out, size_t batch, hipStream_t stream, size_t is2, size_t os2)
{
    const DeviceTables* t = s->tables;
    const unsigned R = (unsigned) t->huge_r, C = (unsigned) t->huge_c;
    const size_t nc = (size_t) R * C;
    const bool real = s->transform == cf::FFT_REAL;
    Setup plan_r { kMagic, (int) R, (int) cf::FFT_COMPLEX, s->device, nullptr, t->huge_plan_r };
    Setup plan_c { kMagic, (int) C, (int) cf::FFT_COMPLEX, s->device, nullptr, t->huge_plan_c };
    const cf::fft_direction_t dir = inverse ? cf::FFT_BACKWARD : cf::FFT_FORWARD;
    std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
    Lanes& L = g_lanes[std::make_pair (s->device, stream)];
    if (L.huge_ws_bytes < nc * sizeof (float2))
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing (stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        {
            set_last_error ("workspace would have to be allocated during stream capture: run the same call once outside the capture first");
            return cf::FFT_GPU_ENOMEM;
        }
        if (L.huge_ws != nullptr)
        {
            (void) hipStreamSynchronize (stream);
            (void) hipFree (L.huge_ws);
        }
        L.huge_ws = nullptr;
        L.huge_ws_bytes = 0;
        if (hipMalloc ((void**) &L.huge_ws, nc * sizeof (float2)) != hipSuccess)
        {
            L.huge_ws = nullptr;
            set_last_error ("workspace allocation failed");
            return cf::FFT_GPU_ENOMEM;
        }
        L.huge_ws_bytes = nc * sizeof (float2);
    }
    float2* W = L.huge_ws;
    for (size_t f = 0; f < batch; ++f)
    {
        const float2* x = in + f * is2;
        float2* y = out + f * os2;
        hipError_t e = hipSuccess;
        if (real && inverse)
        {
            // packed spectrum -> Z (pair stage), then the complex backward transform of Z in place
            e = launch_real_pair (false, x, y, t->real_twiddles, (unsigned) nc, 1, stream, is2, os2);
            x = y;
        }
        if (t->huge2_r > 0 && g_huge_columns.load() != 0)
        {
            // round 4: columns x rows.  The huge2_r-point column transforms run on the frame where it lies (strided tiles of the 2-D column
            // kernel) and multiply by W_N^(k col) on their way into the workspace; the huge2_c-point rows are then contiguous (one batched call, in
            // place); ONE transpose brings X[kr + R kc] from [kr][kc] into natural order -- three sweeps + the row plan's instead of five
            const unsigned R2 = (unsigned) t->huge2_r, C2 = (unsigned) t->huge2_c;
            Setup plan_c2 { kMagic, (int) C2, (int) cf::FFT_COMPLEX, s->device, nullptr, t->huge2_plan_c };
            if (e == hipSuccess)
                e = launch_column_fft (inverse, (int) R2, (int) C2, *t->huge2_cols, W, 1, stream, x, t->huge_tw_lo, t->huge_tw_hi, (unsigned) t->huge_shift);
            if (e != hipSuccess)
            {
                set_last_error ("kernel launch (frame beyond 2^21 points, column pass)", e);
                return cf::FFT_GPU_ELAUNCH;
            }
            const int rc2 = run_transform (&plan_c2, (const float*) W, (float*) W, R2, dir, stream, 0, 0);
            if (rc2 != cf::FFT_GPU_OK)
                return rc2;
            e = launch_transpose (W, y, R2, C2, 0, nullptr, nullptr, 0, stream);
            if (e == hipSuccess && real && ! inverse)
                e = launch_real_pair (true, y, y, t->real_twiddles, (unsigned) nc, 1, stream, os2, os2);
            if (e != hipSuccess)
            {
                set_last_error ("kernel launch (frame beyond 2^21 points)", e);
                return cf::FFT_GPU_ELAUNCH;
            }
            continue;
        }
        float2* A = (x != y) ? y : W; // first intermediate; the second one is the other buffer
        float2* B = (A == y) ? W : y;
        if (e == hipSuccess)
            e = launch_transpose (x, A, R, C, 0, nullptr, nullptr, 0, stream);
        if (e != hipSuccess)
        {
            set_last_error ("kernel launch (frame beyond 2^21 points)", e);
            return cf::FFT_GPU_ELAUNCH;
        }
        int rc = run_transform (&plan_r, (const float*) A, (float*) A, C, dir, stream, 0, 0);
        if (rc != cf::FFT_GPU_OK)
            return rc;
        e = launch_transpose (A, B, C, R, inverse ? -1 : +1, t->huge_tw_lo, t->huge_tw_hi, (unsigned) t->huge_shift, stream);
        if (e != hipSuccess)
        {
            set_last_error ("kernel launch (frame beyond 2^21 points)", e);
            return cf::FFT_GPU_ELAUNCH;
        }
        rc = run_transform (&plan_c, (const float*) B, (float*) B, R, dir, stream, 0, 0);
        if (rc != cf::FFT_GPU_OK)
            return rc;
        if (B == W)
            e = launch_transpose (B, y, R, C, 0, nullptr, nullptr, 0, stream);
        else
        {
            e = launch_transpose (B, W, R, C, 0, nullptr, nullptr, 0, stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync (y, W, nc * sizeof (float2), hipMemcpyDeviceToDevice, stream);
        }
        if (e == hipSuccess && real && ! inverse)
            e = launch_real_pair (true, y, y, t->real_twiddles, (unsigned) nc, 1, stream, os2, os2);
        if (e != hipSuccess)
        {
            set_last_error ("kernel launch (frame beyond 2^21 points)", e);
            return cf::FFT_GPU_ELAUNCH;
        }
    }
    return cf::FFT_GPU_OK;
}

// in_stride / out_stride: floats between consecutive frames (0 = packed); validated by the caller
int run_transform (Setup* s, const float* in, float* out, size_t batch, cf::fft_direction_t dir, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0)
{
    if (batch == 0)
        return cf::FFT_GPU_OK;
    // The reference casts its buffers to SIMD vectors (simd/chowdsp_fft_impl_sse.cpp:1527-1528: 16-byte alignment is part of
    // its contract, a violation is undefined behaviour there).  Here several kernels move frames with 16-byte accesses, and a
    // misaligned device address is a memory fault that takes the process down: refuse it instead.
    if (((reinterpret_cast<uintptr_t> (in) | reinterpret_cast<uintptr_t> (out)) & 15u) != 0)
    {
        set_last_error ("transform: input and output must be 16-byte aligned (as the reference requires of its SIMD buffers)");
        return cf::FFT_GPU_EINVAL;
    }
    const DeviceGuard guard (s->device);
    const bool inverse = (dir == cf::FFT_BACKWARD);
    const DeviceTables* t = s->tables;
    const int nc = complex_len (s);
    const bool lds_resident = (t->core_n == nc);
    const float2* cin = (const float2*) in;
    float2* cout = (float2*) out;
    // strides in float2, always explicit from here on
    const size_t is2 = in_stride ? in_stride / 2 : (size_t) nc, os2 = out_stride ? out_stride / 2 : (size_t) nc;
    hipError_t e = hipSuccess;
    if (t->huge_r > 0)
        return huge_transform (s, inverse, cin, cout, batch, stream, is2, os2);
    if (group_real_applies (s, batch))
        return group_real_transform (s, inverse, cin, cout, batch, stream, is2, os2);
    if (t->real4_r1 > 0)
        return big_real4_chunked (s, inverse, cin, cout, batch, stream, is2, os2);
    if (t->mixed_n2 > 0)
    {
        // long mixed-radix frame (complex length > 20000): two passes; real frames add the pair stage per chunk
        if (s->transform == cf::FFT_COMPLEX)
            return big_c2c_chunked (s, nc, inverse, cin, cout, batch, stream, is2, os2);
        return big_c2c_chunked (s, nc, inverse, cin, cout, batch, stream, is2, os2, inverse ? -1 : +1);
    }
    if (t->gen_ns > 0)
    {
        // N = 2^a 3^b 5^c, not a power of two: generic mixed-radix kernel
        const int mode = (s->transform == cf::FFT_COMPLEX) ? 0 : (inverse ? 2 : 1);
        e = launch_generic (mode, inverse, nc, *t, cin, cout, batch, stream, is2, os2);
    }
    else if (s->transform == cf::FFT_COMPLEX)
    {
        if (! lds_resident)
            return big_c2c_chunked (s, nc, inverse, cin, cout, batch, stream, is2, os2);
        if (nc == cfft::kSplitCoreN)
            e = launch_c2c_split (nc, inverse, cin, cout, t->stage_twiddles, batch, stream, is2, os2);
        else if (nc == 16384 && split16384_enabled())
            e = launch_c2c_split (nc, inverse, cin, cout, t->stage_twiddles + cfft::CorePlan<16384>::type::tw_total, batch, stream, is2, os2);
        else if (nc == 8192 && g_split8192.load() != 0)
            e = launch_c2c_split (nc, inverse, cin, cout, t->stage_twiddles + cfft::CorePlan<8192>::type::tw_total, batch, stream, is2, os2);
        else
            e = launch_c2c (nc, inverse, cin, cout, t->stage_twiddles, batch, stream, is2, os2);
    }
    else if (lds_resident && nc == cfft::kSplitCoreN)
    {
        e = launch_real_split (nc, ! inverse, cin, cout, t->stage_twiddles, t->real_twiddles, batch, stream, is2, os2);
    }
    else if (lds_resident && nc == 16384 && split16384_enabled())
    {
        e = launch_real_split (nc, ! inverse, cin, cout, t->stage_twiddles + cfft::CorePlan<16384>::type::tw_total, t->real_twiddles, batch, stream, is2, os2);
    }
    else if (lds_resident && nc == 8192 && g_split8192.load() != 0)
    {
        e = launch_real_split (nc, ! inverse, cin, cout, t->stage_twiddles + cfft::CorePlan<8192>::type::tw_total, t->real_twiddles, batch, stream, is2, os2);
    }
    else if (lds_resident)
    {
        e = inverse ? launch_c2r (nc, cin, cout, t->stage_twiddles, t->real_twiddles, batch, stream, is2, os2)
                    : launch_r2c (nc, cin, cout, t->stage_twiddles, t->real_twiddles, batch, stream, is2, os2);
    }
    else if (! inverse)
    {
        // long real frames: complex two-pass transform of z[n] = x[2n] + i x[2n+1], then the pair stage in place
        return big_c2c_chunked (s, nc, false, cin, cout, batch, stream, is2, os2, +1);
    }
    else
    {
        // packed spectrum -> Z (pair stage) -> complex two-pass backward, chunk by chunk
        return big_c2c_chunked (s, nc, true, cin, cout, batch, stream, is2, os2, -1);
    }
    if (e != hipSuccess)
    {
        set_last_error ("kernel launch", e);
        return cf::FFT_GPU_ELAUNCH;
    }
    return cf::FFT_GPU_OK;
}

int run_convolve (Setup* s, const float* a, const float* b, float* ab, float scaling, size_t batch, bool b_is_shared, hipStream_t stream)
{
    if (((reinterpret_cast<uintptr_t> (a) | reinterpret_cast<uintptr_t> (b) | reinterpret_cast<uintptr_t> (ab)) & 15u) != 0)
    {
        set_last_error ("convolve: operands must be 16-byte aligned (as the reference requires of its SIMD buffers)");
        return cf::FFT_GPU_EINVAL;
    }
    const DeviceGuard guard (s->device);
    const bool real = (s->transform == cf::FFT_REAL);
    const hipError_t e = launch_conv_mac (real, a, b, ab, scaling, (size_t) complex_len (s), batch, b_is_shared, stream);
    if (e != hipSuccess)
    {
        set_last_error ("convolve kernel launch", e);
        return cf::FFT_GPU_ELAUNCH;
    }
    return cf::FFT_GPU_OK;
}

// ---- call coalescing for the legacy single-frame symbols ---------------------------------------
// A caller that can only make one-frame calls with host pointers pays a launch, two PCIe crossings and a synchronisation
// per frame (about 21 us at N = 4096).  Between fft_gpu_defer_begin() and fft_gpu_defer_flush() the calling thread's
// fft_transform / fft_transform_unordered calls are RECORDED instead; the flush gathers every recorded input into pinned
// memory, runs one batched transform per (plan, direction), and scatters the results.  Semantics stay those of the calls
// executed one after the other: a call whose input or output overlaps the OUTPUT of a recorded call (it depends on it)
// flushes first; inputs are all read before any output is written, and outputs are written in call order.
struct DeferredCall
{
    // the plan BY VALUE: it is immutable and its device tables belong to the process-wide cache (host_common.cpp), so a
    // recorded call stays valid after the caller's fft_destroy_setup -- or, for fft_new_setup_preallocated, after the caller
    // has freed the memory the header lived in (the reference's new_setup -> fft_transform -> destroy sequence inside a
    // deferred region)
    Setup plan;
    const float* in;
    float* out;
    int dir;
};
struct DeferredQueue
{
    bool active = false;
    int sticky_rc = 0; // first failure of an IMPLICIT flush since fft_gpu_defer_begin(): fft_gpu_defer_flush() returns it
    std::vector<DeferredCall> calls;
    size_t floats = 0;
    float* pin = nullptr; // pinned staging: inputs, then outputs
    size_t pin_floats = 0;
    float* dev = nullptr;
    size_t dev_floats = 0;
    int dev_device = -1;
    ~DeferredQueue()
    {
        if (pin != nullptr)
            (void) hipHostFree (pin);
        if (dev != nullptr)
            (void) hipFree (dev);
    }
};
thread_local DeferredQueue g_deferred;
constexpr size_t kDeferredMaxFloats = (size_t) 32 << 20; // 128 MiB of frames: flush by itself beyond that

int deferred_flush()
{
    DeferredQueue& q = g_deferred;
    if (q.calls.empty())
        return cf::FFT_GPU_OK;
    std::vector<DeferredCall> calls;
    calls.swap (q.calls);
    const size_t total = q.floats;
    q.floats = 0;
    // every path out of here gives the recorded calls' table references back
    struct ReleaseRecorded
    {
        const std::vector<DeferredCall>& calls;
        ~ReleaseRecorded()
        {
            for (const DeferredCall& c : calls)
                release_tables (c.plan.tables);
        }
    } release_recorded { calls };
    // groups of calls that share plan (same tables = same device, length and transform) and direction, in order of first appearance
    const auto same_group = [] (const DeferredCall& a, const DeferredCall& b) { return a.plan.tables == b.plan.tables && a.dir == b.dir; };
    std::vector<size_t> groups; // index of each group's first call
    for (size_t i = 0; i < calls.size(); ++i)
    {
        bool seen = false;
        for (const size_t g : groups)
            seen = seen || same_group (calls[g], calls[i]);
        if (! seen)
            groups.push_back (i);
    }
    if (q.pin_floats < 2 * total)
    {
        if (q.pin != nullptr)
            (void) hipHostFree (q.pin);
        q.pin = nullptr;
        q.pin_floats = 0;
        if (hipHostMalloc ((void**) &q.pin, 2 * total * sizeof (float), hipHostMallocDefault) != hipSuccess)
        {
            set_last_error ("fft_gpu_defer_flush: pinned staging allocation failed");
            return cf::FFT_GPU_ENOMEM;
        }
        q.pin_floats = 2 * total;
    }
    // 1. every input, grouped (so that a group is one contiguous batch), before any output is written
    float* pin_in = q.pin;
    float* pin_out = q.pin + total;
    std::vector<size_t> offset (calls.size());
    size_t pos = 0;
    for (const size_t g : groups)
        for (size_t i = 0; i < calls.size(); ++i)
            if (same_group (calls[g], calls[i]))
            {
                const size_t nf = frame_floats (&calls[i].plan);
                std::memcpy (pin_in + pos, calls[i].in, nf * sizeof (float));
                offset[i] = pos;
                pos += nf;
            }
    // 2. one batched transform per group, device memory of the plan's device, copies and kernels on the null stream
    int rc = cf::FFT_GPU_OK;
    pos = 0;
    for (const size_t g : groups)
    {
        Setup* s = &calls[g].plan;
        const int gdir = calls[g].dir;
        const size_t nf = frame_floats (s);
        size_t count = 0;
        for (const DeferredCall& c : calls)
            count += same_group (calls[g], c) ? 1 : 0;
        const DeviceGuard guard (s->device);
        if (q.dev_floats < 2 * count * nf || q.dev_device != s->device)
        {
            if (q.dev != nullptr)
                (void) hipFree (q.dev);
            q.dev = nullptr;
            q.dev_floats = 0;
            if (hipMalloc ((void**) &q.dev, 2 * count * nf * sizeof (float)) != hipSuccess)
            {
                set_last_error ("fft_gpu_defer_flush: device staging allocation failed");
                return cf::FFT_GPU_ENOMEM;
            }
            q.dev_floats = 2 * count * nf;
            q.dev_device = s->device;
        }
        float* din = q.dev;
        float* dout = q.dev + count * nf;
        if (hipMemcpyAsync (din, pin_in + pos, count * nf * sizeof (float), hipMemcpyHostToDevice, nullptr) != hipSuccess)
            rc = cf::FFT_GPU_ELAUNCH;
        if (rc == cf::FFT_GPU_OK)
            rc = run_transform (s, din, dout, count, (cf::fft_direction_t) gdir, nullptr, 0, 0);
        if (rc == cf::FFT_GPU_OK && hipMemcpyAsync (pin_out + pos, dout, count * nf * sizeof (float), hipMemcpyDeviceToHost, nullptr) != hipSuccess)
            rc = cf::FFT_GPU_ELAUNCH;
        bool said = false;
        if (rc == cf::FFT_GPU_OK)
        {
            rc = sync_stream_checked (nullptr, "fft_gpu_defer_flush: synchronisation");
            said = rc != cf::FFT_GPU_OK; // (keeps the hand-off report's text: it names the launch)
        }
        if (rc != cf::FFT_GPU_OK)
        {
            if (! said)
                set_last_error ("fft_gpu_defer_flush: a batched transform failed; the recorded calls' outputs were not written");
            return rc;
        }
        pos += count * nf;
    }
    // 3. results, in call order (a later call that wrote the same buffer wins, as it would have)
    for (size_t i = 0; i < calls.size(); ++i)
        std::memcpy (calls[i].out, pin_out + offset[i], frame_floats (&calls[i].plan) * sizeof (float));
    return cf::FFT_GPU_OK;
}

// a flush nobody asked for by name (a dependent call, a full queue, a call that cannot be recorded, a plan being destroyed):
// its failure has no return value to travel in, so the queue remembers the first one for fft_gpu_defer_flush()
void deferred_flush_implicit()
{
    const int rc = deferred_flush();
    if (rc != cf::FFT_GPU_OK && g_deferred.sticky_rc == cf::FFT_GPU_OK)
        g_deferred.sticky_rc = rc;
}

// true: the call was recorded (deferred mode, host pointers)
bool deferred_record (Setup* s, const float* input, float* output, cf::fft_direction_t direction)
{
    DeferredQueue& q = g_deferred;
    if (! q.active)
        return false;
    const size_t nf = frame_floats (s);
    const char *ib = (const char*) input, *ob = (const char*) output;
    const size_t bytes = nf * sizeof (float);
    // a call that reads or rewrites what a recorded call is going to write depends on it: run what is recorded first
    for (const DeferredCall& c : q.calls)
    {
        const char* cb = (const char*) c.out;
        const size_t cbytes = frame_floats (&c.plan) * sizeof (float);
        const bool in_hits = ib < cb + cbytes && cb < ib + bytes;
        const bool out_hits = ob < cb + cbytes && cb < ob + bytes;
        if (in_hits || (out_hits && c.out != output))
        {
            deferred_flush_implicit();
            break;
        }
    }
    if (q.floats + nf > kDeferredMaxFloats)
        deferred_flush_implicit();
    retain_tables (s->tables); // given back when the flush has run the call (deferred_flush)
    q.calls.push_back (DeferredCall { *s, input, output, (int) direction });
    q.floats += nf;
    return true;
}

// legacy single-frame entry: host pointers are staged, device pointers used in place; synchronous.
// The reference's entry points return void (chowdsp_fft.h:138-160): a failure cannot be reported through the return
// value, so every early return records its reason for fft_gpu_last_error().
void legacy_transform (void* setup, const float* input, float* output, cf::fft_direction_t direction)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr)
    {
        set_last_error ("fft_transform: bad setup handle or NULL buffer");
        return;
    }
    // ONE guard for the whole body: the staging allocation, the copies, the launch and the final synchronisation all
    // have to address the plan's device, whatever device the calling thread has current (a new host thread starts on
    // device 0); the null stream that is synchronised below is a per-device object.
    const DeviceGuard guard (s->device);
    const size_t nf = frame_floats (s);
    const bool in_dev = is_device_pointer (input), out_dev = is_device_pointer (output);
    if (g_deferred.active)
    {
        if (! in_dev && ! out_dev && deferred_record (s, input, output, direction))
            return; // runs at fft_gpu_defer_flush()
        deferred_flush_implicit(); // a call that cannot be recorded keeps its place in the order
    }
    if (in_dev && out_dev)
    {
        if (run_transform (s, input, output, 1, direction, nullptr) == cf::FFT_GPU_OK)
        {
            (void) sync_stream_checked (nullptr, "fft_transform: synchronisation"); // records its reason (void entry point)
        }
        return;
    }
    if (! in_dev && ! out_dev && nf * sizeof (float) <= zerocopy_max_bytes())
    {
        float* pin = g_pinned.get (2 * nf);
        if (pin != nullptr)
        {
            std::memcpy (pin, input, nf * sizeof (float));
            if (run_transform (s, pin, pin + nf, 1, direction, nullptr) != cf::FFT_GPU_OK)
                return; // run_transform recorded the error
            if (sync_stream_checked (nullptr, "fft_transform: synchronisation") != cf::FFT_GPU_OK)
                return;
            std::memcpy (output, pin + nf, nf * sizeof (float));
            return;
        }
    }
    float* stage = g_staging.get (2 * nf, s->device);
    if (stage == nullptr)
    {
        set_last_error ("staging allocation failed");
        return;
    }
    const float* din = input;
    float* dout = output;
    if (! in_dev)
    {
        CFFT_HIP_TRY (hipMemcpyAsync (stage, input, nf * sizeof (float), hipMemcpyHostToDevice, nullptr), (void) 0);
        din = stage;
    }
    if (! out_dev)
        dout = stage + nf;
    if (run_transform (s, din, dout, 1, direction, nullptr) != cf::FFT_GPU_OK)
        return;
    if (! out_dev)
        CFFT_HIP_TRY (hipMemcpyAsync (output, dout, nf * sizeof (float), hipMemcpyDeviceToHost, nullptr), (void) 0);
    (void) sync_stream_checked (nullptr, "fft_transform: synchronisation");
}

// legacy three-operand entry (convolve / accumulate): stage whichever operands live in host memory.
// ab_is_input: the destination is also read (ab += ...), so its old contents must be staged too.
template <class Op>
void legacy_elementwise (Setup* s, const float* a, const float* b, float* ab, size_t nf, Op&& op, bool ab_is_input)
{
    const DeviceGuard guard (s->device); // whole body: see legacy_transform
    if (g_deferred.active)
        deferred_flush_implicit(); // convolve / accumulate are not recorded: they keep their place in the order
    const bool a_dev = is_device_pointer (a), b_dev = is_device_pointer (b), ab_dev = is_device_pointer (ab);
    if (! a_dev && ! b_dev && ! ab_dev && nf * sizeof (float) <= zerocopy_max_bytes())
    {
        // all operands in host memory, short frame: the kernel works on a pinned copy directly (see PinnedStaging)
        float* pin = g_pinned.get (3 * nf);
        if (pin != nullptr)
        {
            float *pa = pin, *pb = pin + nf, *pab = pin + 2 * nf;
            std::memcpy (pa, a, nf * sizeof (float));
            std::memcpy (pb, b, nf * sizeof (float));
            if (ab_is_input)
                std::memcpy (pab, ab, nf * sizeof (float));
            if (op (pa, pb, pab) != cf::FFT_GPU_OK)
                return; // op recorded the error
            CFFT_HIP_TRY (hipStreamSynchronize (nullptr), (void) 0);
            std::memcpy (ab, pab, nf * sizeof (float));
            return;
        }
    }
    float* stage = nullptr;
    if (! (a_dev && b_dev && ab_dev))
    {
        stage = g_staging.get (3 * nf, s->device);
        if (stage == nullptr)
        {
            set_last_error ("staging allocation failed");
            return;
        }
    }
    const float *da = a, *db = b;
    float* dab = ab;
    // the reference allows a, b and ab to alias: stage ab first and reuse it for aliased sources
    if (! ab_dev)
    {
        dab = stage + 2 * nf;
        if (ab_is_input)
            CFFT_HIP_TRY (hipMemcpyAsync (dab, ab, nf * sizeof (float), hipMemcpyHostToDevice, nullptr), (void) 0);
    }
    if (! a_dev)
    {
        CFFT_HIP_TRY (hipMemcpyAsync (stage, a, nf * sizeof (float), hipMemcpyHostToDevice, nullptr), (void) 0);
        da = stage;
    }
    if (! b_dev)
    {
        CFFT_HIP_TRY (hipMemcpyAsync (stage + nf, b, nf * sizeof (float), hipMemcpyHostToDevice, nullptr), (void) 0);
        db = stage + nf;
    }
    if (op (da, db, dab) != cf::FFT_GPU_OK)
        return;
    if (! ab_dev)
        CFFT_HIP_TRY (hipMemcpyAsync (ab, dab, nf * sizeof (float), hipMemcpyDeviceToHost, nullptr), (void) 0);
    CFFT_HIP_TRY (hipStreamSynchronize (nullptr), (void) 0);
}
} // namespace

#pragma GCC visibility push(default)
extern "C"
{
namespace chowdsp::fft
{
// ---- allocation (reference chowdsp_fft.cpp:63-78: 64-byte alignment, original pointer stashed in front)
void* aligned_malloc (size_t nb_bytes)
{
    void* raw = std::malloc (nb_bytes + kHostAlign);
    if (raw == nullptr)
        return nullptr;
    const uintptr_t up = (reinterpret_cast<uintptr_t> (raw) + kHostAlign) & ~(uintptr_t) (kHostAlign - 1);
    void** slot = reinterpret_cast<void**> (up) - 1;
    *slot = raw;
    return reinterpret_cast<void*> (up);
}

void aligned_free (void* p)
{
    if (p != nullptr)
        std::free (*(reinterpret_cast<void**> (p) - 1));
}

// ---- plans -------------------------------------------------------------------------------------
size_t fft_bytes_required (int N, fft_transform_t transform, bool)
{
    // The reference returns 2*Ncvec*V*4 + sizeof(FFT_Setup) = 8N+96 (complex) / 4N+96 (real)
    // (simd/chowdsp_fft_impl_sse.cpp:67-72).  We only need the header but keep the reference's size
    // so that callers sizing buffers for either library are interchangeable.
    const size_t ref = (transform == FFT_COMPLEX ? 8 : 4) * (size_t) (N > 0 ? N : 0) + 96;
    const size_t mine = sizeof (Setup) + kHostAlign;
    return ref > mine ? ref : mine;
}

void* fft_new_setup_preallocated (int N, fft_transform_t transform, void* data, bool)
{
    if (data == nullptr || ! size_supported (N, (int) transform))
        return nullptr;
    DeviceInfo dev;
    if (! current_device_info (dev))
        return nullptr;
    const DeviceTables* tables = acquire_tables (dev.ordinal, N, (int) transform);
    if (tables == nullptr)
        return nullptr;
    // header at the first 16-byte aligned address inside the caller's block
    const uintptr_t up = (reinterpret_cast<uintptr_t> (data) + 15) & ~(uintptr_t) 15;
    auto* s = new (reinterpret_cast<void*> (up)) Setup();
    s->magic = kMagic;
    s->N = N;
    s->transform = (int) transform;
    s->device = dev.ordinal;
    s->owned_block = nullptr;
    s->tables = tables;
    return s;
}

void* fft_new_setup (int N, fft_transform_t transform, bool use_avx_if_available)
{
    if (! size_supported (N, (int) transform))
        return nullptr;
    void* block = aligned_malloc (fft_bytes_required (N, transform, use_avx_if_available));
    if (block == nullptr)
        return nullptr;
    void* h = fft_new_setup_preallocated (N, transform, block, use_avx_if_available);
    if (h == nullptr)
    {
        aligned_free (block);
        return nullptr;
    }
    as_setup (h)->owned_block = block;
    return h;
}

void fft_destroy_setup (void* setup)
{
    Setup* s = as_setup (setup);
    if (s == nullptr)
        return;
    // calls recorded by fft_gpu_defer_begin() regions hold the plan by value and a table reference of their own (DeferredCall):
    // nothing to flush here
    void* block = s->owned_block;
    const DeviceTables* tables = s->tables;
    s->magic = 0;
    s->tables = nullptr;
    // the plan's reference to its device tables (the reference frees everything the setup owns here, sse.cpp:79-82); the last
    // one makes them idle: kept while the idle set fits the cache limit, freed by fft_gpu_trim_tables (host_common.cpp)
    release_tables (tables);
    if (block != nullptr)
        aligned_free (block);
}

int fft_simd_width_bytes (void*) { return 16; }

// ---- legacy single-frame operations --------------------------------------------------------------
void fft_transform (void* setup, const float* input, float* output, float*, fft_direction_t direction)
{
    legacy_transform (setup, input, output, direction);
}

void fft_transform_unordered (void* setup, const float* input, float* output, float*, fft_direction_t direction)
{
    legacy_transform (setup, input, output, direction);
}

void fft_convolve_unordered (void* setup, const float* a, const float* b, float* ab, float scaling)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || a == nullptr || b == nullptr || ab == nullptr)
    {
        set_last_error ("fft_convolve_unordered: bad setup handle or NULL buffer");
        return;
    }
    const size_t nf = frame_floats (s);
    legacy_elementwise (s, a, b, ab, nf, [&] (const float* da, const float* db, float* dab) {
        return run_convolve (s, da, db, dab, scaling, 1, false, nullptr);
    }, true);
}

void fft_accumulate (void* setup, const float* a, const float* b, float* ab, int N)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || a == nullptr || b == nullptr || ab == nullptr || N <= 0)
    {
        set_last_error ("fft_accumulate: bad setup handle, NULL buffer or N <= 0");
        return;
    }
    legacy_elementwise (s, a, b, ab, (size_t) N, [&] (const float* da, const float* db, float* dab) {
        const hipError_t e = launch_accumulate (da, db, dab, (size_t) N, nullptr);
        if (e != hipSuccess)
            set_last_error ("accumulate kernel launch", e);
        return e == hipSuccess ? FFT_GPU_OK : FFT_GPU_ELAUNCH;
    }, false);
}

// ---- extension: device / memory helpers ----------------------------------------------------------
int fft_gpu_device_count (void)
{
    int n = 0;
    if (hipGetDeviceCount (&n) != hipSuccess)
    {
        (void) hipGetLastError();
        return 0;
    }
    return n;
}

int fft_gpu_set_device (int ordinal)
{
    CFFT_HIP_TRY (hipSetDevice (ordinal), FFT_GPU_ENODEV);
    return FFT_GPU_OK;
}

void* fft_gpu_malloc (size_t nb_bytes)
{
    void* p = nullptr;
    CFFT_HIP_TRY (hipMalloc (&p, nb_bytes), nullptr);
    return p;
}

void fft_gpu_free (void* p)
{
    if (p != nullptr)
        (void) hipFree (p);
}

void* fft_gpu_host_malloc (size_t nb_bytes)
{
    void* p = nullptr;
    CFFT_HIP_TRY (hipHostMalloc (&p, nb_bytes, hipHostMallocDefault), nullptr);
    return p;
}

void fft_gpu_host_free (void* p)
{
    if (p != nullptr)
        (void) hipHostFree (p);
}

int fft_gpu_memcpy (void* dst, const void* src, size_t nb_bytes, int kind, void* stream)
{
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    CFFT_HIP_TRY (hipMemcpyAsync (dst, src, nb_bytes, k, (hipStream_t) stream), FFT_GPU_ELAUNCH);
    return FFT_GPU_OK;
}

int fft_gpu_stream_synchronize (void* stream)
{
    // the one-launch two-pass forms validate their hand-offs here: a launch whose bounded waits ran out is reported by the
    // synchronisation that follows it, by number
    return sync_stream_checked ((hipStream_t) stream, "fft_gpu_stream_synchronize");
}

const char* fft_gpu_last_error (void) { return last_error(); }

int fft_gpu_size_supported (int N, fft_transform_t transform) { return size_supported (N, (int) transform) ? 1 : 0; }

int fft_gpu_plan_describe (int N, fft_transform_t transform, int desc[8])
{
    if (desc == nullptr || ! size_supported (N, (int) transform))
        return FFT_GPU_EINVAL;
    const int nc = (transform == FFT_COMPLEX) ? N : N / 2;
    int q1 = 0, q2 = 0;
    if (cfft::host::runs_as_huge (nc))
    {
        // beyond 2^21 complex points (and not the product of two static tile lengths): the plans of two factors + three transposes ("3": more than two passes through device memory)
        if (! split_huge (nc, q1, q2))
            return FFT_GPU_EINVAL;
        desc[0] = nc;
        desc[1] = 3;
        desc[2] = q1;
        desc[3] = q2;
        desc[4] = desc[5] = desc[6] = desc[7] = 0;
        // what executes by default (option "huge_columns" = 1) where the frame splits that way: h2r-point strided COLUMNS (the 2-D column
        // kernel, twiddle at its stores) x h2c-point contiguous rows + ONE transpose -- reported in desc[4], desc[5] (0, 0: only the
        // three-transposes form above exists for this length)
        int h2r = 0, h2c = 0;
        if (g_huge_columns.load() != 0 && cfft::host::split_huge_columns (nc, h2r, h2c))
        {
            desc[4] = h2r;
            desc[5] = h2c;
        }
        return FFT_GPU_OK;
    }
    if (transform == FFT_REAL && cfft::host::split_real4_generic (N, q1, q2))
    {
        // long mixed-radix real frame: the real four-step with run-time stage lists for both cores, N = q1 x q2 (rows x columns)
        desc[0] = nc;
        desc[1] = 2;
        desc[2] = q1;
        desc[3] = q2;
        desc[4] = desc[5] = desc[6] = desc[7] = 0;
        return FFT_GPU_OK;
    }
    if (transform == FFT_REAL && split_real4 (N, q1, q2))
    {
        // long real frame: the four-step algorithm on the N real samples themselves, N = q1 x q2 (rows x columns)
        int rr[4] = { 0, 0, 0, 0 };
        if (! core_radices (q2, rr))
            return FFT_GPU_EINVAL;
        desc[0] = nc;
        desc[1] = 2;
        desc[2] = q1;
        desc[3] = q2;
        for (int i = 0; i < 4; ++i)
            desc[4 + i] = rr[i];
        return FFT_GPU_OK;
    }
    if ((nc & (nc - 1)) != 0)
    {
        int m1 = 0, m2 = 0;
        if (nc > 20000 && ! split_mixed (nc, m1, m2) && split_general (nc, m1, m2))
        {
            // general two-pass form: generic n2-point columns and generic n1-point rows
            desc[0] = nc;
            desc[1] = 2;
            desc[2] = m1;
            desc[3] = m2;
            for (int i = 0; i < 4; ++i)
                desc[4 + i] = 0;
            return FFT_GPU_OK;
        }
        if (nc > 20000 && split_mixed (nc, m1, m2))
        {
            // long mixed-radix frame: two passes, generic n2-point columns then power-of-two n1-point rows (radices of n1)
            int radix1[4] = { 0, 0, 0, 0 };
            if (! core_radices (m1, radix1))
                return FFT_GPU_EINVAL;
            desc[0] = nc;
            desc[1] = 2;
            desc[2] = m1;
            desc[3] = m2;
            for (int i = 0; i < 4; ++i)
                desc[4 + i] = radix1[i];
            return FFT_GPU_OK;
        }
        // mixed-radix plan of the generic kernel: one pass, first four stage radices reported
        std::vector<int> r;
        generic_plan_radices (nc, r);
        desc[0] = nc;
        desc[1] = 1;
        desc[2] = desc[3] = 0;
        for (int i = 0; i < 4; ++i)
            desc[4 + i] = i < (int) r.size() ? r[(size_t) i] : 0;
        return FFT_GPU_OK;
    }
    int n1 = 0, n2 = 0;
    if ((nc >= (1 << 19) && nc <= (1 << 21) && transform == FFT_COMPLEX) || (nc >= (1 << 20) && nc <= (1 << 21)))
    {
        // 2^19 .. 2^21 complex points (and the complex cores of real 2^21 / 2^22): one radix-2 split around the register-direct two-pass
        // transform of the two half-length sub-frames -- reported as rows of 2 x 1024 points, columns of nc / 2048 points; CHOWDSP_FFT_R2 / CHOWDSP_FFT_R2_MIN_LOG2 (A/B switches, tables.cpp) are not reflected here
        desc[0] = nc;
        desc[1] = 2;
        desc[2] = 2048;
        desc[3] = nc / 2048;
        desc[4] = 32; // (as for every two-pass plan: the radices of the n2-point column core)
        desc[5] = nc / 2048 / 32;
        desc[6] = 0;
        desc[7] = 0;
        return FFT_GPU_OK;
    }
    if (nc > (1 << 20))
    {
        // powers of two beyond 2^20: the general two-pass form
        if (! split_general (nc, n1, n2))
            return FFT_GPU_EINVAL;
        desc[0] = nc;
        desc[1] = 2;
        desc[2] = n1;
        desc[3] = n2;
        for (int i = 0; i < 4; ++i)
            desc[4 + i] = 0;
        return FFT_GPU_OK;
    }
    split_large (nc, n1, n2);
    int radix[4] = { 0, 0, 0, 0 };
    if (n1 == 256 && n2 == 256)
        radix[0] = radix[1] = 16; // the 65536 fast path has its own radix-16 x 16 plan
    else if (nc == cfft::kSplitCoreN)
        radix[0] = radix[1] = radix[2] = 32; // split-plane single pass
    else if (! core_radices (n2 != 0 ? n2 : nc, radix))
        return FFT_GPU_EINVAL;
    desc[0] = nc;
    desc[1] = (n2 != 0) ? 2 : 1;
    desc[2] = n1;
    desc[3] = n2;
    for (int i = 0; i < 4; ++i)
        desc[4 + i] = radix[i];
    return FFT_GPU_OK;
}

int fft_gpu_defer_begin (void)
{
    if (! g_deferred.active)
        g_deferred.sticky_rc = FFT_GPU_OK;
    g_deferred.active = true;
    return FFT_GPU_OK;
}

int fft_gpu_defer_flush (void)
{
    int rc = deferred_flush();
    // an implicit flush since fft_gpu_defer_begin() failed: its calls' outputs were not written, and this is the only
    // return value that can say so
    if (g_deferred.sticky_rc != FFT_GPU_OK)
    {
        if (rc == FFT_GPU_OK)
        {
            rc = g_deferred.sticky_rc;
            set_last_error ("fft_gpu_defer_flush: an earlier implicit flush of this deferred region failed; its calls' outputs were not written");
        }
        g_deferred.sticky_rc = FFT_GPU_OK;
    }
    g_deferred.active = false;
    return rc;
}

int fft_gpu_release_stream_resources (void* stream)
{
    // what the two-pass plans cached for (any device, this stream): two helper streams, three events, the workspaces, the
    // one-launch form's queue state.  Call it before destroying a stream (a re-used handle would inherit the entry) or to
    // give the memory back; the next two-pass call on the stream builds a fresh set.
    std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
    int released = 0;
    unsigned pending_error = 0;
    for (auto it = g_lanes.begin(); it != g_lanes.end();)
    {
        if (it->first.second != (hipStream_t) stream)
        {
            ++it;
            continue;
        }
        const DeviceGuard guard (it->first.first);
        Lanes& L = it->second;
        (void) hipStreamSynchronize ((hipStream_t) stream);
        if (L.fused_err_host != nullptr && *L.fused_err_host != 0)
        {
            pending_error = *L.fused_err_host; // the report would be lost with the entry
            *L.fused_err_host = 0;
        }
        for (int k = 0; k < 2; ++k)
        {
            if (L.lane[k] != nullptr)
            {
                (void) hipStreamSynchronize (L.lane[k]);
                (void) hipStreamDestroy (L.lane[k]);
            }
            if (L.done[k] != nullptr)
                (void) hipEventDestroy (L.done[k]);
            if (L.ws[k] != nullptr)
                (void) hipFree (L.ws[k]);
        }
        if (L.start != nullptr)
            (void) hipEventDestroy (L.start);
        if (L.fused_ws != nullptr)
            (void) hipFree (L.fused_ws);
        if (L.fused_state != nullptr)
            (void) hipFree (L.fused_state);
        if (L.fused_err_host != nullptr)
            (void) hipHostFree (L.fused_err_host);
        if (L.huge_ws != nullptr)
            (void) hipFree (L.huge_ws);
        it = g_lanes.erase (it);
        ++released;
    }
    // "give the memory back": the device tables of plans that have all been destroyed go too (fft_gpu_trim_tables)
    (void) trim_tables (-1);
    if (pending_error != 0)
    {
        set_last_error (fused_error_text (pending_error).c_str());
        return FFT_GPU_ELAUNCH;
    }
    return released;
}

int fft_gpu_set_option (const char* name, long value)
{
    if (name == nullptr)
        return FFT_GPU_EINVAL;
    FusedOptions& o = fused_options();
    // the superseded forms exist in EXPERIMENTS=1 builds only (launch.hpp CFFT_EXPERIMENTS): their options are refused otherwise
    const bool experiment_option = std::strcmp (name, "wave4096") == 0 || std::strcmp (name, "split8192") == 0 || std::strcmp (name, "conv_wave2048") == 0
                                   || std::strcmp (name, "fused_lag") == 0 || std::strcmp (name, "fused_slots") == 0
                                   || (std::strcmp (name, "fused_two_pass") == 0 && value == 1);
    if (experiment_option && ! CFFT_EXPERIMENTS)
    {
        set_last_error ("fft_gpu_set_option: this option selects a superseded form that only a library built with EXPERIMENTS=1 contains");
        return FFT_GPU_EINVAL;
    }
    if (std::strcmp (name, "fused_two_pass") == 0 && value >= 0 && value <= 2)
        o.enabled = value;
    else if (std::strcmp (name, "fused_min_batch") == 0 && value >= 1)
        o.min_batch = value;
    else if (std::strcmp (name, "fused_lag") == 0 && value >= 1 && value <= 6)
    {
        o.lag = value;
        if (o.slots.load() <= value || o.slots.load() > 8)
            o.slots = value + 2;
    }
    else if (std::strcmp (name, "fused_slots") == 0 && value > o.lag.load() && value <= 8)
        o.slots = value;
    else if (std::strcmp (name, "group_force_solo") == 0)
        set_group_options (value != 0 ? 1 : 0, -1);
    else if (std::strcmp (name, "group_reg_polls") == 0 && value >= 0 && value <= 1000000)
        set_group_options (-1, (int) value);
    else if (std::strcmp (name, "group_spin_limit") == 0 && value >= 0 && value < 65536 && (value == 0 || developer_options()))
        set_group_spin_limit ((int) value); // developer option (CHOWDSP_FFT_DEVELOPER=1): tests force the hand-off waits to run out (1); 0 = built-in limit
    else if (std::strcmp (name, "wave4096") == 0)
        set_wave4096 (value != 0);
    else if (std::strcmp (name, "split16384") == 0)
        g_split16384 = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "split8192") == 0)
        g_split8192 = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "conv_wave2048") == 0)
        g_conv_wave2048 = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "fused_conv_split") == 0)
        g_fused_conv_split = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "fused_conv_generic") == 0)
        g_fused_conv_generic = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "fused_conv_group") == 0)
        g_fused_conv_group = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "group_fast") == 0)
        g_group_fast = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "group_real") == 0)
        g_group_real = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "huge_columns") == 0)
        g_huge_columns = value != 0 ? 1 : 0;
    else if (std::strcmp (name, "real4_generic") == 0 && value >= 0 && value <= 2)
        cfft::host::set_real4_generic ((int) value); // plans MADE from now on: long mixed-radix real frames by the real four-step with generic cores
    else if (std::strcmp (name, "group_handoff") == 0 && (value == 0 || value == 1))
        set_group_handoff ((int) value); // 1: the hand-off form the CDNA4 guide lists, for the one-launch forms that offer it (chowdsp_fft_gpu.h)
    else if (std::strcmp (name, "workspace_mib") == 0 && value >= 1 && value <= (1 << 16))
        g_workspace_cap = (size_t) value << 20; // chunks of the two-pass plans: at most this much intermediate per chunk
    else if (std::strcmp (name, "table_cache_mib") == 0 && value >= 0 && value <= (1 << 20))
        set_table_cache_limit ((size_t) value << 20); // device tables of destroyed plans kept for re-use: at most this much
    else
    {
        set_last_error ("fft_gpu_set_option: unknown option or value out of range");
        return FFT_GPU_EINVAL;
    }
    return FFT_GPU_OK;
}

long fft_gpu_get_option (const char* name)
{
    if (name == nullptr)
        return -1;
    FusedOptions& o = fused_options();
    if (std::strcmp (name, "experiments") == 0)
        return CFFT_EXPERIMENTS ? 1 : 0; // 1: the library carries the superseded forms (built with EXPERIMENTS=1)
    if (std::strcmp (name, "fused_two_pass") == 0)
        return o.enabled.load();
    if (std::strcmp (name, "fused_min_batch") == 0)
        return o.min_batch.load();
    if (std::strcmp (name, "group_spin_limit") == 0)
        return get_group_spin_limit();
    if (std::strcmp (name, "fused_lag") == 0)
        return o.lag.load();
    if (std::strcmp (name, "fused_slots") == 0)
        return o.slots.load();
    if (std::strcmp (name, "split16384") == 0)
        return split16384_enabled() ? 1 : 0;
    if (std::strcmp (name, "split8192") == 0)
        return g_split8192.load();
    if (std::strcmp (name, "conv_wave2048") == 0)
        return g_conv_wave2048.load();
    if (std::strcmp (name, "fused_conv_split") == 0)
        return g_fused_conv_split.load();
    if (std::strcmp (name, "fused_conv_generic") == 0)
        return g_fused_conv_generic.load();
    if (std::strcmp (name, "fused_conv_group") == 0)
        return g_fused_conv_group.load();
    if (std::strcmp (name, "group_fast") == 0)
        return g_group_fast.load();
    if (std::strcmp (name, "group_real") == 0)
        return g_group_real.load();
    if (std::strcmp (name, "huge_columns") == 0)
        return g_huge_columns.load();
    if (std::strcmp (name, "table_cache_mib") == 0)
        return (long) (table_cache_limit() >> 20);
    if (std::strcmp (name, "workspace_mib") == 0)
        return (long) (workspace_cap() >> 20);
    if (std::strcmp (name, "group_handoff") == 0)
        return get_group_handoff();
    if (std::strcmp (name, "real4_generic") == 0)
        return cfft::host::get_real4_generic();
    return -1;
}

size_t fft_gpu_trim_tables (int device)
{
    return trim_tables (device);
}

int fft_gpu_table_cache_info (int device, size_t info[4])
{
    if (info == nullptr)
        return FFT_GPU_EINVAL;
    const TableCacheStats st = table_cache_stats (device);
    info[0] = st.entries;
    info[1] = st.bytes;
    info[2] = st.idle_entries;
    info[3] = st.idle_bytes;
    return FFT_GPU_OK;
}

int fft_gpu_setup_size (void* setup)
{
    Setup* s = as_setup (setup);
    return s != nullptr ? s->N : FFT_GPU_EINVAL;
}

int fft_gpu_setup_is_real (void* setup)
{
    Setup* s = as_setup (setup);
    return s != nullptr ? (s->transform == FFT_REAL ? 1 : 0) : FFT_GPU_EINVAL;
}

int fft_gpu_setup_device (void* setup)
{
    Setup* s = as_setup (setup);
    return s != nullptr ? s->device : -1;
}

size_t fft_gpu_workspace_bytes (void* setup, size_t batch)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || s->tables->core_n == complex_len (s) || (s->tables->gen_ns > 0 && s->tables->mixed_n2 == 0))
        return 0;
    if (s->tables->huge_r > 0)
    {
        // one frame for the transposes + what the two factors' plans cache for their batches of huge_c / huge_r frames
        Setup plan_r { kMagic, s->tables->huge_r, (int) FFT_COMPLEX, s->device, nullptr, s->tables->huge_plan_r };
        Setup plan_c { kMagic, s->tables->huge_c, (int) FFT_COMPLEX, s->device, nullptr, s->tables->huge_plan_c };
        size_t inner = std::max (fft_gpu_workspace_bytes (&plan_r, (size_t) s->tables->huge_c), fft_gpu_workspace_bytes (&plan_c, (size_t) s->tables->huge_r));
        if (s->tables->huge2_r > 0)
        {
            // (the columns x rows form: its rows' plan for a batch of huge2_r frames)
            Setup plan_c2 { kMagic, s->tables->huge2_c, (int) FFT_COMPLEX, s->device, nullptr, s->tables->huge2_plan_c };
            inner = std::max (inner, fft_gpu_workspace_bytes (&plan_c2, (size_t) s->tables->huge2_r));
        }
        return (size_t) complex_len (s) * sizeof (float2) + inner;
    }
    const size_t frame_bytes = workspace_frame_bytes (s->tables, complex_len (s));
    int n1 = 0, n2 = 0;
    if (s->tables->mixed_n2 == 0)
        split_large (complex_len (s), n1, n2);
    if (group_real_applies (s, batch))
        return group_workspace_bytes() + std::max (fused_state_bytes(), group_state_bytes());
    if (s->transform == FFT_COMPLEX && fused_enabled() && fused_supported (n1, n2) && batch >= fused_min_batch())
        return (fused_form() == 2 ? group_workspace_bytes() : fused_workspace_bytes (complex_len (s), fused_slots())) + std::max (fused_state_bytes(), group_state_bytes());
    // (the same conditions as big_c2c_chunked's register-direct group branch, N = 2^17 / 2^18)
    if (s->transform == FFT_COMPLEX && s->tables->mixed_n2 == 0 && fused_form() == 2 && s->tables->fast_n1 > 0 && groupfast_supported (n1, n2) && g_group_fast.load() != 0 && get_group_handoff() == 0
        && batch >= std::max<size_t> (1, fused_min_batch() / 4))
        return groupfast_workspace_bytes (n1, n2) + std::max (fused_state_bytes(), group_state_bytes());
    const int lanes = lane_count (lds_image_passes (s->tables));
    size_t chunk = workspace_cap() / (size_t) lanes / frame_bytes;
    if (s->transform == FFT_REAL && s->tables->real4_r1 == 0)
        chunk /= pair_chunk_divisor(); // the pair stage's operands share the cache with the two workspaces
    if (s->transform == FFT_REAL && s->tables->real4_r1 != 0)
        chunk /= real4_chunk_divisor(); // (big_real4_chunked)
    if (chunk < 1)
        chunk = 1;
    if (chunk > batch)
        chunk = batch;
    return (size_t) lanes * chunk * frame_bytes;
}

// ---- extension: batched operations ---------------------------------------------------------------
int fft_transform_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr)
        return FFT_GPU_EINVAL;
    return run_transform (s, input, output, batch, direction, (hipStream_t) stream);
}

// frames embedded in larger buffers.  Strides in floats.
int fft_transform_batch_strided (void* setup, const float* input, size_t in_stride, float* output, size_t out_stride, size_t batch, fft_direction_t direction, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr)
        return FFT_GPU_EINVAL;
    const size_t nf = frame_floats (s);
    const auto bad_stride = [nf] (size_t st) { return st < nf || (st % 4) != 0 || st > 2 * kMaxFrameStride; };
    if (bad_stride (in_stride) || bad_stride (out_stride)
        || (reinterpret_cast<uintptr_t> (input) % 16) != 0 || (reinterpret_cast<uintptr_t> (output) % 16) != 0)
    {
        set_last_error ("fft_transform_batch_strided: strides must be multiples of 4 floats, >= the frame length, buffers 16-byte aligned");
        return FFT_GPU_EINVAL;
    }
    if (batch > 0)
    {
        // overlapping buffers are only meaningful as the same frames in place
        const float* in_end = input + (batch - 1) * in_stride + nf;
        const float* out_end = output + (batch - 1) * out_stride + nf;
        const bool overlap = input < out_end && output < in_end;
        if (overlap && ! (input == output && in_stride == out_stride))
        {
            set_last_error ("fft_transform_batch_strided: input and output overlap without being the same frames");
            return FFT_GPU_EINVAL;
        }
    }
    return run_transform (s, input, output, batch, direction, (hipStream_t) stream, in_stride, out_stride);
}

// 2-D.  Row pass = the ordinary batched transform of batch * rows frames of `cols` points; column pass in place.
int fft_transform_2d_batch (void* setup, int rows, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr || rows < 1)
        return FFT_GPU_EINVAL;
    if (s->transform != FFT_COMPLEX)
    {
        set_last_error ("fft_transform_2d_batch: needs a COMPLEX setup (N = the row length)");
        return FFT_GPU_EINVAL;
    }
    if (batch > ((size_t) 1 << 40) / (size_t) rows)
        return FFT_GPU_EINVAL;
    const DeviceGuard guard (s->device);
    const DeviceTables* ct = rows > 1 ? acquire_column_tables (s->device, rows) : nullptr;
    if (rows > 1 && ct == nullptr)
    {
        set_last_error ("fft_transform_2d_batch: the number of rows must be 2^a 3^b 5^c and at most 2559");
        return FFT_GPU_EINVAL;
    }
    // the call's own reference to the column tables goes back on every path (idle tables stay cached; freeing them waits for
    // the device, so the column pass launched below is never left reading freed memory)
    struct ReleaseColumns
    {
        const DeviceTables* t;
        ~ReleaseColumns() { release_tables (t); }
    } release_columns { ct };
    const int rc = run_transform (s, input, output, batch * (size_t) rows, direction, (hipStream_t) stream);
    if (rc != FFT_GPU_OK || rows == 1)
        return rc;
    const hipError_t e = launch_column_fft (direction == FFT_BACKWARD, rows, s->N, *ct, (float2*) output, batch, (hipStream_t) stream);
    if (e != hipSuccess)
    {
        set_last_error ("fft_transform_2d_batch: column pass launch", e);
        return FFT_GPU_ELAUNCH;
    }
    return FFT_GPU_OK;
}

int fft_transform_unordered_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream)
{
    return fft_transform_batch (setup, input, output, batch, direction, stream);
}

int fft_convolve_unordered_batch (void* setup, const float* dft_a, const float* dft_b, float* dft_ab, float scaling, size_t batch, int b_is_shared, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || dft_a == nullptr || dft_b == nullptr || dft_ab == nullptr)
        return FFT_GPU_EINVAL;
    return run_convolve (s, dft_a, dft_b, dft_ab, scaling, batch, b_is_shared != 0, (hipStream_t) stream);
}

int fft_accumulate_batch (void* setup, const float* a, const float* b, float* ab, size_t count, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || a == nullptr || b == nullptr || ab == nullptr)
        return FFT_GPU_EINVAL;
    if (((reinterpret_cast<uintptr_t> (a) | reinterpret_cast<uintptr_t> (b) | reinterpret_cast<uintptr_t> (ab)) & 15u) != 0)
    {
        set_last_error ("fft_accumulate_batch: operands must be 16-byte aligned");
        return FFT_GPU_EINVAL;
    }
    const DeviceGuard guard (s->device);
    const hipError_t e = launch_accumulate (a, b, ab, count, (hipStream_t) stream);
    if (e != hipSuccess)
    {
        set_last_error ("accumulate kernel launch", e);
        return FFT_GPU_ELAUNCH;
    }
    return FFT_GPU_OK;
}

int fft_convolve_frames_batch (void* setup, const float* x, const float* dft_filter, float* y, float scaling, size_t batch, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || x == nullptr || dft_filter == nullptr || y == nullptr)
        return FFT_GPU_EINVAL;
    if (((reinterpret_cast<uintptr_t> (x) | reinterpret_cast<uintptr_t> (dft_filter) | reinterpret_cast<uintptr_t> (y)) & 15u) != 0)
    {
        set_last_error ("fft_convolve_frames_batch: buffers must be 16-byte aligned");
        return FFT_GPU_EINVAL;
    }
    const DeviceGuard guard (s->device);
    const DeviceTables* t = s->tables;
    const int nc = complex_len (s);
    if (g_fused_conv_split.load() != 0 && t->core_n == nc && t->gen_ns == 0 && (nc == cfft::kSplitCoreN || nc == 16384 || nc == 8192 || (nc == 2048 && g_conv_wave2048.load() != 0)))
    {
        // frame in registers (split_plane_kernels.hpp): both cores and the multiplication in one launch
        const float2* tw = (nc == 16384)  ? t->stage_twiddles + cfft::CorePlan<16384>::type::tw_total
                           : (nc == 8192) ? t->stage_twiddles + cfft::CorePlan<8192>::type::tw_total
                           : (nc == 2048) ? t->stage_twiddles + cfft::CorePlan<2048>::type::tw_total
                                          : t->stage_twiddles;
        const hipError_t es = launch_conv_frames_split (nc, s->transform == FFT_REAL, (const float2*) x, (const float2*) dft_filter, (float2*) y, tw, t->real_twiddles, scaling, batch, (hipStream_t) stream);
        if (es != hipSuccess)
        {
            set_last_error ("fused convolution kernel launch (split-plane)", es);
            return FFT_GPU_ELAUNCH;
        }
        return FFT_GPU_OK;
    }
    if (s->transform == FFT_COMPLEX && nc == 65536 && t->fast_n1 == 0 && t->mixed_n2 == 0 && fused_form() == 2 && batch >= fused_min_batch() && g_fused_conv_group.load() != 0 && get_group_handoff() == 0)
    {
        // N = 65536 complex: both transforms and the product in ONE launch on the static groups (group_twopass_kernels.hpp)
        std::lock_guard<std::recursive_mutex> lock (g_ws_mutex);
        Lanes* L = get_fused (s->device, (hipStream_t) stream, group_workspace_bytes());
        if (L == nullptr)
            return FFT_GPU_ENOMEM;
        if (*L->fused_err_host != 0)
        {
            const unsigned w = *L->fused_err_host;
            *L->fused_err_host = 0;
            set_last_error (fused_error_text (w).c_str());
            return FFT_GPU_ELAUNCH;
        }
        const unsigned seq = ++L->fused_seq;
        const hipError_t eg = launch_group_conv ((const float2*) x, (const float2*) dft_filter, L->fused_ws, (float2*) y, t->stage_twiddles, t->big_twiddles, L->fused_state, L->fused_err_host, seq, scaling, batch, (hipStream_t) stream);
        if (eg != hipSuccess)
        {
            set_last_error ("fused convolution kernel launch (N = 65536, static groups)", eg);
            return FFT_GPU_ELAUNCH;
        }
        return FFT_GPU_OK;
    }
    if (s->transform == FFT_COMPLEX && (t->fast_n1 > 0 || t->r2_n > 0) && t->mixed_n2 == 0 && g_fused_conv_group.load() != 0)
    {
        // N = 2^17 ... 2^20 complex (register-direct two-pass tiles): the product rides on the forward transform's pass-B
        // stores, the backward transform follows in place -- four sweeps over the batch instead of five
        const int rc = big_c2c_chunked (s, nc, false, (const float2*) x, (float2*) y, batch, (hipStream_t) stream, (size_t) nc, (size_t) nc, 0, (const float2*) dft_filter, scaling);
        if (rc != FFT_GPU_OK)
            return rc;
        return run_transform (s, y, y, batch, FFT_BACKWARD, (hipStream_t) stream);
    }
    if (group_real_applies (s, batch) && g_fused_conv_group.load() != 0)
    {
        // real N = 131072 on the static groups: the product on the forward kernel's stores (pair stage and product in registers), the backward
        // transform in place -- two launches
        const int rc = group_real_transform (s, false, (const float2*) x, (float2*) y, batch, (hipStream_t) stream, (size_t) nc, (size_t) nc, (const float2*) dft_filter, scaling);
        if (rc != FFT_GPU_OK)
            return rc;
        return run_transform (s, y, y, batch, FFT_BACKWARD, (hipStream_t) stream);
    }
    if (s->transform == FFT_COMPLEX && t->mixed_n2 > 0 && conv4_generic_applies (*t) && g_fused_conv_generic.load() != 0)
    {
        // long mixed-radix complex frames whose two factors are both at most 639 points: columns forward, ONE fused row pass (forward rows,
        // product, backward rows, in place in the workspace), columns backward -- three sweeps (conv4_generic_kernels.hpp)
        return big_c2c_chunked (s, nc, false, (const float2*) x, (float2*) y, batch, (hipStream_t) stream, (size_t) nc, (size_t) nc, 0, (const float2*) dft_filter, scaling);
    }
    if (s->transform == FFT_REAL && t->real4_generic != 0 && g_fused_conv_generic.load() != 0)
    {
        // long mixed-radix real frames on the real four-step with generic cores: columns forward, ONE fused row pass (forward rows, product,
        // backward rows, in place in the workspace), columns backward -- three sweeps (real4_generic_kernels.hpp real4g_rows_conv_kernel)
        return big_real4_chunked (s, false, (const float2*) x, (float2*) y, batch, (hipStream_t) stream, (size_t) nc, (size_t) nc, (const float2*) dft_filter, scaling);
    }
    if (s->transform == FFT_REAL && t->real4_r1 > 0 && t->real4_fast != 0 && g_fused_conv_group.load() != 0)
    {
        // real N = 2^17 ... 2^20 (real four-step on the same tiles): the same, with the packed spectrum's bin 0 as two real products
        const int rc = big_real4_chunked (s, false, (const float2*) x, (float2*) y, batch, (hipStream_t) stream, (size_t) nc, (size_t) nc, (const float2*) dft_filter, scaling);
        if (rc != FFT_GPU_OK)
            return rc;
        return run_transform (s, y, y, batch, FFT_BACKWARD, (hipStream_t) stream);
    }
    if (t->gen_ns > 0 && t->mixed_n2 == 0 && g_fused_conv_generic.load() != 0)
    {
        // mixed-radix frame that stays on chip (up to 20000 complex points): forward, product and backward in one launch
        const hipError_t eg = launch_generic_conv (s->transform == FFT_REAL, nc, *t, (const float2*) x, (const float2*) dft_filter, (float2*) y, scaling, batch, (hipStream_t) stream);
        if (eg == hipSuccess)
            return FFT_GPU_OK;
        if (eg != hipErrorNotSupported)
        {
            set_last_error ("fused convolution kernel launch (mixed radix)", eg);
            return FFT_GPU_ELAUNCH;
        }
        (void) hipGetLastError(); // no one-launch form for this plan: the composition below
    }
    if (t->core_n != nc || nc > cfft::kMaxFusedN)
    {
        // No fused kernel for this size (two-pass, 8192-point and mixed-radix plans): the same three steps as separate
        // launches, with `y` holding the spectra in between -- forward x -> y, y = y * H * scaling, backward in place.
        int rc = run_transform (s, x, y, batch, FFT_FORWARD, (hipStream_t) stream);
        if (rc != FFT_GPU_OK)
            return rc;
        const hipError_t em = launch_conv_mac (s->transform == FFT_REAL, y, dft_filter, y, scaling, (size_t) nc, batch, true, (hipStream_t) stream, false);
        if (em != hipSuccess)
        {
            set_last_error ("fft_convolve_frames_batch: multiply kernel launch", em);
            return FFT_GPU_ELAUNCH;
        }
        return run_transform (s, y, y, batch, FFT_BACKWARD, (hipStream_t) stream);
    }
    const hipError_t e = (s->transform == FFT_COMPLEX)
                             ? launch_conv_frames_c2c (nc, (const float2*) x, (const float2*) dft_filter, (float2*) y, t->stage_twiddles, scaling, batch, (hipStream_t) stream)
                             : launch_conv_frames_real (nc, (const float2*) x, (const float2*) dft_filter, (float2*) y, t->stage_twiddles, t->real_twiddles, scaling, batch, (hipStream_t) stream);
    if (e != hipSuccess)
    {
        set_last_error ("fused convolution kernel launch", e);
        return FFT_GPU_ELAUNCH;
    }
    return FFT_GPU_OK;
}
} // namespace chowdsp::fft
} // extern "C"
#pragma GCC visibility pop
