// Host-memory streaming mode: many frames in host memory through the GPU with pinned
// buffers and two alternating streams, instead of one staged copy + launch + sync per frame (the legacy symbols).
// Callers it serves: reference bench/bench.cpp:36-45, chowdsp_fft_juce/chowdsp_fft_juce.cpp:32-46 (frames in malloc'd memory).
//
// How caller memory reaches the DMA engines (round 3; replaces a size threshold tied to glibc's mmap behaviour):
//
//   * LARGE buffers are pinned IN PLACE for the duration of the call -- but only the pages that lie ENTIRELY inside the
//     caller's buffer, [ceil (p, page), floor (p + bytes, page)).  The first and the last page of an arbitrary pointer range
//     (a view into a larger array, `ptr + offset`, a block from a pooled allocator, a heap block below the mmap threshold)
//     are shared with whatever else lives there, and pinning / unpinning pages under other code is what made a later,
//     unrelated pageable copy of a NEIGHBOURING heap object fault in round 2 ("Memory access fault ... on address <process
//     heap>", session r02j; tools/repro_pin.py reproduces the sequence).  The partial head and tail (< one page each) travel
//     through small library-owned pinned bounce buffers instead.
//   * SMALL buffers (below `pin_min_bytes`: registering costs more than it saves) never get registered at all: their
//     chunks go through a library-owned, double-buffered pinned staging ring (host copy in, DMA, kernel, DMA, host copy out,
//     the two slots alternating on the two streams).
//
// Either way no page is ever pinned that holds a byte the caller did not hand to this call.
#include "../../include/chowdsp_fft_gpu.h"

#include "host_common.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include <unistd.h>

using namespace cfft::host;

namespace
{
constexpr size_t kChunkBytes = 32ull << 20; // per direction and stream: large enough for PCIe efficiency, small enough to pipeline
constexpr size_t kRingSlotBytes = 4ull << 20; // staging ring slot (small buffers): per direction and stream

struct Pipe
{
    hipStream_t st[2] = { nullptr, nullptr };
    float* din[2] = { nullptr, nullptr };
    float* dout[2] = { nullptr, nullptr };
    size_t floats = 0; // capacity of each device buffer
    // library-owned pinned host memory: the staging ring (two slots per direction) and the bounce buffers for the partial
    // first / last page of a caller buffer (one page each: input head, input tail, output head, output tail)
    char* ring_in[2] = { nullptr, nullptr };
    char* ring_out[2] = { nullptr, nullptr };
    char* bounce = nullptr;
    bool ok = false;
};
std::mutex g_pipe_mutex; // one streaming call at a time per process (it saturates PCIe anyway)
std::map<int, Pipe> g_pipes; // streams and device buffers belong to ONE device: keyed by the plan's device ordinal

size_t page_bytes()
{
    static const size_t v = [] {
        const long p = sysconf (_SC_PAGESIZE);
        return p > 0 ? (size_t) p : (size_t) 4096;
    }();
    return v;
}

// caller holds g_pipe_mutex and has the plan's device current
bool grow_pipe (Pipe& p, size_t floats)
{
    if (! p.ok)
    {
        for (int k = 0; k < 2; ++k)
            if (hipStreamCreateWithFlags (&p.st[k], hipStreamNonBlocking) != hipSuccess)
                return false;
        for (int k = 0; k < 2; ++k)
            if (hipHostMalloc ((void**) &p.ring_in[k], kRingSlotBytes, hipHostMallocDefault) != hipSuccess
                || hipHostMalloc ((void**) &p.ring_out[k], kRingSlotBytes, hipHostMallocDefault) != hipSuccess)
                return false;
        if (hipHostMalloc ((void**) &p.bounce, 4 * page_bytes(), hipHostMallocDefault) != hipSuccess)
            return false;
        p.ok = true;
    }
    if (p.floats >= floats)
        return true;
    for (int k = 0; k < 2; ++k)
    {
        if (p.din[k] != nullptr)
            (void) hipFree (p.din[k]);
        if (p.dout[k] != nullptr)
            (void) hipFree (p.dout[k]);
        p.din[k] = p.dout[k] = nullptr;
    }
    p.floats = 0;
    for (int k = 0; k < 2; ++k)
        if (hipMalloc ((void**) &p.din[k], floats * sizeof (float)) != hipSuccess || hipMalloc ((void**) &p.dout[k], floats * sizeof (float)) != hipSuccess)
            return false;
    p.floats = floats;
    return true;
}

Pipe* ensure_pipe (int device, size_t floats)
{
    Pipe& p = g_pipes[device];
    return grow_pipe (p, floats) ? &p : nullptr;
}

// From this many bytes on a caller buffer is pinned in place (its whole pages only); below, its chunks go through the staging
// ring.  A PERFORMANCE threshold only (registering a range costs page-table work proportional to its size plus a fixed part;
// measured break-even a few MiB) -- safety does not depend on it.  CHOWDSP_FFT_HOST_PIN_MIN_BYTES overrides;
// CHOWDSP_FFT_HOST_PIN_LEGACY=1 brings back round 2's registration of the WHOLE range including its partial pages
// (tools/repro_pin.py only: that is the form that faulted).
size_t pin_min_bytes()
{
    static const size_t v = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_HOST_PIN_MIN_BYTES");
        return e != nullptr ? (size_t) std::strtoull (e, nullptr, 10) : (size_t) 8 << 20;
    }();
    return v;
}
bool pin_legacy()
{
    static const bool v = std::getenv ("CHOWDSP_FFT_HOST_PIN_LEGACY") != nullptr && std::atoi (std::getenv ("CHOWDSP_FFT_HOST_PIN_LEGACY")) != 0;
    return v;
}

// One caller buffer as the copies see it: [base, base + bytes) with the whole pages [lo, hi) inside it registered (or not:
// lo == hi), everything outside [lo, hi) going through a bounce page.
struct HostRange
{
    char* base = nullptr;
    size_t bytes = 0;
    char* lo = nullptr; // registered interior, lo == hi: nothing registered
    char* hi = nullptr;
    bool registered_by_us = false;
    char* bounce_head = nullptr; // one pinned page each
    char* bounce_tail = nullptr;
};

HostRange make_range (const void* p, size_t bytes, char* bounce_head, char* bounce_tail)
{
    HostRange r;
    r.base = (char*) const_cast<void*> (p);
    r.bytes = bytes;
    r.bounce_head = bounce_head;
    r.bounce_tail = bounce_tail;
    const uintptr_t pg = (uintptr_t) page_bytes();
    const uintptr_t a = ((uintptr_t) r.base + pg - 1) / pg * pg;
    const uintptr_t b = ((uintptr_t) r.base + bytes) / pg * pg;
    r.lo = r.hi = r.base;
    if (pin_legacy())
    {
        if (hipHostRegister (r.base, bytes, hipHostRegisterDefault) == hipSuccess)
        {
            r.lo = r.base;
            r.hi = r.base + bytes;
            r.registered_by_us = true;
        }
        else
            (void) hipGetLastError();
        return r;
    }
    if (b > a && (size_t) (b - a) >= pin_min_bytes())
    {
        const hipError_t e = hipHostRegister ((void*) a, (size_t) (b - a), hipHostRegisterDefault);
        if (e == hipSuccess)
        {
            r.lo = (char*) a;
            r.hi = (char*) b;
            r.registered_by_us = true;
        }
        else
        {
            (void) hipGetLastError();
            // already pinned by the caller (fft_gpu_host_malloc, hipHostMalloc, a registration of theirs): DMA it as it is,
            // head and tail included -- the pages are theirs to pin.  Anything else that is not registrable: staging ring.
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes (&attr, r.base) == hipSuccess && attr.type == hipMemoryTypeHost)
            {
                r.lo = r.base;
                r.hi = r.base + bytes;
            }
            else
                (void) hipGetLastError();
        }
    }
    else
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes (&attr, r.base) == hipSuccess && attr.type == hipMemoryTypeHost)
        {
            r.lo = r.base; // small, but already pinned by its owner
            r.hi = r.base + bytes;
        }
        else
            (void) hipGetLastError();
    }
    return r;
}
void release_range (HostRange& r)
{
    if (r.registered_by_us)
        (void) hipHostUnregister (r.lo);
    r.registered_by_us = false;
}
bool direct (const HostRange& r) { return r.hi > r.lo; }

// host -> device copy of [src, src + bytes) of range r: the part inside the registered interior directly, a partial head /
// tail page through its bounce page.  Every byte of the head [base, lo) and of the tail [hi, base + bytes) has a place of its
// OWN in the bounce page (offset = its distance from base resp. hi): with frames shorter than a page the tail can be split
// between the last two chunks (N = 64 complex, batch = k * chunk_frames + 1, base = page + 16: chunk i - 1 owns 16 bytes of
// it, chunk i 512), which run on different streams -- they must neither share bounce bytes nor a pending record.
hipError_t copy_in (const HostRange& r, float* dst, const char* src, size_t bytes, hipStream_t st)
{
    const char* end = src + bytes;
    char* d = (char*) dst;
    if (src < r.lo)
    {
        const size_t n = (size_t) (std::min<const char*> (end, r.lo) - src);
        char* bh = r.bounce_head + (src - r.base);
        std::memcpy (bh, src, n);
        const hipError_t e = hipMemcpyAsync (d, bh, n, hipMemcpyHostToDevice, st);
        if (e != hipSuccess)
            return e;
    }
    const char* a = std::max<const char*> (src, r.lo);
    const char* b = std::min<const char*> (end, r.hi);
    if (b > a)
    {
        const hipError_t e = hipMemcpyAsync (d + (a - src), a, (size_t) (b - a), hipMemcpyHostToDevice, st);
        if (e != hipSuccess)
            return e;
    }
    if (end > r.hi)
    {
        const char* t = std::max<const char*> (src, r.hi);
        const size_t n = (size_t) (end - t);
        char* bt = r.bounce_tail + (t - r.hi);
        std::memcpy (bt, t, n);
        const hipError_t e = hipMemcpyAsync (d + (t - src), bt, n, hipMemcpyHostToDevice, st);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

// what copy_out left in the bounce pages: moved to the caller's buffer after the streams have been synchronised
struct Pending
{
    char* dst = nullptr;
    const char* src = nullptr;
    size_t bytes = 0;
};
hipError_t copy_out (const HostRange& r, char* dst, const float* src, size_t bytes, hipStream_t st, Pending (&pend)[2])
{
    char* end = dst + bytes;
    const char* s = (const char*) src;
    if (dst < r.lo)
    {
        const size_t n = (size_t) (std::min<char*> (end, r.lo) - dst);
        const hipError_t e = hipMemcpyAsync (r.bounce_head + (dst - r.base), s, n, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess)
            return e;
        pend[0] = Pending { r.base, r.bounce_head, (size_t) (r.lo - r.base) }; // the whole head, once (every byte of it gets written by some chunk)
    }
    char* a = std::max<char*> (dst, r.lo);
    char* b = std::min<char*> (end, r.hi);
    if (b > a)
    {
        const hipError_t e = hipMemcpyAsync (a, s + (a - dst), (size_t) (b - a), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess)
            return e;
    }
    if (end > r.hi)
    {
        char* t = std::max<char*> (dst, r.hi);
        const size_t n = (size_t) (end - t);
        const hipError_t e = hipMemcpyAsync (r.bounce_tail + (t - r.hi), s + (t - dst), n, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess)
            return e;
        pend[1] = Pending { r.hi, r.bounce_tail, (size_t) (r.base + r.bytes - r.hi) }; // the whole tail, once
    }
    return hipSuccess;
}
} // namespace

#pragma GCC visibility push(default)
extern "C"
{
namespace chowdsp::fft
{
int fft_transform_host_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr)
        return FFT_GPU_EINVAL;
    if (batch == 0)
        return FFT_GPU_OK;
    const size_t frame = (s->transform == FFT_COMPLEX) ? 2 * (size_t) s->N : (size_t) s->N;
    const size_t frame_bytes = frame * sizeof (float);
    const size_t total_bytes = batch * frame_bytes;
    // in place (input == output) is fine; any other overlap would let one chunk's result overwrite another chunk's input
    if (input != output)
    {
        const char *ib = (const char*) input, *ob = (const char*) output;
        if (ib < ob + total_bytes && ob < ib + total_bytes)
        {
            set_last_error ("fft_transform_host_batch: input and output overlap without being the same buffer");
            return FFT_GPU_EINVAL;
        }
    }
    const DeviceGuard guard (s->device); // the pipe's streams / buffers and the copies below belong to the plan's device
    std::lock_guard<std::mutex> lock (g_pipe_mutex);
    Pipe* pipe = ensure_pipe (s->device, 1);
    if (pipe == nullptr)
    {
        set_last_error ("fft_transform_host_batch: stream / pinned staging creation failed");
        return FFT_GPU_ENOMEM;
    }
    const size_t pg = page_bytes();
    HostRange rin = make_range (input, total_bytes, pipe->bounce, pipe->bounce + pg);
    HostRange rout = (output == input) ? rin : make_range (output, total_bytes, pipe->bounce + 2 * pg, pipe->bounce + 3 * pg);
    if (output == input)
    {
        rout.registered_by_us = false; // one registration, released through rin
        rout.bounce_head = pipe->bounce + 2 * pg;
        rout.bounce_tail = pipe->bounce + 3 * pg;
    }
    // a buffer that is neither registered now nor pinned by its owner goes through the staging ring: chunks of one ring slot
    const bool ring_in = ! direct (rin), ring_out = ! direct (rout);
    size_t chunk_bytes = kChunkBytes;
    if (ring_in || ring_out)
        chunk_bytes = kRingSlotBytes;
    const size_t chunk_frames = std::max<size_t> (1, std::min (batch, chunk_bytes / frame_bytes));
    const bool frame_fits_ring = frame_bytes <= kRingSlotBytes;
    int rc = FFT_GPU_OK;
    if (! grow_pipe (*pipe, chunk_frames * frame))
    {
        set_last_error ("fft_transform_host_batch: device buffer creation failed");
        rc = FFT_GPU_ENOMEM;
    }
    Pipe& g_pipe = *pipe;
    Pending pend[2];
    // ring slots in flight: the output of the chunk that used slot k last still has to be copied out of ring_out[k]
    struct InFlight
    {
        char* dst = nullptr;
        size_t bytes = 0;
    } fly[2];
    const auto drain_slot = [&] (int k) -> bool {
        if (fly[k].bytes == 0)
            return true;
        if (sync_stream_checked (g_pipe.st[k], "fft_transform_host_batch: stream synchronisation") != FFT_GPU_OK)
            return false;
        std::memcpy (fly[k].dst, g_pipe.ring_out[k], fly[k].bytes);
        fly[k].bytes = 0;
        return true;
    };
    size_t i = 0;
    for (size_t f0 = 0; f0 < batch && rc == FFT_GPU_OK; f0 += chunk_frames, ++i)
    {
        const int k = (int) (i & 1);
        const size_t nb = std::min (chunk_frames, batch - f0);
        const size_t bytes = nb * frame_bytes;
        const char* src = (const char*) input + f0 * frame_bytes;
        char* dst = (char*) output + f0 * frame_bytes;
        hipError_t e = hipSuccess;
        // stream order on st[k] protects din[k] / dout[k] from the chunk two steps back; the ring slots are host memory that
        // the CPU writes, so slot k is drained (its stream waited for, its output handed to the caller) before it is refilled
        if ((ring_in || ring_out) && ! drain_slot (k))
        {
            rc = FFT_GPU_ELAUNCH; // (sync_stream_checked recorded the reason)
            break;
        }
        if (ring_in && frame_fits_ring)
        {
            std::memcpy (g_pipe.ring_in[k], src, bytes);
            e = hipMemcpyAsync (g_pipe.din[k], g_pipe.ring_in[k], bytes, hipMemcpyHostToDevice, g_pipe.st[k]);
        }
        else if (ring_in) // a single frame larger than a ring slot in an unpinnable buffer: the runtime's own staging
            e = hipMemcpyAsync (g_pipe.din[k], src, bytes, hipMemcpyHostToDevice, g_pipe.st[k]);
        else
            e = copy_in (rin, g_pipe.din[k], src, bytes, g_pipe.st[k]);
        if (e != hipSuccess)
        {
            set_last_error ("fft_transform_host_batch: host->device copy failed", e);
            rc = FFT_GPU_ELAUNCH;
            break;
        }
        rc = fft_transform_batch (setup, g_pipe.din[k], g_pipe.dout[k], nb, direction, g_pipe.st[k]);
        if (rc != FFT_GPU_OK)
            break;
        if (ring_out && frame_fits_ring)
        {
            e = hipMemcpyAsync (g_pipe.ring_out[k], g_pipe.dout[k], bytes, hipMemcpyDeviceToHost, g_pipe.st[k]);
            fly[k].dst = dst;
            fly[k].bytes = bytes;
        }
        else if (ring_out)
            e = hipMemcpyAsync (dst, g_pipe.dout[k], bytes, hipMemcpyDeviceToHost, g_pipe.st[k]);
        else
            e = copy_out (rout, dst, g_pipe.dout[k], bytes, g_pipe.st[k], pend);
        if (e != hipSuccess)
        {
            set_last_error ("fft_transform_host_batch: device->host copy failed", e);
            rc = FFT_GPU_ELAUNCH;
        }
    }
    for (int k = 0; k < 2; ++k)
    {
        // (always both streams, also after a failure: nothing of this call may still be in flight when it returns)
        const int src = sync_stream_checked (g_pipe.st[k], "fft_transform_host_batch: stream synchronisation");
        if (src != FFT_GPU_OK && rc == FFT_GPU_OK)
            rc = src;
        if (fly[k].bytes != 0 && rc == FFT_GPU_OK)
            std::memcpy (fly[k].dst, g_pipe.ring_out[k], fly[k].bytes);
        fly[k].bytes = 0;
    }
    if (rc == FFT_GPU_OK)
        for (const Pending& p : pend)
            if (p.bytes != 0)
                std::memcpy (p.dst, p.src, p.bytes);
    release_range (rin);
    release_range (rout);
    return rc;
}
} // namespace chowdsp::fft
} // extern "C"
#pragma GCC visibility pop
