// Several devices inside ONE process, at the C ABI (SURVEY section 8 row e): contiguous block sharding of a batch of
// independent frames, replicated plans, no data-path collective; the only exchange is the optional collection of results,
// done as direct one-hop peer copies (every source streams its shard to every destination on its own stream, so that on
// the xGMI full mesh all links carry traffic at once; a ring would be bound by one link).
// The torch.distributed (RCCL) form of the same thing, one process per GPU, is chowdsp_fft_amd/sharding.py + bench.py.
#include "../../include/chowdsp_fft_gpu.h"

#include "host_common.hpp"

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace cfft::host;

namespace
{
std::mutex g_mutex;
std::map<int, hipStream_t> g_shard_stream;                  // device -> the stream its shard runs on
std::map<std::pair<int, int>, hipStream_t> g_copy_stream;   // (source device, destination device) -> copy stream on the source

hipStream_t shard_stream (int device) // caller has `device` current
{
    std::lock_guard<std::mutex> lock (g_mutex);
    auto it = g_shard_stream.find (device);
    if (it != g_shard_stream.end())
        return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags (&s, hipStreamNonBlocking) != hipSuccess)
        return nullptr;
    g_shard_stream.emplace (device, s);
    return s;
}

hipStream_t copy_stream (int src, int dst) // caller has `src` current
{
    std::lock_guard<std::mutex> lock (g_mutex);
    const auto key = std::make_pair (src, dst);
    auto it = g_copy_stream.find (key);
    if (it != g_copy_stream.end())
        return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags (&s, hipStreamNonBlocking) != hipSuccess)
        return nullptr;
    if (src != dst)
    {
        // direct loads / stores over xGMI where the platform allows it; the copy engine works either way
        int can = 0;
        if (hipDeviceCanAccessPeer (&can, src, dst) == hipSuccess && can != 0)
        {
            const hipError_t e = hipDeviceEnablePeerAccess (dst, 0);
            if (e != hipSuccess)
                (void) hipGetLastError(); // already enabled
        }
    }
    g_copy_stream.emplace (key, s);
    return s;
}
} // namespace

namespace
{
using namespace chowdsp::fft;
// One host thread per shard: a thread selects its plan's device, enqueues the shard (`enqueue (d, stream)`) on that device's
// own stream and waits for it; the shards run concurrently.  Errors are collected per shard (last_error is thread local).
template <class Enqueue>
int run_shards (void* const* setups, int shards, const size_t* counts, const char* what, Enqueue&& enqueue)
{
    std::vector<int> rc ((size_t) shards, FFT_GPU_OK);
    std::vector<std::string> msg ((size_t) shards);
    auto work = [&] (int d) {
        const Setup* s = as_setup (setups[d]);
        if (counts[d] == 0)
            return;
        if (hipSetDevice (s->device) != hipSuccess)
        {
            rc[(size_t) d] = FFT_GPU_ENODEV;
            msg[(size_t) d] = "hipSetDevice failed";
            return;
        }
        hipStream_t st = shard_stream (s->device);
        if (st == nullptr)
        {
            rc[(size_t) d] = FFT_GPU_ELAUNCH;
            msg[(size_t) d] = "stream creation failed";
            return;
        }
        int r = enqueue (d, st);
        if (r == FFT_GPU_OK)
            r = sync_stream_checked (st, what); // incl. the one-launch forms' hand-off report
        rc[(size_t) d] = r;
        if (r != FFT_GPU_OK)
            msg[(size_t) d] = last_error();
    };
    if (shards == 1)
    {
        const DeviceGuard guard (as_setup (setups[0])->device);
        work (0);
    }
    else
    {
        std::vector<std::thread> pool;
        for (int d = 0; d < shards; ++d)
            pool.emplace_back (work, d);
        for (auto& t : pool)
            t.join();
    }
    for (int d = 0; d < shards; ++d)
        if (rc[(size_t) d] != FFT_GPU_OK)
        {
            set_last_error (msg[(size_t) d].c_str());
            return rc[(size_t) d];
        }
    return FFT_GPU_OK;
}
} // namespace

#pragma GCC visibility push(default)
extern "C"
{
namespace chowdsp::fft
{
int fft_gpu_shard_range (size_t total_frames, int shards, int rank, size_t* first, size_t* count)
{
    if (shards < 1 || rank < 0 || rank >= shards || first == nullptr || count == nullptr)
        return FFT_GPU_EINVAL;
    // blocks differ by at most one frame and cover [0, total) exactly once (= chowdsp_fft_amd/sharding.py::shard_range)
    const size_t base = total_frames / (size_t) shards, extra = total_frames % (size_t) shards;
    const size_t r = (size_t) rank;
    *first = r * base + (r < extra ? r : extra);
    *count = base + (r < extra ? 1 : 0);
    return FFT_GPU_OK;
}

int fft_transform_batch_multi_gpu (void* const* setups, int shards, const float* const* inputs, float* const* outputs, const size_t* counts, fft_direction_t direction)
{
    if (setups == nullptr || inputs == nullptr || outputs == nullptr || counts == nullptr || shards < 1)
        return FFT_GPU_EINVAL;
    for (int d = 0; d < shards; ++d)
        if (as_setup (setups[d]) == nullptr || (counts[d] != 0 && (inputs[d] == nullptr || outputs[d] == nullptr)))
            return FFT_GPU_EINVAL;
    return run_shards (setups, shards, counts, "fft_transform_batch_multi_gpu: shard synchronisation",
                       [&] (int d, hipStream_t st) { return fft_transform_batch (setups[d], inputs[d], outputs[d], counts[d], direction, st); });
}

// BASELINE configs[3] at the C ABI: the fused forward -> x H x scaling -> backward of every shard on its own device, filter
// spectrum replicated per device (reference call sequence test/test.cpp:161-166; plan / filter shareable chowdsp_fft.h:87-91)
int fft_convolve_frames_batch_multi_gpu (void* const* setups, int shards, const float* const* x, const float* const* dft_filters, float* const* y, const size_t* counts, float scaling)
{
    if (setups == nullptr || x == nullptr || dft_filters == nullptr || y == nullptr || counts == nullptr || shards < 1)
        return FFT_GPU_EINVAL;
    for (int d = 0; d < shards; ++d)
        if (as_setup (setups[d]) == nullptr || (counts[d] != 0 && (x[d] == nullptr || y[d] == nullptr || dft_filters[d] == nullptr)))
            return FFT_GPU_EINVAL;
    return run_shards (setups, shards, counts, "fft_convolve_frames_batch_multi_gpu: shard synchronisation",
                       [&] (int d, hipStream_t st) { return fft_convolve_frames_batch (setups[d], x[d], dft_filters[d], y[d], scaling, counts[d], st); });
}

int fft_gather_multi_gpu (const float* const* shard_ptrs, const int* shard_devices, const size_t* shard_floats, int shards, float* const* dst_ptrs, const int* dst_devices, int ndst)
{
    if (shard_ptrs == nullptr || shard_devices == nullptr || shard_floats == nullptr || dst_ptrs == nullptr || dst_devices == nullptr || shards < 1 || ndst < 1)
        return FFT_GPU_EINVAL;
    int prev = -1;
    (void) hipGetDevice (&prev);
    int rc = FFT_GPU_OK;
    std::vector<hipStream_t> used;
    std::vector<int> used_dev;
    // The copies run on the library's own streams, which know nothing of the streams that produced the shards or last used
    // the destinations: wait for every device involved first (whatever the caller enqueued there -- fft_transform_batch on
    // a stream of theirs, a fill of the destination -- is complete before the first byte moves).
    for (int pass = 0; pass < 2; ++pass)
    {
        const int count = pass == 0 ? shards : ndst;
        for (int i = 0; i < count; ++i)
        {
            if (pass == 0 ? shard_floats[i] == 0 : dst_ptrs[i] == nullptr)
                continue;
            const int dev = pass == 0 ? shard_devices[i] : dst_devices[i];
            if (hipSetDevice (dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            {
                set_last_error ("fft_gather_multi_gpu: could not synchronise a source / destination device");
                if (prev >= 0)
                    (void) hipSetDevice (prev);
                return FFT_GPU_ENODEV;
            }
        }
    }
    // every (source, destination) pair gets its own stream on the source device: all copies are in flight together
    for (int s = 0; s < shards && rc == FFT_GPU_OK; ++s)
    {
        if (shard_floats[s] == 0)
            continue;
        if (hipSetDevice (shard_devices[s]) != hipSuccess)
        {
            set_last_error ("fft_gather_multi_gpu: hipSetDevice failed");
            rc = FFT_GPU_ENODEV;
            break;
        }
        size_t offset = 0;
        for (int k = 0; k < s; ++k)
            offset += shard_floats[k];
        for (int d = 0; d < ndst; ++d)
        {
            if (dst_ptrs[d] == nullptr)
                continue;
            hipStream_t st = copy_stream (shard_devices[s], dst_devices[d]);
            if (st == nullptr)
            {
                set_last_error ("fft_gather_multi_gpu: stream creation failed");
                rc = FFT_GPU_ELAUNCH;
                break;
            }
            const hipError_t e = hipMemcpyPeerAsync (dst_ptrs[d] + offset, dst_devices[d], shard_ptrs[s], shard_devices[s], shard_floats[s] * sizeof (float), st);
            if (e != hipSuccess)
            {
                set_last_error ("fft_gather_multi_gpu: hipMemcpyPeerAsync", e);
                rc = FFT_GPU_ELAUNCH;
                break;
            }
            used.push_back (st);
            used_dev.push_back (shard_devices[s]);
        }
    }
    for (size_t i = 0; i < used.size(); ++i)
    {
        (void) hipSetDevice (used_dev[i]);
        if (hipStreamSynchronize (used[i]) != hipSuccess && rc == FFT_GPU_OK)
        {
            set_last_error ("fft_gather_multi_gpu: copy synchronisation failed");
            rc = FFT_GPU_ELAUNCH;
        }
    }
    if (prev >= 0)
        (void) hipSetDevice (prev);
    return rc;
}
} // namespace chowdsp::fft
} // extern "C"
#pragma GCC visibility pop
