// Host-memory streaming mode: many frames in host memory through the GPU with pinned
// buffers and two alternating streams, instead of one staged copy + launch + sync per frame (the legacy symbols).
#include "../../include/chowdsp_fft_gpu.h"

#include "host_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

using namespace cfft::host;

namespace
{
constexpr size_t kChunkBytes = 32ull << 20; // per direction and stream: large enough for PCIe efficiency, small enough to pipeline

struct Pipe
{
    hipStream_t st[2] = { nullptr, nullptr };
    float* din[2] = { nullptr, nullptr };
    float* dout[2] = { nullptr, nullptr };
    size_t floats = 0; // capacity of each device buffer
    bool ok = false;
};
std::mutex g_pipe_mutex; // one streaming call at a time per process (it saturates PCIe anyway)
std::map<int, Pipe> g_pipes; // streams and device buffers belong to ONE device: keyed by the plan's device ordinal

// caller holds g_pipe_mutex and has the plan's device current
bool grow_pipe (Pipe& p, size_t floats)
{
    if (! p.ok)
    {
        for (int k = 0; k < 2; ++k)
            if (hipStreamCreateWithFlags (&p.st[k], hipStreamNonBlocking) != hipSuccess)
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

// Buffers this large are registered (pinned) in place for the duration of the call; smaller ones are handed to hipMemcpyAsync
// as they are (the runtime stages them).  Two reasons: below a few tens of MiB registering costs more than it saves, and a
// block larger than glibc's largest mmap threshold (32 MiB) is always a mapping of its own, whereas a smaller block may lie
// inside the process heap and share its first and last page with unrelated objects -- registering and unregistering pages
// that other code (the HIP runtime's own pageable-copy path among it) may be using at the same time is not something a
// library should do to its host.  (Session r02j: a test run aborted twice with "Memory access fault ... on address <process
// heap>" inside a later, unrelated pageable copy, each time right after the host-streaming tests had registered small heap
// blocks; CHOWDSP_FFT_HOST_PIN_MIN_BYTES overrides.)
size_t pin_min_bytes()
{
    static const size_t v = [] {
        const char* e = std::getenv ("CHOWDSP_FFT_HOST_PIN_MIN_BYTES");
        return e != nullptr ? (size_t) std::strtoull (e, nullptr, 10) : ((size_t) 32 << 20) + 4096;
    }();
    return v;
}

// pin a caller buffer in place; returns true if WE registered it (and must unregister)
bool pin (const void* p, size_t bytes)
{
    if (bytes < pin_min_bytes())
        return false;
    const hipError_t e = hipHostRegister (const_cast<void*> (p), bytes, hipHostRegisterDefault);
    if (e == hipSuccess)
        return true;
    (void) hipGetLastError(); // already registered / pinned by the caller (fft_gpu_host_malloc) or not pinnable: copies still work
    return false;
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
    const size_t chunk_frames = std::max<size_t> (1, std::min (batch, kChunkBytes / (frame * sizeof (float))));
    const size_t total_bytes = batch * frame * sizeof (float);
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
    Pipe* pipe = ensure_pipe (s->device, chunk_frames * frame);
    if (pipe == nullptr)
    {
        set_last_error ("fft_transform_host_batch: stream / device buffer creation failed");
        return FFT_GPU_ENOMEM;
    }
    Pipe& g_pipe = *pipe;
    const bool pinned_in = pin (input, total_bytes);
    const bool pinned_out = (output != input) ? pin (output, total_bytes) : false;
    int rc = FFT_GPU_OK;
    size_t i = 0;
    for (size_t f0 = 0; f0 < batch && rc == FFT_GPU_OK; f0 += chunk_frames, ++i)
    {
        const int k = (int) (i & 1);
        const size_t nb = std::min (chunk_frames, batch - f0);
        const size_t bytes = nb * frame * sizeof (float);
        // stream order on st[k] protects din[k]/dout[k] from the chunk two steps back
        if (hipMemcpyAsync (g_pipe.din[k], input + f0 * frame, bytes, hipMemcpyHostToDevice, g_pipe.st[k]) != hipSuccess)
        {
            set_last_error ("fft_transform_host_batch: host->device copy failed");
            rc = FFT_GPU_ELAUNCH;
            break;
        }
        rc = fft_transform_batch (setup, g_pipe.din[k], g_pipe.dout[k], nb, direction, g_pipe.st[k]);
        if (rc != FFT_GPU_OK)
            break;
        if (hipMemcpyAsync (output + f0 * frame, g_pipe.dout[k], bytes, hipMemcpyDeviceToHost, g_pipe.st[k]) != hipSuccess)
        {
            set_last_error ("fft_transform_host_batch: device->host copy failed");
            rc = FFT_GPU_ELAUNCH;
        }
    }
    for (int k = 0; k < 2; ++k)
        if (hipStreamSynchronize (g_pipe.st[k]) != hipSuccess && rc == FFT_GPU_OK)
        {
            set_last_error ("fft_transform_host_batch: stream synchronisation failed");
            rc = FFT_GPU_ELAUNCH;
        }
    if (pinned_in)
        (void) hipHostUnregister (const_cast<float*> (input));
    if (pinned_out)
        (void) hipHostUnregister (output);
    return rc;
}
} // namespace chowdsp::fft
} // extern "C"
#pragma GCC visibility pop
