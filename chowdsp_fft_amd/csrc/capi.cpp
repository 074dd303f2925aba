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
#include "tables.hpp"

#include <cstdlib>
#include <new>

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

int run_transform (Setup* s, const float* in, float* out, size_t batch, cf::fft_direction_t dir, hipStream_t stream)
{
    const bool inverse = (dir == cf::FFT_BACKWARD);
    const DeviceTables* t = s->tables;
    hipError_t e = hipErrorInvalidValue;
    if (s->transform == cf::FFT_COMPLEX && t->core_n == s->N)
        e = launch_c2c (s->N, inverse, (const float2*) in, (float2*) out, t->stage_twiddles, batch, stream);
    else
    {
        set_last_error ("transform kind/size not available in this build");
        return cf::FFT_GPU_ENOTSUP;
    }
    if (e != hipSuccess)
    {
        set_last_error ("kernel launch", e);
        return cf::FFT_GPU_ELAUNCH;
    }
    return cf::FFT_GPU_OK;
}

// legacy single-frame entry: host pointers are staged, device pointers used in place; synchronous.
void legacy_transform (void* setup, const float* input, float* output, cf::fft_direction_t direction)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr)
        return;
    const size_t nf = frame_floats (s);
    const bool in_dev = is_device_pointer (input), out_dev = is_device_pointer (output);
    if (in_dev && out_dev)
    {
        if (run_transform (s, input, output, 1, direction, nullptr) == cf::FFT_GPU_OK)
            (void) hipStreamSynchronize (nullptr);
        return;
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
        if (hipMemcpyAsync (stage, input, nf * sizeof (float), hipMemcpyHostToDevice, nullptr) != hipSuccess)
            return;
        din = stage;
    }
    if (! out_dev)
        dout = stage + nf;
    if (run_transform (s, din, dout, 1, direction, nullptr) != cf::FFT_GPU_OK)
        return;
    if (! out_dev)
        (void) hipMemcpyAsync (output, dout, nf * sizeof (float), hipMemcpyDeviceToHost, nullptr);
    (void) hipStreamSynchronize (nullptr);
}
} // namespace

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
    void* block = s->owned_block;
    s->magic = 0;
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

void fft_convolve_unordered (void*, const float*, const float*, float*, float)
{
    set_last_error ("fft_convolve_unordered: not available in this build");
}

void fft_accumulate (void*, const float*, const float*, float*, int)
{
    set_last_error ("fft_accumulate: not available in this build");
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
    CFFT_HIP_TRY (hipStreamSynchronize ((hipStream_t) stream), FFT_GPU_ELAUNCH);
    return FFT_GPU_OK;
}

const char* fft_gpu_last_error (void) { return last_error(); }

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

size_t fft_gpu_workspace_bytes (void*, size_t) { return 0; }

// ---- extension: batched operations ---------------------------------------------------------------
int fft_transform_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream)
{
    Setup* s = as_setup (setup);
    if (s == nullptr || input == nullptr || output == nullptr)
        return FFT_GPU_EINVAL;
    return run_transform (s, input, output, batch, direction, (hipStream_t) stream);
}

int fft_transform_unordered_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream)
{
    return fft_transform_batch (setup, input, output, batch, direction, stream);
}

int fft_convolve_unordered_batch (void*, const float*, const float*, float*, float, size_t, int, void*)
{
    return FFT_GPU_ENOTSUP;
}

int fft_accumulate_batch (void*, const float*, const float*, float*, size_t, void*) { return FFT_GPU_ENOTSUP; }

int fft_convolve_frames_batch (void*, const float*, const float*, float*, float, size_t, void*) { return FFT_GPU_ENOTSUP; }
} // namespace chowdsp::fft
} // extern "C"
