// Streaming uniformly-partitioned overlap-save convolution engine, host side.
// Blocks up to 4096 samples run as two fused kernels per block (conv_kernels.hpp: ols_r2c_kernel,
// ols_mac_c2r_kernel); longer blocks are built from the library's public batched operations plus three small
// kernels (frame assembly, delay-line multiply-accumulate, output extraction).  See include/chowdsp_fft_gpu.h.
#include "../../include/chowdsp_fft_gpu.h"

#include "host_common.hpp"
#include "launch.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace cfft::host;
namespace cf = chowdsp::fft;

namespace
{
constexpr uint32_t kEngineMagic = 0xC0EF1A55u;

struct Engine
{
    uint32_t magic = kEngineMagic;
    int block = 0, channels = 0, partitions = 0;
    bool shared = false;
    unsigned head = 0; // delay-line slot that receives the next spectrum
    void* setup = nullptr;
    float *prev = nullptr, *frame = nullptr, *fdl = nullptr, *H = nullptr, *Y = nullptr;
    // fused path (block <= 4096): two kernels per block, the previous input block double-buffered
    bool fused = false;
    float* prev2 = nullptr;
    void* state = nullptr; // device-resident position (OlsState): a captured process() call can be replayed block after block
};

Engine* as_engine (void* p)
{
    auto* e = reinterpret_cast<Engine*> (p);
    return (e != nullptr && e->magic == kEngineMagic) ? e : nullptr;
}

void release (Engine* e)
{
    if (e == nullptr)
        return;
    for (float* p : { e->prev, e->prev2, e->frame, e->fdl, e->H, e->Y })
        if (p != nullptr)
            (void) hipFree (p);
    if (e->state != nullptr)
        (void) hipFree (e->state);
    if (e->setup != nullptr)
        cf::fft_destroy_setup (e->setup);
    e->magic = 0;
    delete e;
}

bool dev_alloc (float*& p, size_t floats)
{
    return hipMalloc ((void**) &p, floats * sizeof (float)) == hipSuccess;
}
} // namespace

#pragma GCC visibility push(default)
extern "C"
{
namespace chowdsp::fft
{
void* fft_conv_engine_create (int block, int channels, const float* ir, int ir_len, int ir_is_shared, void* stream_)
{
    const hipStream_t stream = (hipStream_t) stream_;
    if (block < 16 || (block & (block - 1)) != 0 || channels < 1 || ir == nullptr || ir_len < 1)
    {
        set_last_error ("fft_conv_engine_create: bad arguments");
        return nullptr;
    }
    const int N = 2 * block;
    auto* e = new (std::nothrow) Engine();
    if (e == nullptr)
        return nullptr;
    e->block = block;
    e->channels = channels;
    e->partitions = (ir_len + block - 1) / block;
    e->shared = ir_is_shared != 0;
    e->setup = fft_new_setup (N, FFT_REAL, true);
    const size_t C = (size_t) channels, P = (size_t) e->partitions, Hc = e->shared ? 1 : C;
    // CHOWDSP_FFT_ENGINE_FUSED=0 forces the five-kernel path (A/B measurements, tests of both paths)
    const char* fenv = std::getenv ("CHOWDSP_FFT_ENGINE_FUSED");
    e->fused = ols_fused_supported (block) && ! (fenv != nullptr && fenv[0] == '0');
    bool ok_alloc = e->setup != nullptr && dev_alloc (e->prev, C * block) && dev_alloc (e->fdl, P * C * N) && dev_alloc (e->H, P * Hc * N);
    if (e->fused)
        ok_alloc = ok_alloc && dev_alloc (e->prev2, C * block) && hipMalloc (&e->state, ols_state_bytes()) == hipSuccess;
    else
        ok_alloc = ok_alloc && dev_alloc (e->frame, C * N) && dev_alloc (e->Y, C * N);
    if (! ok_alloc)
    {
        set_last_error ("fft_conv_engine_create: setup or device allocation failed");
        release (e);
        return nullptr;
    }
    // filter partitions: [p][c][ taps p*block .. (p+1)*block | block zeros ], transformed in place (unordered layout)
    hipPointerAttribute_t attr;
    const bool ir_on_device = hipPointerGetAttributes (&attr, ir) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    if (! ir_on_device)
        (void) hipGetLastError();
    bool ok = hipMemsetAsync (e->H, 0, P * Hc * N * sizeof (float), stream) == hipSuccess;
    for (size_t p = 0; p < P && ok; ++p)
    {
        const size_t first = p * (size_t) block;
        const size_t taps = std::min<size_t> ((size_t) block, (size_t) ir_len - first);
        ok = hipMemcpy2DAsync (e->H + p * Hc * N, (size_t) N * sizeof (float), ir + first, (size_t) ir_len * sizeof (float), taps * sizeof (float), Hc,
                               ir_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream)
             == hipSuccess;
    }
    ok = ok && fft_transform_unordered_batch (e->setup, e->H, e->H, P * Hc, FFT_FORWARD, stream) == FFT_GPU_OK;
    ok = ok && fft_conv_engine_reset (e, stream) == FFT_GPU_OK;
    if (! ir_on_device)
        ok = ok && hipStreamSynchronize (stream) == hipSuccess; // the caller may free a host ir right after we return
    if (! ok)
    {
        set_last_error ("fft_conv_engine_create: filter preparation failed");
        release (e);
        return nullptr;
    }
    return e;
}

int fft_conv_engine_reset (void* engine, void* stream_)
{
    Engine* e = as_engine (engine);
    if (e == nullptr || as_setup (e->setup) == nullptr)
        return FFT_GPU_EINVAL;
    const DeviceGuard guard (as_setup (e->setup)->device);
    const hipStream_t stream = (hipStream_t) stream_;
    const size_t C = (size_t) e->channels, N = 2 * (size_t) e->block;
    CFFT_HIP_TRY (hipMemsetAsync (e->prev, 0, C * e->block * sizeof (float), stream), FFT_GPU_ELAUNCH);
    if (e->prev2 != nullptr)
        CFFT_HIP_TRY (hipMemsetAsync (e->prev2, 0, C * e->block * sizeof (float), stream), FFT_GPU_ELAUNCH);
    if (e->state != nullptr)
        CFFT_HIP_TRY (hipMemsetAsync (e->state, 0, ols_state_bytes(), stream), FFT_GPU_ELAUNCH);
    CFFT_HIP_TRY (hipMemsetAsync (e->fdl, 0, (size_t) e->partitions * C * N * sizeof (float), stream), FFT_GPU_ELAUNCH);
    e->head = 0;
    return FFT_GPU_OK;
}

int fft_conv_engine_process (void* engine, const float* in, float* out, void* stream_)
{
    Engine* e = as_engine (engine);
    if (e == nullptr || in == nullptr || out == nullptr || as_setup (e->setup) == nullptr)
        return FFT_GPU_EINVAL;
    const DeviceGuard guard (as_setup (e->setup)->device); // whole call: the unfused path launches helper kernels itself
    const hipStream_t stream = (hipStream_t) stream_;
    const size_t C = (size_t) e->channels, N = 2 * (size_t) e->block;
    if (e->fused)
    {
        const Setup* su = as_setup (e->setup);
        const DeviceTables* t = su->tables;
        // CHOWDSP_FFT_ENGINE_SINGLE=0: two kernels per block (ols_r2c_kernel, ols_mac_c2r_kernel) instead of one
        static const bool single = [] {
            const char* v = std::getenv ("CHOWDSP_FFT_ENGINE_SINGLE");
            return ! (v != nullptr && v[0] == '0');
        }();
        hipError_t hf = hipSuccess;
        if (single)
            hf = launch_ols_block (e->block, e->prev, e->prev2, in, e->fdl, e->H, out, t->stage_twiddles, t->real_twiddles, 1.0f / (float) N,
                                   (unsigned) C, (unsigned) e->partitions, e->state, e->shared, stream);
        else
        {
            hf = launch_ols_r2c (e->block, e->prev, e->prev2, in, e->fdl, t->stage_twiddles, t->real_twiddles, (unsigned) C, e->state, stream);
            if (hf == hipSuccess)
                hf = launch_ols_mac_c2r (e->block, e->fdl, e->H, out, t->stage_twiddles, t->real_twiddles, 1.0f / (float) N, (unsigned) C,
                                         (unsigned) e->partitions, e->state, e->shared, stream);
        }
        if (hf != hipSuccess)
        {
            set_last_error ("engine: fused block kernels", hf);
            return FFT_GPU_ELAUNCH;
        }
        return FFT_GPU_OK;
    }
    hipError_t he = launch_ols_assemble (in, e->prev, e->frame, (size_t) e->block, (unsigned) C, stream);
    if (he != hipSuccess)
    {
        set_last_error ("engine: assemble", he);
        return FFT_GPU_ELAUNCH;
    }
    float* slot = e->fdl + (size_t) e->head * C * N;
    int rc = fft_transform_unordered_batch (e->setup, e->frame, slot, C, FFT_FORWARD, stream);
    if (rc != FFT_GPU_OK)
        return rc;
    he = launch_fdl_mac (e->fdl, e->H, e->Y, 1.0f / (float) N, N, (unsigned) C, (unsigned) e->partitions, e->head, e->shared, stream);
    if (he != hipSuccess)
    {
        set_last_error ("engine: delay-line multiply-accumulate", he);
        return FFT_GPU_ELAUNCH;
    }
    rc = fft_transform_unordered_batch (e->setup, e->Y, e->frame, C, FFT_BACKWARD, stream);
    if (rc != FFT_GPU_OK)
        return rc;
    he = launch_ols_extract (e->frame, out, (size_t) e->block, (unsigned) C, stream);
    if (he != hipSuccess)
    {
        set_last_error ("engine: extract", he);
        return FFT_GPU_ELAUNCH;
    }
    e->head = (e->head + 1) % (unsigned) e->partitions;
    return FFT_GPU_OK;
}

int fft_conv_engine_partitions (void* engine)
{
    Engine* e = as_engine (engine);
    return e != nullptr ? e->partitions : FFT_GPU_EINVAL;
}

void fft_conv_engine_destroy (void* engine) { release (as_engine (engine)); }
} // namespace chowdsp::fft
} // extern "C"
#pragma GCC visibility pop
