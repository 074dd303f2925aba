// Calibration micro-benchmarks for the HBM roofline on MI355X: plain copies with 8- and 16-byte
// accesses and the exact global access pattern of the N=4096 FFT kernel with the arithmetic removed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
    do                                                                           \
    {                                                                            \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess)                                                    \
        {                                                                        \
            printf ("HIP error %s at %s:%d\n", hipGetErrorString (e_), __FILE__, __LINE__); \
            exit (1);                                                            \
        }                                                                        \
    } while (0)

template <typename T>
__global__ void copy_oneshot (const T* __restrict__ in, T* __restrict__ out, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = in[i];
}

template <typename T, int U>
__global__ void copy_stride (const T* __restrict__ in, T* __restrict__ out, size_t n)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride)
    {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
            out[i + u * stride] = v[u];
    }
    for (; i < n; i += stride)
        out[i] = in[i];
}

// FFT-shaped: one 256-thread workgroup per 4096-point frame (32 KiB); thread t loads x[t + 256 a],
// a = 0..15 (16 x 8 B, 2 KiB apart), then stores the same elements.  PERSIST: grid-stride over frames.
template <bool PERSIST>
__global__ __launch_bounds__ (256) void frame_copy_f2 (const float2* __restrict__ in, float2* __restrict__ out, unsigned frames)
{
    const unsigned t = threadIdx.x;
    for (unsigned f = blockIdx.x; f < frames; f += gridDim.x)
    {
        const float2* src = in + (size_t) f * 4096;
        float2* dst = out + (size_t) f * 4096;
        float2 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a)
            v[a] = src[t + 256 * a];
#pragma unroll
        for (int a = 0; a < 16; ++a)
            dst[t + 256 * a] = v[a];
        if (! PERSIST)
            break;
    }
}

// same bytes, 16-byte accesses: thread t loads float4 #(t + 256 a), a = 0..7
template <bool PERSIST>
__global__ __launch_bounds__ (256) void frame_copy_f4 (const float4* __restrict__ in, float4* __restrict__ out, unsigned frames)
{
    const unsigned t = threadIdx.x;
    for (unsigned f = blockIdx.x; f < frames; f += gridDim.x)
    {
        const float4* src = in + (size_t) f * 2048;
        float4* dst = out + (size_t) f * 2048;
        float4 v[8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
            v[a] = src[t + 256 * a];
#pragma unroll
        for (int a = 0; a < 8; ++a)
            dst[t + 256 * a] = v[a];
        if (! PERSIST)
            break;
    }
}

template <class F>
float time_ms (F&& launch, int iters)
{
    hipEvent_t a, b;
    CK (hipEventCreate (&a));
    CK (hipEventCreate (&b));
    launch();
    launch();
    CK (hipDeviceSynchronize());
    CK (hipEventRecord (a));
    for (int i = 0; i < iters; ++i)
        launch();
    CK (hipEventRecord (b));
    CK (hipEventSynchronize (b));
    float ms = 0;
    CK (hipEventElapsedTime (&ms, a, b));
    CK (hipEventDestroy (a));
    CK (hipEventDestroy (b));
    return ms / iters;
}

int main (int argc, char** argv)
{
    const size_t frames = argc > 1 ? (size_t) atol (argv[1]) : 65536;
    const size_t bytes = frames * 32768;
    hipDeviceProp_t prop;
    CK (hipGetDeviceProperties (&prop, 0));
    printf ("device: %s  CUs=%d  clock=%d MHz  memclk=%d MHz  bus=%d bit  L2=%d KiB\n", prop.name, prop.multiProcessorCount,
            prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth, prop.l2CacheSize / 1024);
    void *in, *out;
    CK (hipMalloc (&in, bytes));
    CK (hipMalloc (&out, bytes));
    CK (hipMemset (in, 1, bytes));
    CK (hipMemset (out, 0, bytes));
    const int iters = 10;
    auto report = [&] (const char* name, float ms) {
        printf ("%-44s %8.3f ms  %8.1f GB/s (read+write)\n", name, ms, 2.0 * bytes / ms * 1e-6);
        fflush (stdout);
    };
    const int cus = prop.multiProcessorCount;
    {
        const size_t n = bytes / 16;
        report ("copy float4 one-shot", time_ms ([&] { copy_oneshot<float4><<<(unsigned) ((n + 255) / 256), 256>>> ((const float4*) in, (float4*) out, n); }, iters));
        for (int bpc : { 4, 8, 16 })
        {
            char nm[96];
            snprintf (nm, sizeof nm, "copy float4 grid-stride U=4 (%d blk/CU)", bpc);
            report (nm, time_ms ([&] { copy_stride<float4, 4><<<cus * bpc, 256>>> ((const float4*) in, (float4*) out, n); }, iters));
        }
        report ("copy float4 grid-stride U=8 (8 blk/CU)", time_ms ([&] { copy_stride<float4, 8><<<cus * 8, 256>>> ((const float4*) in, (float4*) out, n); }, iters));
    }
    {
        const size_t n = bytes / 8;
        report ("copy float2 one-shot", time_ms ([&] { copy_oneshot<float2><<<(unsigned) ((n + 255) / 256), 256>>> ((const float2*) in, (float2*) out, n); }, iters));
        report ("copy float2 grid-stride U=8 (8 blk/CU)", time_ms ([&] { copy_stride<float2, 8><<<cus * 8, 256>>> ((const float2*) in, (float2*) out, n); }, iters));
        report ("copy float2 grid-stride U=16 (8 blk/CU)", time_ms ([&] { copy_stride<float2, 16><<<cus * 8, 256>>> ((const float2*) in, (float2*) out, n); }, iters));
    }
    {
        const size_t n = bytes / 4;
        report ("copy float one-shot", time_ms ([&] { copy_oneshot<float><<<(unsigned) ((n + 255) / 256), 256>>> ((const float*) in, (float*) out, n); }, iters));
    }
    report ("frame copy float2 x16, one frame per WG", time_ms ([&] { frame_copy_f2<false><<<(unsigned) frames, 256>>> ((const float2*) in, (float2*) out, (unsigned) frames); }, iters));
    report ("frame copy float4 x8,  one frame per WG", time_ms ([&] { frame_copy_f4<false><<<(unsigned) frames, 256>>> ((const float4*) in, (float4*) out, (unsigned) frames); }, iters));
    for (int bpc : { 2, 3, 4, 6, 8 })
    {
        char nm[96];
        snprintf (nm, sizeof nm, "frame copy float2 x16 persistent (%d WG/CU)", bpc);
        report (nm, time_ms ([&] { frame_copy_f2<true><<<cus * bpc, 256>>> ((const float2*) in, (float2*) out, (unsigned) frames); }, iters));
        snprintf (nm, sizeof nm, "frame copy float4 x8  persistent (%d WG/CU)", bpc);
        report (nm, time_ms ([&] { frame_copy_f4<true><<<cus * bpc, 256>>> ((const float4*) in, (float4*) out, (unsigned) frames); }, iters));
    }
    // in-place variants (the FFT API allows in == out): half the footprint
    report ("frame copy float2 x16 persistent in-place (4)", time_ms ([&] { frame_copy_f2<true><<<cus * 4, 256>>> ((const float2*) in, (float2*) in, (unsigned) frames); }, iters));
    CK (hipFree (in));
    CK (hipFree (out));
    return 0;
}
