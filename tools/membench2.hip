// Second calibration micro-benchmark: what about the FFT kernels' "load a whole frame, then store a
// whole frame" structure costs bandwidth relative to a plain one-shot float4 copy?
// Sweeps: elements per work-item U (depth of loads in flight per wave), access width, batch vs interleaved
// load/store order, nontemporal hints, workgroup size.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/membench2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__ ((ext_vector_type (4)));
typedef float v2f __attribute__ ((ext_vector_type (2)));

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

// MODE 0: load U, then store U (frame-copy structure).  MODE 1: load 1 / store 1, U times.
// NT: nontemporal loads and stores.  Each workgroup owns one contiguous chunk of BLOCK*U elements.
template <typename T, int U, int BLOCK, int MODE, bool NT>
__global__ __launch_bounds__ (BLOCK) void chunk_copy (const T* __restrict__ in, T* __restrict__ out)
{
    const size_t base = (size_t) blockIdx.x * (BLOCK * U) + threadIdx.x;
    if (MODE == 0)
    {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = NT ? __builtin_nontemporal_load (&in[base + (size_t) u * BLOCK]) : in[base + (size_t) u * BLOCK];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (NT)
                __builtin_nontemporal_store (v[u], &out[base + (size_t) u * BLOCK]);
            else
                out[base + (size_t) u * BLOCK] = v[u];
        }
    }
    else
    {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const T v = NT ? __builtin_nontemporal_load (&in[base + (size_t) u * BLOCK]) : in[base + (size_t) u * BLOCK];
            if (NT)
                __builtin_nontemporal_store (v, &out[base + (size_t) u * BLOCK]);
            else
                out[base + (size_t) u * BLOCK] = v;
            asm volatile("" ::: "memory");
        }
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

static void* g_in;
static void* g_out;
static size_t g_bytes;

template <typename T, int U, int BLOCK, int MODE, bool NT>
void run (const char* tname)
{
    const size_t chunk = (size_t) BLOCK * U * sizeof (T);
    const unsigned grid = (unsigned) (g_bytes / chunk);
    const float ms = time_ms ([&] { chunk_copy<T, U, BLOCK, MODE, NT><<<grid, BLOCK>>> ((const T*) g_in, (T*) g_out); }, 10);
    printf ("%-7s U=%2d block=%4d chunk=%6zu B %-11s %-3s %8.3f ms %8.1f GB/s\n", tname, U, BLOCK, chunk,
            MODE == 0 ? "batch" : "interleaved", NT ? "nt" : "", ms, 2.0 * g_bytes / ms * 1e-6);
    fflush (stdout);
}

template <typename T, int BLOCK>
void sweep (const char* tname)
{
    run<T, 1, BLOCK, 0, false> (tname);
    run<T, 2, BLOCK, 0, false> (tname);
    run<T, 4, BLOCK, 0, false> (tname);
    run<T, 8, BLOCK, 0, false> (tname);
    run<T, 16, BLOCK, 0, false> (tname);
    run<T, 4, BLOCK, 1, false> (tname);
    run<T, 16, BLOCK, 1, false> (tname);
    run<T, 1, BLOCK, 0, true> (tname);
    run<T, 8, BLOCK, 0, true> (tname);
    run<T, 16, BLOCK, 0, true> (tname);
}

int main()
{
    g_bytes = 2ull << 30;
    CK (hipMalloc (&g_in, g_bytes));
    CK (hipMalloc (&g_out, g_bytes));
    CK (hipMemset (g_in, 1, g_bytes));
    CK (hipMemset (g_out, 0, g_bytes));
    sweep<v4f, 256> ("float4");
    sweep<v2f, 256> ("float2");
    sweep<v4f, 64> ("float4");
    sweep<v2f, 1024> ("float2");
    sweep<v4f, 1024> ("float4");
    return 0;
}
