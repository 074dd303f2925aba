// Write-path calibration for the two-pass bound: how fast can this chip WRITE when nothing, or only an
// L2-resident table, is read?  If write-only traffic runs well above the 3.4 TB/s that each direction of a copy carries, a
// two-pass transform (8 GiB of stores per 4 GiB of frames) is not bounded by the store path itself.
//   fill        : out[i] = const                      (4 GiB of stores, no loads)
//   l2_to_hbm   : out[i] = small[i mod 1 MiB]         (loads that hit the XCD's L2, 4 GiB of stores) -- the shape of pass B
//                                                       of the one-launch two-pass form
//   copy        : out[i] = in[i]                      (4 GiB + 4 GiB) for reference
//   tile copy   : persistent workgroups move 32 KiB tiles (16 x float2 per work-item, all loads, then all stores)
// build: hipcc -O3 --offload-arch=gfx950 tools/membench4.hip -o tools/membench4
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__ ((ext_vector_type (4)));
typedef float f2 __attribute__ ((ext_vector_type (2)));

template <int MODE, bool NT>
__global__ __launch_bounds__ (256) void stream_kernel (const f4* __restrict__ in, const f4* __restrict__ small, f4* __restrict__ out, size_t quads)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride)
    {
        f4 v = { 1.f, 2.f, 3.f, 4.f };
        if (MODE == 1)
            v = small[i & 0xFFFF]; // 1 MiB table
        if (MODE == 2)
            v = NT ? __builtin_nontemporal_load (&in[i]) : in[i];
        if (NT)
            __builtin_nontemporal_store (v, &out[i]);
        else
            out[i] = v;
    }
}

// persistent workgroups, one 32 KiB tile per iteration: U loads per work-item, then U stores (the FFT kernels' shape)
template <int MODE, bool NT>
__global__ __launch_bounds__ (256) void tile_kernel (const f2* __restrict__ in, const f2* __restrict__ small, f2* __restrict__ out, unsigned tiles)
{
    for (unsigned g = blockIdx.x; g < tiles; g += gridDim.x)
    {
        f2 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a)
        {
            if (MODE == 0)
                v[a] = f2 { (float) a, 1.f };
            else if (MODE == 1)
                v[a] = small[(threadIdx.x + 256 * a + 4096 * (g & 31)) & 0x1FFFF];
            else
                v[a] = NT ? __builtin_nontemporal_load (&in[(size_t) g * 4096 + threadIdx.x + 256 * a]) : in[(size_t) g * 4096 + threadIdx.x + 256 * a];
        }
#pragma unroll
        for (int a = 0; a < 16; ++a)
        {
            if (NT)
                __builtin_nontemporal_store (v[a], &out[(size_t) g * 4096 + threadIdx.x + 256 * a]);
            else
                out[(size_t) g * 4096 + threadIdx.x + 256 * a] = v[a];
        }
    }
}

template <class F>
float time_it (F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate (&e0);
    hipEventCreate (&e1);
    for (int w = 0; w < 3; ++w)
        launch();
    hipEventRecord (e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r)
        launch();
    hipEventRecord (e1);
    hipEventSynchronize (e1);
    float ms = 0;
    hipEventElapsedTime (&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const size_t bytes = 4ull << 30;
    const size_t quads = bytes / 16;
    f4 *in = nullptr, *out = nullptr, *small = nullptr;
    if (hipMalloc ((void**) &in, bytes) != hipSuccess || hipMalloc ((void**) &out, bytes) != hipSuccess || hipMalloc ((void**) &small, 1 << 20) != hipSuccess)
        return 1;
    hipMemset (in, 0, bytes);
    hipMemset (small, 0, 1 << 20);
    const char* names[3] = { "fill (stores only)", "L2 table -> HBM", "copy" };
    for (int grid : { 4096, 16384, 65536 })
    {
#define RUN(MODE, NT) \
    { \
        const float ms = time_it ([&] { hipLaunchKernelGGL ((stream_kernel<MODE, NT>), dim3 (grid), dim3 (256), 0, 0, in, small, out, quads); }); \
        printf ("stream %-20s %-3s grid=%6d  %7.3f ms  stores %7.1f GB/s  total %7.1f GB/s\n", names[MODE], NT ? "nt" : "", grid, ms, bytes / ms * 1e-6, (MODE == 2 ? 2.0 : 1.0) * bytes / ms * 1e-6); \
    }
        RUN (0, false) RUN (0, true) RUN (1, false) RUN (1, true) RUN (2, false) RUN (2, true)
#undef RUN
    }
    const unsigned tiles = (unsigned) (bytes / 32768);
    for (int grid : { 512, 768, 1024, 2048 })
    {
#define RUNT(MODE, NT) \
    { \
        const float ms = time_it ([&] { hipLaunchKernelGGL ((tile_kernel<MODE, NT>), dim3 (grid), dim3 (256), 0, 0, (const f2*) in, (const f2*) small, (f2*) out, tiles); }); \
        printf ("tiles  %-20s %-3s grid=%6d  %7.3f ms  stores %7.1f GB/s  total %7.1f GB/s\n", names[MODE], NT ? "nt" : "", grid, ms, bytes / ms * 1e-6, (MODE == 2 ? 2.0 : 1.0) * bytes / ms * 1e-6); \
    }
        RUNT (0, false) RUNT (0, true) RUNT (1, false) RUNT (1, true) RUNT (2, false) RUNT (2, true)
#undef RUNT
    }
    hipFree (in);
    hipFree (out);
    hipFree (small);
    return 0;
}
