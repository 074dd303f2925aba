// Fifth calibration micro-benchmark (round 4): does LDS-DMA input (global_load_lds_dwordx4, nontemporal) raise the ceiling
// of the headline kernel's access pattern?  The N = 4096 transform is a persistent 256-thread workgroup (two per CU) that
// reads one whole 32 KiB frame, transforms it and writes it, with the NEXT frame's loads in flight meanwhile
// (c2c_kernel_pf).  With the arithmetic removed that structure copied at 6.05 TB/s (tools/membench2.hip, round 1) against
// 6.6-6.8 TB/s of a one-shot float4 copy and the 6.4-6.8 TB/s the CDNA4 guide quotes for LDS-DMA streams.  This program
// runs the SAME persistent frame loop three ways:
//   reg   next frame into 32 VGPRs per work-item (16 x global_load_dwordx2 nt), as the kernel does
//   dma   next frame into a 32 KiB LDS landing buffer by 8 x global_load_lds_dwordx4 nt per work-item... (2 per wave-row),
//         read back by 16 x ds_read_b64 (the kernel's stage-0 operand pattern x[t + 256 s]) -- two extra barriers per frame
//   dma2  the same with TWO landing buffers (two frames ahead; 64 KiB + the kernel's 34 KiB image = one workgroup per CU only:
//         run at one and at two workgroups per CU to separate the effects)
// and `work` dummy FMAs per element stand in for the transform's arithmetic (0 = pure copy).
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench5.hip -o tools/membench5      Run: tools/membench5
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                       \
    do                                                                                              \
    {                                                                                               \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
        {                                                                                           \
            printf ("HIP error %s at %s:%d\n", hipGetErrorString (e_), __FILE__, __LINE__);         \
            exit (1);                                                                               \
        }                                                                                           \
    } while (0)

typedef float v2f __attribute__ ((ext_vector_type (2)));
#define float2 v2f // (the nontemporal builtins want a native vector type)
constexpr int kN = 4096;       // complex points per frame
constexpr int kT = 256;        // work-items per workgroup
constexpr int kR = kN / kT;    // 16 points per work-item

__device__ __forceinline__ float2 work_on (float2 v, int work)
{
    for (int i = 0; i < work; ++i)
    {
        v.x = fmaf (v.x, 1.0000001f, 1e-9f);
        v.y = fmaf (v.y, 0.9999999f, -1e-9f);
    }
    return v;
}

// ---- reg: the kernel's own structure --------------------------------------------------------------------------------
__global__ __launch_bounds__ (kT, 2) void copy_reg (const float2* __restrict__ in, float2* __restrict__ out, unsigned frames, int work, unsigned lds_pad_bytes)
{
    extern __shared__ float2 pad[]; // (occupies the LDS the real kernel's image would: fixes the workgroups per CU)
    const int t = threadIdx.x;
    if (lds_pad_bytes == 1)
        pad[t] = (v2f) { 0.f, 0.f };
    float2 nx[kR];
    unsigned f = blockIdx.x;
    if (f < frames)
    {
#pragma unroll
        for (int s = 0; s < kR; ++s)
            nx[s] = __builtin_nontemporal_load (&in[(size_t) f * kN + t + kT * s]);
    }
    for (; f < frames; f += gridDim.x)
    {
        float2 v[kR];
#pragma unroll
        for (int s = 0; s < kR; ++s)
            v[s] = nx[s];
        const unsigned fn = f + gridDim.x;
        if (fn < frames)
        {
#pragma unroll
            for (int s = 0; s < kR; ++s)
                nx[s] = __builtin_nontemporal_load (&in[(size_t) fn * kN + t + kT * s]);
        }
#pragma unroll
        for (int s = 0; s < kR; ++s)
            v[s] = work_on (v[s], work);
#pragma unroll
        for (int s = 0; s < kR; ++s)
            __builtin_nontemporal_store (v[s], &out[(size_t) f * kN + t + kT * s]);
    }
}

// one frame by LDS-DMA: 8 instructions per work-item, each wave-instruction lands 1 KiB contiguously (lane-linear)
__device__ __forceinline__ void dma_frame (const float2* __restrict__ src, float2* landing, int t)
{
    // 32 KiB = 2048 x 16 B; work-item t moves 16-byte pieces t + 256 j, j < 8
    const float4* __restrict__ s4 = reinterpret_cast<const float4*> (src);
    float4* l4 = reinterpret_cast<float4*> (landing);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds (&s4[t + kT * j], (__attribute__ ((address_space (3))) void*) (l4 + (t & ~63) + kT * j), 16, 0, 2 /* nt */);
}

template <int BUFS>
__global__ __launch_bounds__ (kT, 2) void copy_dma (const float2* __restrict__ in, float2* __restrict__ out, unsigned frames, int work, unsigned lds_pad_bytes)
{
    extern __shared__ __attribute__ ((aligned (16))) float2 lds[]; // BUFS landing buffers of kN float2 (+ padding to fix the occupancy)
    const int t = threadIdx.x;
    unsigned f = blockIdx.x;
    // prologue: BUFS frames in flight
#pragma unroll
    for (int b = 0; b < BUFS; ++b)
        if (f + b * gridDim.x < frames)
            dma_frame (in + (size_t) (f + b * gridDim.x) * kN, lds + b * kN, t);
    unsigned k = 0;
    for (; f < frames; f += gridDim.x, ++k)
    {
        float2* landing = lds + (BUFS == 1 ? 0 : (k % BUFS)) * kN;
        // this frame has landed: all but the DMAs of the frames behind it and the previous frame's stores
        // (vmcnt counts loads, stores and LDS-DMA together, in order: [dma k][dma k+1 ..][stores k-1] -> wait for dma k only)
        if (BUFS == 1)
            asm volatile ("s_waitcnt vmcnt(16)" ::: "memory"); // 16 stores of frame k-1 may stay out (first iteration: nothing there, harmless)
        else
            asm volatile ("s_waitcnt vmcnt(24)" ::: "memory"); // 8 DMAs of frame k+1 + 16 stores of frame k-1
        __builtin_amdgcn_s_barrier();
        float2 v[kR];
#pragma unroll
        for (int s = 0; s < kR; ++s)
            v[s] = landing[t + kT * s];
        asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); // everybody has read the buffer: it may be refilled
        const unsigned fn = f + BUFS * gridDim.x;
        if (fn < frames)
            dma_frame (in + (size_t) fn * kN, landing, t);
#pragma unroll
        for (int s = 0; s < kR; ++s)
            v[s] = work_on (v[s], work);
#pragma unroll
        for (int s = 0; s < kR; ++s)
            __builtin_nontemporal_store (v[s], &out[(size_t) f * kN + t + kT * s]);
    }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class F>
float time_ms (F&& launch, int iters)
{
    hipEvent_t a, b;
    CK (hipEventCreate (&a));
    CK (hipEventCreate (&b));
    for (int i = 0; i < 5; ++i)
        launch();
    CK (hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep)
    {
        CK (hipEventRecord (a));
        for (int i = 0; i < iters; ++i)
            launch();
        CK (hipEventRecord (b));
        CK (hipEventSynchronize (b));
        float ms = 0;
        CK (hipEventElapsedTime (&ms, a, b));
        best = ms / iters < best ? ms / iters : best;
    }
    return best;
}

int main()
{
    const unsigned frames = 65536; // 2 GiB in + 2 GiB out: the headline batch
    const size_t bytes = (size_t) frames * kN * sizeof (float2);
    float2 *in, *out;
    CK (hipMalloc (&in, bytes));
    CK (hipMalloc (&out, bytes));
    CK (hipMemset (in, 1, bytes));
    CK (hipMemset (out, 0, bytes));
    hipDeviceProp_t prop;
    CK (hipGetDeviceProperties (&prop, 0));
    const int cus = prop.multiProcessorCount;
    CK (hipFuncSetAttribute ((const void*) copy_dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK (hipFuncSetAttribute ((const void*) copy_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK (hipFuncSetAttribute ((const void*) copy_reg, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf ("# persistent frame copy, N = 4096 complex (32 KiB frames), %u frames, %d CUs; GB/s = read + written bytes / time (best of 5 x 10 launches)\n", frames, cus);
    for (int work : { 0, 8, 16, 24 })
    {
        for (int wg_per_cu : { 2, 1 })
        {
            const unsigned grid = (unsigned) (cus * wg_per_cu);
            // LDS footprint chosen so that exactly wg_per_cu workgroups fit a CU: 66 KiB (two) / 100 KiB (one)
            const unsigned lds = wg_per_cu == 2 ? 66 * 1024 : 100 * 1024;
            float ms = time_ms ([&] { copy_reg<<<grid, kT, lds>>> (in, out, frames, work, lds); }, 10);
            printf ("work %2d  %d wg/CU  reg   %7.3f ms  %7.1f GB/s\n", work, wg_per_cu, ms, 2.0 * bytes / ms * 1e-6);
            ms = time_ms ([&] { copy_dma<1><<<grid, kT, lds>>> (in, out, frames, work, lds); }, 10);
            printf ("work %2d  %d wg/CU  dma   %7.3f ms  %7.1f GB/s\n", work, wg_per_cu, ms, 2.0 * bytes / ms * 1e-6);
            if (wg_per_cu == 1 || true)
            {
                const unsigned lds2 = wg_per_cu == 2 ? 80 * 1024 : 100 * 1024; // two landing buffers = 64 KiB (two workgroups: without the image's 34 KiB)
                ms = time_ms ([&] { copy_dma<2><<<grid, kT, lds2>>> (in, out, frames, work, lds2); }, 10);
                printf ("work %2d  %d wg/CU  dma2  %7.3f ms  %7.1f GB/s   (%u KiB LDS per workgroup)\n", work, wg_per_cu, ms, 2.0 * bytes / ms * 1e-6, lds2 / 1024);
            }
            fflush (stdout);
        }
    }
    // sanity: the copies copy (a fresh output buffer, the single-buffer DMA form without arithmetic)
    CK (hipMemset (out, 0, bytes));
    copy_dma<1><<<(unsigned) (cus * 2), kT, 66 * 1024>>> (in, out, frames, 0, 0);
    CK (hipDeviceSynchronize());
    float2 h[4];
    CK (hipMemcpy (h, out + (size_t) (frames - 1) * kN + kN - 4, sizeof (h), hipMemcpyDeviceToHost));
    printf ("# last elements %g %g (expected the memset pattern %g)\n", h[3].x, h[3].y, [] { unsigned u = 0x01010101u; float f; memcpy (&f, &u, 4); return f; }());
    return 0;
}
