// Read-mostly bandwidth calibration for the delay-line MAC kernel (fdl_mac_kernel): each work-item reads S
// float4 streams `slot` bytes apart, adds them, writes one float4.  Tells how close 5.0 TB/s is to what a
// 16-reads-per-write pattern can reach on MI355X.
// build: hipcc -O3 --offload-arch=gfx950 tools/membench3.hip -o tools/membench3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__ ((ext_vector_type (4)));

template <int S, bool NT>
__global__ __launch_bounds__ (256) void multi_read (const f4* __restrict__ in, f4* __restrict__ out, size_t quads_per_slot)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < quads_per_slot; i += stride)
    {
        f4 acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = 0; s < S; ++s)
            acc += NT ? __builtin_nontemporal_load (&in[(size_t) s * quads_per_slot + i]) : in[(size_t) s * quads_per_slot + i];
        if (NT)
            __builtin_nontemporal_store (acc, &out[i]);
        else
            out[i] = acc;
    }
}

template <int S, bool NT>
void run (const f4* in, f4* out, size_t quads, int grid, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate (&e0);
    hipEventCreate (&e1);
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL ((multi_read<S, NT>), dim3 (grid), dim3 (256), 0, 0, in, out, quads);
    hipEventRecord (e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL ((multi_read<S, NT>), dim3 (grid), dim3 (256), 0, 0, in, out, quads);
    hipEventRecord (e1);
    hipEventSynchronize (e1);
    float ms = 0;
    hipEventElapsedTime (&ms, e0, e1);
    ms /= reps;
    const double bytes = (double) (S + 1) * quads * 16;
    printf ("%-40s S=%2d grid=%5d  %7.3f ms  %7.1f GB/s (reads + 1 write)\n", name, S, grid, ms, bytes / ms * 1e-6);
}

int main()
{
    const size_t slot_bytes = 128ull << 20; // one delay-line slot of the bench workload: 8192 channels x 16 KiB
    const size_t quads = slot_bytes / 16;
    f4 *in = nullptr, *out = nullptr;
    if (hipMalloc ((void**) &in, 16 * slot_bytes) != hipSuccess || hipMalloc ((void**) &out, slot_bytes) != hipSuccess)
        return 1;
    hipMemset (in, 0, 16 * slot_bytes);
    for (int grid : { 2048, 4096, 8192, 32768 })
    {
        run<1, false> (in, out, quads, grid, "plain");
        run<4, false> (in, out, quads, grid, "plain");
        run<16, false> (in, out, quads, grid, "plain");
        run<1, true> (in, out, quads, grid, "nontemporal");
        run<4, true> (in, out, quads, grid, "nontemporal");
        run<16, true> (in, out, quads, grid, "nontemporal");
    }
    hipFree (in);
    hipFree (out);
    return 0;
}
