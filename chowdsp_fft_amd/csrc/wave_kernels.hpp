// N = 4096 complex points by ONE WAVEFRONT per frame: 64 lanes x 64 points, 4096 = 64 x 64.
//
// The workgroup form (c2c_kernel_pf: 256 work-items x 16 points, 16 x 16 x 16) needs two exchanges and four workgroup
// barriers per frame and keeps 2 frames (+ 2 prefetched) in flight per CU.  Here a frame never leaves its wavefront:
// two radix-64 stages in registers, ONE exchange through a wave-private LDS plane -- one component at a time, so that
// the plane is 16 KiB + padding and eight waves (two 256-thread workgroups) fit a CU -- and no workgroup barrier at all
// (LDS operations of one wave execute in order).  Eight frames per CU are in flight, each with its own 32 KiB of loads.
// The stage-1 twiddles W_4096^(a t), a < 64, depend only on the lane: they live in registers for the whole launch, as
// 7 + 7 values with the radix-64 butterfly split 8 x 8 (a = a_lo + 8 a_hi; the trick of split_plane_kernels.hpp).
// Replaces the reference's pass loop + finalize + zreorder (simd/chowdsp_fft_impl_sse.cpp:372-496, 1466-1515) per frame.
#pragma once
#include "stockham.hpp"

namespace cfft
{
constexpr int kWaveN = 4096;
constexpr int kWavePlaneFloats = kWaveN + 64; // one float of padding after every 64

// cos / sin of pi * m / 32
__device__ __forceinline__ constexpr float cos32 (int m)
{
    constexpr float c[17] = { 1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f,
                              0.88192126434835502971f, 0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f,
                              0.63439328416364549822f, 0.55557023301960222474f, 0.47139673682599764856f, 0.38268343236508977173f,
                              0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f, 0.0f };
    m &= 63;
    if (m > 32)
        m = 64 - m;
    return m <= 16 ? c[m] : -c[32 - m];
}
__device__ __forceinline__ constexpr float sin32 (int m) { return cos32 (m - 16); }

template <int M>
__device__ __forceinline__ float2 mul_w64 (float2 u)
{
    constexpr int m = M & 63;
    if constexpr (m == 0)
        return u;
    else if constexpr (m == 16)
        return mul_mi (u);
    else if constexpr (m == 32)
        return make_float2 (-u.x, -u.y);
    else if constexpr (m == 48)
        return mul_pi (u);
    else if constexpr (m == 8)
        return mul_w8_1 (u);
    else if constexpr (m == 24)
        return mul_w8_3 (u);
    else
        return cmul (u, make_float2 (cos32 (m), -sin32 (m)));
}

template <int N1>
__device__ __forceinline__ void w64_row (float2 (&u)[64])
{
    // u[N1 + 8 k2] *= W64^{N1 k2}, k2 = 1..7
    u[N1 + 8] = mul_w64<N1 * 1> (u[N1 + 8]);
    u[N1 + 16] = mul_w64<N1 * 2> (u[N1 + 16]);
    u[N1 + 24] = mul_w64<N1 * 3> (u[N1 + 24]);
    u[N1 + 32] = mul_w64<N1 * 4> (u[N1 + 32]);
    u[N1 + 40] = mul_w64<N1 * 5> (u[N1 + 40]);
    u[N1 + 48] = mul_w64<N1 * 6> (u[N1 + 48]);
    u[N1 + 56] = mul_w64<N1 * 7> (u[N1 + 56]);
}

// 64-point forward DFT in registers, natural order in and out, 64 = 8 (n1) x 8 (n2):
//     X[8 k1 + k2] = sum_{n1<8} W8^{n1 k1} [ W64^{n1 k2} sum_{n2<8} x[n1 + 8 n2] W8^{n2 k2} ]
// TW: inputs first multiplied by W^(a k), a = n1 + 8 n2: whi[n2 - 1] = W^(8 n2 k) before the inner transforms,
// wlo[n1 - 1] = W^(n1 k) after them.
template <bool TW>
__device__ __forceinline__ void fft_reg64 (float2 (&u)[64], const float2 (&wlo)[7], const float2 (&whi)[7])
{
    if constexpr (TW)
    {
#pragma unroll
        for (int n2 = 1; n2 < 8; ++n2)
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1)
                u[n1 + 8 * n2] = cmul (u[n1 + 8 * n2], whi[n2 - 1]);
    }
    // inner 8-point DFTs over n2 for each n1: A[n1][k2] lands in u[n1 + 8 k2]
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
    {
        float2 a[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2)
            a[n2] = u[n1 + 8 * n2];
        fft_reg<8> (a);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2)
            u[n1 + 8 * k2] = a[k2];
    }
    if constexpr (TW)
    {
#pragma unroll
        for (int n1 = 1; n1 < 8; ++n1)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2)
                u[n1 + 8 * k2] = cmul (u[n1 + 8 * k2], wlo[n1 - 1]);
    }
    w64_row<1> (u);
    w64_row<2> (u);
    w64_row<3> (u);
    w64_row<4> (u);
    w64_row<5> (u);
    w64_row<6> (u);
    w64_row<7> (u);
    // outer 8-point DFTs over n1 for each k2: X[8 k1 + k2]
    float2 o[64];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
    {
        float2 a[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1)
            a[n1] = u[8 * k2 + n1];
        fft_reg<8> (a);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
            o[8 * k1 + k2] = a[k1];
    }
#pragma unroll
    for (int i = 0; i < 64; ++i)
        u[i] = o[i];
}

__device__ __forceinline__ void wave_lds_fence()
{
    // one wavefront: its LDS operations execute in order; only the compiler has to be kept from moving them
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
}

// frames handled by wave w of workgroup b: (b * 4 + w) + k * (gridDim.x * 4)
template <bool INV>
__global__ __launch_bounds__ (256, 2) void c2c_wave4096_kernel (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ tw, // [14][64]: W_4096^(a t), a = 1..7, 8, 16, .. 56
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float wave_planes[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* __restrict__ plane = wave_planes + wave * kWavePlaneFloats;
    float2 wlo[7], whi[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
    {
        wlo[j] = tw[j * 64 + lane];
        whi[j] = tw[(7 + j) * 64 + lane];
    }
    // exchange addresses: write phys (64 t + b) = 65 t + b, read phys (t + 64 a) = t + 65 a
    float* __restrict__ wp = plane + 65 * lane;
    const float* __restrict__ rp = plane + lane;
    const unsigned nwaves = gridDim.x * 4u;
    for (unsigned g = blockIdx.x * 4u + (unsigned) wave; g < batch; g += nwaves)
    {
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        int l = lane;
        CFFT_LAUNDER_INDEX (l); // keeps the 2 x 64 global offsets from being formed once, outside the loop, and held (stockham.hpp)
        float2 v[64];
#pragma unroll
        for (int a = 0; a < 64; ++a)
        {
            const float2 x = gload (&src[(unsigned) (l + 64 * a)]);
            v[a] = INV ? make_float2 (x.y, x.x) : x;
        }
        CFFT_PHASE_FENCE();
        fft_reg64<false> (v, wlo, whi);
        CFFT_PHASE_FENCE();
        wave_lds_fence(); // the previous frame's reads of the plane are done
#pragma unroll
        for (int b = 0; b < 64; ++b)
            wp[b] = v[b].x;
        wave_lds_fence();
#pragma unroll
        for (int a = 0; a < 64; ++a)
            v[a].x = rp[65 * a];
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 64; ++b)
            wp[b] = v[b].y;
        wave_lds_fence();
#pragma unroll
        for (int a = 0; a < 64; ++a)
            v[a].y = rp[65 * a];
        CFFT_PHASE_FENCE();
        fft_reg64<true> (v, wlo, whi);
        CFFT_PHASE_FENCE();
        int ls = lane;
        CFFT_LAUNDER_INDEX (ls);
#pragma unroll
        for (int b = 0; b < 64; ++b)
            gstore (&dst[(unsigned) (ls + 64 * b)], INV ? make_float2 (v[b].y, v[b].x) : v[b]);
    }
}
} // namespace cfft
