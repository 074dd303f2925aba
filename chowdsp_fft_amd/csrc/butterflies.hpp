// In-register forward DFT butterflies of length 2, 4, 8 and 16 on float2 (re, im) values.
//
// These replace the per-pass butterflies of the reference's SIMD code
// (passf2_ps / passf4_ps, simd/chowdsp_fft_impl_sse.cpp:148-180, 221-296): there one "scalar" is a
// 4- or 8-wide vector and every radix-4 pass is a sweep over the whole frame in memory; here one
// work-item keeps 16 complex points in VGPRs and finishes log2(16) = 4 radix-2 levels per pass, so a
// 4096-point frame needs 3 passes instead of 6.
//
// Only the FORWARD (e^{-i...}) direction is implemented: the backward transform is obtained by
// swapping re<->im on load and on store (IDFT(x) = swap(DFT(swap(x)))), which costs nothing in registers.
//
// Output is in natural order: after fft_reg<R>(u), u[k] = sum_n u_in[n] * exp(-2*pi*i*n*k/R).
#pragma once
#include <hip/hip_runtime.h>

// Streaming-data hint: frames are read once and written once, so global accesses to them carry the
// nontemporal hint (membench2 on MI355X: +5..6 % bandwidth for the frame-sized load-all/store-all pattern).
// Tables that every workgroup re-reads (twiddles, filter spectra) use plain loads so they stay in L2.
#ifndef CFFT_STREAM_NT
#define CFFT_STREAM_NT 1
#endif

namespace cfft
{
typedef float v2f_native __attribute__ ((ext_vector_type (2)));
typedef float v4f_native __attribute__ ((ext_vector_type (4)));

__device__ __forceinline__ float2 gload (const float2* p)
{
#if CFFT_STREAM_NT
    const v2f_native v = __builtin_nontemporal_load (reinterpret_cast<const v2f_native*> (p));
    return make_float2 (v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void gstore (float2* p, float2 v)
{
#if CFFT_STREAM_NT
    v2f_native n;
    n.x = v.x;
    n.y = v.y;
    __builtin_nontemporal_store (n, reinterpret_cast<v2f_native*> (p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ float4 gload4 (const float4* p)
{
#if CFFT_STREAM_NT
    const v4f_native v = __builtin_nontemporal_load (reinterpret_cast<const v4f_native*> (p));
    return make_float4 (v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
// 16 bytes at an address that is only 8-byte aligned (two adjacent complex values that start at an odd index): the same
// global_load_dwordx4 / global_store_dwordx4 -- the hardware asks for dword alignment -- through a type that says so
typedef float v4f_native_a8 __attribute__ ((ext_vector_type (4), aligned (8)));
__device__ __forceinline__ float4 gload4_a8 (const float2* p)
{
    const v4f_native_a8 v = __builtin_nontemporal_load (reinterpret_cast<const v4f_native_a8*> (p));
    return make_float4 (v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load4_a8 (const float2* p) // plain (cached) form: tables that stay in L2
{
    const v4f_native_a8 v = *reinterpret_cast<const v4f_native_a8*> (p);
    return make_float4 (v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstore4_a8 (float2* p, float4 v)
{
    v4f_native_a8 n;
    n.x = v.x;
    n.y = v.y;
    n.z = v.z;
    n.w = v.w;
    __builtin_nontemporal_store (n, reinterpret_cast<v4f_native_a8*> (p));
}
__device__ __forceinline__ void gstore4 (float4* p, float4 v)
{
#if CFFT_STREAM_NT
    v4f_native n;
    n.x = v.x;
    n.y = v.y;
    n.z = v.z;
    n.w = v.w;
    __builtin_nontemporal_store (n, reinterpret_cast<v4f_native*> (p));
#else
    *p = v;
#endif
}

// Complex arithmetic.  CFFT_PACKED = 1 (default): written on the native 2 x f32 vector inside HIP's float2 so that
// every complex add / sub is ONE v_pk_add_f32 and every complex multiply is v_pk_mul_f32 + v_pk_fma_f32 (op_sel
// swizzles and neg modifiers do the re/im shuffling for free) -- about 45 % fewer VALU instructions than the
// scalar form, which matters because the fused kernels are VALU-issue bound, not only HBM bound.
#ifndef CFFT_PACKED
#define CFFT_PACKED 1
#endif

static constexpr float kSqrtHalf = 0.70710678118654752440f;
static constexpr float kCosPi8 = 0.92387953251128675613f;
static constexpr float kSinPi8 = 0.38268343236508977173f;

#if CFFT_PACKED
__device__ __forceinline__ float2 from_native (v2f_native d) { return make_float2 (d.x, d.y); }
__device__ __forceinline__ v2f_native nat (float2 a) { return (v2f_native) { a.x, a.y }; }
__device__ __forceinline__ v2f_native pkfma (v2f_native a, v2f_native b, v2f_native c) { return __builtin_elementwise_fma (a, b, c); }

__device__ __forceinline__ float2 cadd (float2 a, float2 b) { return from_native (nat (a) + nat (b)); }
__device__ __forceinline__ float2 csub (float2 a, float2 b) { return from_native (nat (a) - nat (b)); }
// a * b = a * b.xx + a.yx * (-b.y, b.y)
__device__ __forceinline__ float2 cmul (float2 a, float2 b)
{
    const v2f_native t = nat (a).yx * (v2f_native) { -b.y, b.y };
    return from_native (pkfma (nat (a), nat (b).xx, t));
}
// a * conj(b) = a * b.xx + a.yx * (b.y, -b.y)
__device__ __forceinline__ float2 cmulc (float2 a, float2 b)
{
    const v2f_native t = nat (a).yx * (v2f_native) { b.y, -b.y };
    return from_native (pkfma (nat (a), nat (b).xx, t));
}
// a * (-i) = (a.y, -a.x)
__device__ __forceinline__ float2 mul_mi (float2 a) { return from_native (nat (a).yx * (v2f_native) { 1.f, -1.f }); }
// a * (+i) = (-a.y, a.x)
__device__ __forceinline__ float2 mul_pi (float2 a) { return from_native (nat (a).yx * (v2f_native) { -1.f, 1.f }); }
// c + a*(-i)  and  c - a*(-i): the rotation rides on the fma's swizzle
__device__ __forceinline__ float2 add_mi (float2 c, float2 a) { return from_native (pkfma (nat (a).yx, (v2f_native) { 1.f, -1.f }, nat (c))); }
__device__ __forceinline__ float2 sub_mi (float2 c, float2 a) { return from_native (pkfma (nat (a).yx, (v2f_native) { -1.f, 1.f }, nat (c))); }
// a * W8^1 = a * (1 - i)/sqrt2 = (a + a*(-i)) / sqrt2
__device__ __forceinline__ float2 mul_w8_1 (float2 a)
{
    return from_native (pkfma (nat (a).yx, (v2f_native) { 1.f, -1.f }, nat (a)) * kSqrtHalf);
}
// a * W8^3 = a * (-1 - i)/sqrt2 = (a*(-i) - a) / sqrt2
__device__ __forceinline__ float2 mul_w8_3 (float2 a)
{
    return from_native (pkfma (nat (a).yx, (v2f_native) { 1.f, -1.f }, -nat (a)) * kSqrtHalf);
}
#else
__device__ __forceinline__ float2 cadd (float2 a, float2 b) { return make_float2 (a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub (float2 a, float2 b) { return make_float2 (a.x - b.x, a.y - b.y); }
// a * b
__device__ __forceinline__ float2 cmul (float2 a, float2 b)
{
    return make_float2 (a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc (float2 a, float2 b)
{
    return make_float2 (a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// a * (-i)
__device__ __forceinline__ float2 mul_mi (float2 a) { return make_float2 (a.y, -a.x); }
// a * (+i)
__device__ __forceinline__ float2 mul_pi (float2 a) { return make_float2 (-a.y, a.x); }
__device__ __forceinline__ float2 add_mi (float2 c, float2 a) { return make_float2 (c.x + a.y, c.y - a.x); }
__device__ __forceinline__ float2 sub_mi (float2 c, float2 a) { return make_float2 (c.x - a.y, c.y + a.x); }
// a * W8^1 = a * (1 - i)/sqrt2
__device__ __forceinline__ float2 mul_w8_1 (float2 a) { return make_float2 ((a.x + a.y) * kSqrtHalf, (a.y - a.x) * kSqrtHalf); }
// a * W8^3 = a * (-1 - i)/sqrt2
__device__ __forceinline__ float2 mul_w8_3 (float2 a) { return make_float2 ((a.y - a.x) * kSqrtHalf, -(a.x + a.y) * kSqrtHalf); }
#endif

__device__ __forceinline__ void bfly2 (float2& a, float2& b)
{
    const float2 t = csub (a, b);
    a = cadd (a, b);
    b = t;
}

// 4-point forward DFT, natural order in / natural order out.
__device__ __forceinline__ void bfly4 (float2& x0, float2& x1, float2& x2, float2& x3)
{
    const float2 s02 = cadd (x0, x2), d02 = csub (x0, x2);
    const float2 s13 = cadd (x1, x3), d13 = csub (x1, x3);
    x0 = cadd (s02, s13);
    x1 = add_mi (d02, d13); // d02 + (-i) d13
    x2 = csub (s02, s13);
    x3 = sub_mi (d02, d13); // d02 - (-i) d13
}

template <int R>
__device__ __forceinline__ void fft_reg (float2 (&u)[R]);

template <>
__device__ __forceinline__ void fft_reg<2> (float2 (&u)[2])
{
    bfly2 (u[0], u[1]);
}

template <>
__device__ __forceinline__ void fft_reg<4> (float2 (&u)[4])
{
    bfly4 (u[0], u[1], u[2], u[3]);
}

// 8 = 2 (n2) x 4 (n1):  X[2*k1 + k2] = sum_{n1<4} W4^{n1 k1} [ W8^{n1 k2} sum_{n2<2} x[n1 + 4 n2] W2^{n2 k2} ]
template <>
__device__ __forceinline__ void fft_reg<8> (float2 (&u)[8])
{
    bfly2 (u[0], u[4]);
    bfly2 (u[1], u[5]);
    bfly2 (u[2], u[6]);
    bfly2 (u[3], u[7]);
    // k2 = 1 column gets W8^{n1}
    u[5] = mul_w8_1 (u[5]);
    u[6] = mul_mi (u[6]);
    u[7] = mul_w8_3 (u[7]);
    bfly4 (u[0], u[1], u[2], u[3]); // k2 = 0 -> X[0], X[2], X[4], X[6]
    bfly4 (u[4], u[5], u[6], u[7]); // k2 = 1 -> X[1], X[3], X[5], X[7]
    // reorder to natural order
    const float2 e0 = u[0], e1 = u[1], e2 = u[2], e3 = u[3];
    const float2 o0 = u[4], o1 = u[5], o2 = u[6], o3 = u[7];
    u[0] = e0; u[1] = o0; u[2] = e1; u[3] = o1;
    u[4] = e2; u[5] = o2; u[6] = e3; u[7] = o3;
}

// 16 = 4 (n2) x 4 (n1):  X[4*k1 + k2] = sum_{n1<4} W4^{n1 k1} [ W16^{n1 k2} sum_{n2<4} x[n1 + 4 n2] W4^{n2 k2} ]
template <>
__device__ __forceinline__ void fft_reg<16> (float2 (&u)[16])
{
    // inner 4-point DFTs over n2 for each n1 : A[n1][k2] lands in u[n1 + 4*k2]
    bfly4 (u[0], u[4], u[8], u[12]);
    bfly4 (u[1], u[5], u[9], u[13]);
    bfly4 (u[2], u[6], u[10], u[14]);
    bfly4 (u[3], u[7], u[11], u[15]);
    // twiddles W16^{n1*k2} on u[n1 + 4*k2]
    const float2 w1 = make_float2 (kCosPi8, -kSinPi8); // W16^1
    const float2 w3 = make_float2 (kSinPi8, -kCosPi8); // W16^3
    u[5] = cmul (u[5], w1);          // n1=1,k2=1 : W^1
    u[6] = mul_w8_1 (u[6]);          // n1=2,k2=1 : W^2
    u[7] = cmul (u[7], w3);          // n1=3,k2=1 : W^3
    u[9] = mul_w8_1 (u[9]);          // n1=1,k2=2 : W^2
    u[10] = mul_mi (u[10]);          // n1=2,k2=2 : W^4
    u[11] = mul_w8_3 (u[11]);        // n1=3,k2=2 : W^6
    u[13] = cmul (u[13], w3);        // n1=1,k2=3 : W^3
    u[14] = mul_w8_3 (u[14]);        // n1=2,k2=3 : W^6
    {                                // n1=3,k2=3 : W^9 = -W^1
        const float2 t = cmul (u[15], w1);
        u[15] = make_float2 (-t.x, -t.y);
    }
    // outer 4-point DFTs over n1 for each k2 : X[4*k1 + k2] lands in u[4*k2 + k1]
    bfly4 (u[0], u[1], u[2], u[3]);
    bfly4 (u[4], u[5], u[6], u[7]);
    bfly4 (u[8], u[9], u[10], u[11]);
    bfly4 (u[12], u[13], u[14], u[15]);
    // transpose the 4x4 register tile so that u[k] = X[k]
    float2 t;
    t = u[1]; u[1] = u[4]; u[4] = t;
    t = u[2]; u[2] = u[8]; u[8] = t;
    t = u[3]; u[3] = u[12]; u[12] = t;
    t = u[6]; u[6] = u[9]; u[9] = t;
    t = u[7]; u[7] = u[13]; u[13] = t;
    t = u[11]; u[11] = u[14]; u[14] = t;
}
} // namespace cfft
