// N = 32768 complex points (and real N = 65536 on top of it) in ONE pass over HBM.
//
// (Since round 2 also N = 16384 / real N = 32768 with 512 work-items: SplitGeom below.)
//
// A 32768-point frame is 256 KiB: its interleaved LDS image does not fit the CU's 160 KiB, which is why round 1 ran it as
// two passes through device memory (2 x the traffic, 2.9 TB/s).  Here one 1024-thread workgroup keeps the whole frame in
// registers -- 32 complex points per work-item -- and the two Stockham exchanges go through LDS ONE COMPONENT AT A TIME:
// the real parts of all 32768 points (128 KiB + padding), then the imaginary parts through the same 132 KiB plane.
//     32768 = 32 x 32 x 32: three radix-32 stages in registers, two exchanges, eight workgroup barriers per frame.
// Same autosort index algebra as stockham.hpp (stage s, p = product of the radices before it):
//     butterfly i = t:   k = i mod p,  j = (i - k) * 32 + k;   inputs X[i + a * 1024], outputs Y[j + b * p]
// so stage 0 reads global memory lane-contiguously, the last stage writes it lane-contiguously, and the result is in
// natural order (the reference's finalize / zreorder steps, simd/chowdsp_fft_impl_sse.cpp:434-496, 1466-1515, have no
// counterpart).
//
// Registers: 1024 work-items leave 128 VGPRs each, 64 of them hold the frame.  A radix-32 butterfly needs 31 twiddles
// W^(a k), a = 1..31; instead of loading 31 (62 VGPRs) the input index is split a = a_lo + 8 a_hi, matching the 8 x 4
// decomposition of the butterfly itself:  W^(a k) = W^(a_lo k) W^(8 a_hi k)  -- 7 + 3 table entries per butterfly,
// the a_hi factor applied before the inner 4-point transforms, the a_lo factor (common to all a_hi) after them.
// The tables (10 x 32 and 10 x 1024 entries) stay in L2; they are requested before the exchange they follow.
#pragma once
#include "real_kernels.hpp"

namespace cfft
{
// Geometry for T work-items per frame, 32 points each: N = 32 T.
//   T = 1024: N = 32768 = 32 x 32 x 32                      (one workgroup per CU, 132 KiB plane)
//   T =  512: N = 16384 = 32 x 32 x 16, the last stage as two radix-16 butterflies per work-item
//             (two workgroups per CU, 66 KiB plane each; replaces the interleaved-image plan with streamed twiddles)
//   T =  256: N =  8192 = 32 x 32 x 8, the last stage as four radix-8 butterflies per work-item (four workgroups per CU,
//             33 KiB plane each); used by the fused convolution, where the interleaved-image plan has no registers left
//   T =   64: N =  2048 = 32 x 32 x 2, ONE WAVEFRONT per frame: sixteen radix-2 butterflies per work-item, an 8.25 KiB
//             wave-private plane, no workgroup barrier that another wave takes part in (fused convolution of real N = 4096)
// Invariant: register slot c of work-item t holds element t + T c -- on input, after each exchange and on output.
template <int T_>
struct SplitGeom
{
    static constexpr int T = T_;
    static constexpr int N = 32 * T_;
    static constexpr int LAST = N / 1024;                    // radix of the last stage: 32, 16 or 8
    static constexpr int RS = T_ + T_ / 32;                  // physical distance of consecutive slots in the plane
    static constexpr int plane_floats = N + N / 32;          // one float per point, one float of padding after every 32
    static constexpr size_t lds_bytes = (size_t) plane_floats * sizeof (float);
    // twiddle table: stage 1 (p = 32): [10][32] rows a = 1..7, 8, 16, 24 of W_1024^(a k); stage 2 (p = 1024):
    // LAST = 32: [10][1024] the same rows of W_N^(a k); LAST = 16: [6][1024] rows a = 1, 2, 3, 4, 8, 12; LAST = 8: [7][1024]
    // rows a = 1..7
    static constexpr int tw_stage2 = 10 * 32;
    static constexpr int tw_rows2 = LAST == 32 ? 10 : (LAST == 16 ? 6 : LAST - 1); // LAST = 8 / 4 / 2: rows a = 1 .. LAST - 1
    static constexpr int tw_total = tw_stage2 + tw_rows2 * 1024;
};
constexpr int kSplitN = SplitGeom<1024>::N;   // the longest frame of this family
constexpr int kSplitT = 1024;

__device__ __forceinline__ constexpr int sp_phys (int idx) { return idx + (idx >> 5); }

// cos / sin of pi * m / 16
__device__ __forceinline__ constexpr float cos16 (int m)
{
    constexpr float c[9] = { 1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                             0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f };
    m &= 31;
    if (m > 16)
        m = 32 - m;
    return m <= 8 ? c[m] : -c[16 - m];
}
__device__ __forceinline__ constexpr float sin16 (int m) { return cos16 (m - 8); }

// u *= W32^M   (M a compile-time constant: multiples of 8 and 4 take the cheap forms)
template <int M>
__device__ __forceinline__ float2 mul_w32 (float2 u)
{
    constexpr int m = M & 31;
    if constexpr (m == 0)
        return u;
    else if constexpr (m == 8)
        return mul_mi (u);
    else if constexpr (m == 16)
        return make_float2 (-u.x, -u.y);
    else if constexpr (m == 24)
        return mul_pi (u);
    else if constexpr (m == 4)
        return mul_w8_1 (u);
    else if constexpr (m == 12)
        return mul_w8_3 (u);
    else
        return cmul (u, make_float2 (cos16 (m), -sin16 (m)));
}

// 32-point forward DFT in registers, natural order in and out, 32 = 8 (n1) x 4 (n2):
//     X[4 k1 + k2] = sum_{n1<8} W8^{n1 k1} [ W32^{n1 k2} sum_{n2<4} x[n1 + 8 n2] W4^{n2 k2} ]
// TW: the inputs are first multiplied by the stage twiddle W^(a k), a = n1 + 8 n2, given as wlo[n1 - 1] = W^(n1 k) and
// whi[n2 - 1] = W^(8 n2 k): whi before the inner transforms, wlo (common to the four n2) after them.
template <bool TW>
__device__ __forceinline__ void fft_reg32 (float2 (&u)[32], const float2 (&w)[10])
{
    if constexpr (TW)
    {
#pragma unroll
        for (int n2 = 1; n2 < 4; ++n2)
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1)
                u[n1 + 8 * n2] = cmul (u[n1 + 8 * n2], w[7 + n2 - 1]);
    }
    // inner 4-point DFTs over n2: A[n1][k2] lands in u[n1 + 8 k2]
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
        bfly4 (u[n1], u[n1 + 8], u[n1 + 16], u[n1 + 24]);
    if constexpr (TW)
    {
#pragma unroll
        for (int n1 = 1; n1 < 8; ++n1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
                u[n1 + 8 * k2] = cmul (u[n1 + 8 * k2], w[n1 - 1]);
    }
    // W32^{n1 k2}
#define CFFT_W32(N1, K2) u[N1 + 8 * K2] = mul_w32<N1 * K2> (u[N1 + 8 * K2]);
#define CFFT_W32_ROW(K2) CFFT_W32 (1, K2) CFFT_W32 (2, K2) CFFT_W32 (3, K2) CFFT_W32 (4, K2) CFFT_W32 (5, K2) CFFT_W32 (6, K2) CFFT_W32 (7, K2)
    CFFT_W32_ROW (1)
    CFFT_W32_ROW (2)
    CFFT_W32_ROW (3)
#undef CFFT_W32_ROW
#undef CFFT_W32
    // outer 8-point DFTs over n1 for each k2: X[4 k1 + k2]
    float2 o[32];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2)
    {
        float2 a[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1)
            a[n1] = u[8 * k2 + n1];
        fft_reg<8> (a);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
            o[4 * k1 + k2] = a[k1];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i)
        u[i] = o[i];
}

// the 10 twiddles of work-item t's radix-32 butterfly in stage S (1, or 2 when the last stage is radix 32)
template <class G, int S>
__device__ __forceinline__ void sp_load_twiddles (float2 (&w)[10], const float2* __restrict__ tw, int t)
{
    constexpr int p = (S == 1) ? 32 : 1024;
    constexpr int off = (S == 1) ? 0 : G::tw_stage2;
    const int k = t & (p - 1);
#pragma unroll
    for (int j = 0; j < 10; ++j)
        w[j] = tw[off + j * p + k];
}
// the 2 x 6 twiddles of work-item t's two radix-16 butterflies (k = t and t + 512) of the last stage of N = 16384
template <class G>
__device__ __forceinline__ void sp_load_twiddles16 (float2 (&w)[2][6], const float2* __restrict__ tw, int t)
{
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            w[q][j] = tw[G::tw_stage2 + j * 1024 + t + 512 * q];
}

// 16-point forward DFT with the stage twiddle W^(a k) applied first, a = n1 + 4 n2 (fft_reg<16> is 4 x 4):
// w[0..2] = W^(n1 k), n1 = 1..3, applied after the inner transforms; w[3..5] = W^(4 n2 k), n2 = 1..3, before them.
__device__ __forceinline__ void fft_reg16_tw (float2 (&u)[16], const float2 (&w)[6])
{
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2)
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1)
            u[n1 + 4 * n2] = cmul (u[n1 + 4 * n2], w[3 + n2 - 1]);
    bfly4 (u[0], u[4], u[8], u[12]);
    bfly4 (u[1], u[5], u[9], u[13]);
    bfly4 (u[2], u[6], u[10], u[14]);
    bfly4 (u[3], u[7], u[11], u[15]);
#pragma unroll
    for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
            u[n1 + 4 * k2] = cmul (u[n1 + 4 * k2], w[n1 - 1]);
    const float2 w1 = make_float2 (kCosPi8, -kSinPi8); // W16^1
    const float2 w3 = make_float2 (kSinPi8, -kCosPi8); // W16^3
    u[5] = cmul (u[5], w1);
    u[6] = mul_w8_1 (u[6]);
    u[7] = cmul (u[7], w3);
    u[9] = mul_w8_1 (u[9]);
    u[10] = mul_mi (u[10]);
    u[11] = mul_w8_3 (u[11]);
    u[13] = cmul (u[13], w3);
    u[14] = mul_w8_3 (u[14]);
    {
        const float2 tt = cmul (u[15], w1);
        u[15] = make_float2 (-tt.x, -tt.y);
    }
    bfly4 (u[0], u[1], u[2], u[3]);
    bfly4 (u[4], u[5], u[6], u[7]);
    bfly4 (u[8], u[9], u[10], u[11]);
    bfly4 (u[12], u[13], u[14], u[15]);
    float2 tt;
    tt = u[1]; u[1] = u[4]; u[4] = tt;
    tt = u[2]; u[2] = u[8]; u[8] = tt;
    tt = u[3]; u[3] = u[12]; u[12] = tt;
    tt = u[6]; u[6] = u[9]; u[9] = tt;
    tt = u[7]; u[7] = u[13]; u[13] = tt;
    tt = u[11]; u[11] = u[14]; u[14] = tt;
}

// Stockham exchange after stage S - 1 (p = product of the radices so far: 32^S / 32 ... P_ = 1 after stage 0, 32 after
// stage 1), one component at a time through the plane.  Bank check (ds_*_b32: 32 banks per 32-lane group): the writes go
// to phys (j + b p) with bank (t + b) mod 32, the reads to phys (t + 1024 a) with bank (t + t / 32) mod 32 -- both
// conflict-free over 32 consecutive t.
template <class G, int P_>
__device__ __forceinline__ void sp_exchange (float2 (&v)[32], float* __restrict__ plane, int t)
{
    // Addresses written out as ONE base per side + compile-time offsets (so that they fold into the instructions'
    // immediate fields instead of costing a VGPR each):
    //   write  phys (j + b p):  p = 1:  j = 32 t          -> 33 t + b
    //                           p = 32: j = 32 (t - k) + k -> 33 (t - k) + k + 33 b        (k = t mod 32)
    //   read   phys (t + T c) = t + t / 32 + (T + T / 32) c
    const int k = t & (P_ - 1);
    float* __restrict__ wp = plane + ((P_ == 1) ? 33 * t : 33 * (t - k) + k);
    constexpr int ws = (P_ == 1) ? 1 : 33;
    const float* __restrict__ rp0 = plane + (t + (t >> 5));
    int r1 = t + (t >> 5) + 16 * G::RS; // ds offsets are 16 bits: a second base for the upper half, kept from being folded
    CFFT_LAUNDER_INDEX (r1);            // back into the first one (which costs one address VGPR per read instead)
    const float* __restrict__ rp1 = plane + r1;
    __syncthreads(); // whoever read the plane last (previous exchange, previous frame) is done
#pragma unroll
    for (int b = 0; b < 32; ++b)
        wp[b * ws] = v[b].x;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
    {
        v[a].x = rp0[a * G::RS];
        v[a + 16].x = rp1[a * G::RS];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 32; ++b)
        wp[b * ws] = v[b].y;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a)
    {
        v[a].y = rp0[a * G::RS];
        v[a + 16].y = rp1[a * G::RS];
    }
}

// v (v[c] = x[t + T c]) -> v[c] = X[t + T c]
template <class G>
__device__ __forceinline__ void sp_core (float2 (&v)[32], const float2* __restrict__ tw, float* __restrict__ plane, int t)
{
    float2 w[10];
    fft_reg32<false> (v, w);
    CFFT_PHASE_FENCE();
    sp_load_twiddles<G, 1> (w, tw, t);
    sp_exchange<G, 1> (v, plane, t);
    CFFT_PHASE_FENCE();
    fft_reg32<true> (v, w);
    CFFT_PHASE_FENCE();
    CFFT_LAUNDER_INDEX (t);
    if constexpr (G::LAST == 32)
    {
        sp_load_twiddles<G, 2> (w, tw, t);
        sp_exchange<G, 32> (v, plane, t);
        CFFT_PHASE_FENCE();
        fft_reg32<true> (v, w);
    }
    else if constexpr (G::LAST == 2)
    {
        // N = 2048: 1024 radix-2 butterflies, sixteen per work-item: slots q and q + 16 (elements i = t + 64 q and i + 1024),
        // twiddle W_N^i = W_N^t W_32^q on the second input: ONE table entry per work-item, the W_32^q are constants
        const float2 w0 = tw[G::tw_stage2 + t];
        sp_exchange<G, 32> (v, plane, t);
        CFFT_PHASE_FENCE();
#define CFFT_R2(Q) \
    { \
        const float2 a = v[Q], b = cmul (mul_w32<Q> (v[Q + 16]), w0); \
        v[Q] = make_float2 (a.x + b.x, a.y + b.y); \
        v[Q + 16] = make_float2 (a.x - b.x, a.y - b.y); \
    }
        CFFT_R2 (0) CFFT_R2 (1) CFFT_R2 (2) CFFT_R2 (3) CFFT_R2 (4) CFFT_R2 (5) CFFT_R2 (6) CFFT_R2 (7)
        CFFT_R2 (8) CFFT_R2 (9) CFFT_R2 (10) CFFT_R2 (11) CFFT_R2 (12) CFFT_R2 (13) CFFT_R2 (14) CFFT_R2 (15)
#undef CFFT_R2
    }
    else if constexpr (G::LAST == 8)
    {
        // N = 8192: 1024 radix-8 butterflies, four per work-item: butterfly q works on the slots c = q + 4 a (elements
        // t + 256 q + 1024 a) and leaves X[t + 256 q + 1024 b] in slot q + 4 b; its seven twiddles W_N^(a (t + 256 q)) are
        // requested before the exchange, two butterflies at a time (28 VGPRs)
        float2 w8[2][7];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 7; ++j)
                w8[q][j] = tw[G::tw_stage2 + j * 1024 + t + 256 * q];
        sp_exchange<G, 32> (v, plane, t);
        CFFT_PHASE_FENCE();
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            if (h == 1)
            {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 7; ++j)
                        w8[q][j] = tw[G::tw_stage2 + j * 1024 + t + 256 * (2 + q)];
            }
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
            {
                const int q = 2 * h + q2;
                float2 u[8];
                u[0] = v[q];
#pragma unroll
                for (int a = 1; a < 8; ++a)
                    u[a] = cmul (v[q + 4 * a], w8[q2][a - 1]);
                fft_reg<8> (u);
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    v[q + 4 * b] = u[b];
            }
            CFFT_PHASE_FENCE();
        }
    }
    else
    {
        // N = 16384: 1024 radix-16 butterflies, two per work-item: butterfly q works on the slots c = q + 2 a (elements
        // t + 512 q + 1024 a) and leaves X[t + 512 q + 1024 b] in slot q + 2 b
        float2 w16[2][6];
        sp_load_twiddles16<G> (w16, tw, t);
        sp_exchange<G, 32> (v, plane, t);
        CFFT_PHASE_FENCE();
#pragma unroll
        for (int q = 0; q < 2; ++q)
        {
            float2 u[16];
#pragma unroll
            for (int a = 0; a < 16; ++a)
                u[a] = v[q + 2 * a];
            fft_reg16_tw (u, w16[q]);
#pragma unroll
            for (int b = 0; b < 16; ++b)
                v[q + 2 * b] = u[b];
        }
    }
    CFFT_PHASE_FENCE();
}

// ---- complex ------------------------------------------------------------------------------------------------------
// CFFT_SPLIT_PREFETCH (round 4): the first PF of the NEXT frame's 32 operands per work-item are requested before this frame's
// transform and wait in registers (T = 1024, one workgroup per CU: its loads, cores and stores did not overlap at all; the
// 128-register budget of four waves per SIMD has room for 16 of the 32).  0 = round 3's form.
#ifndef CFFT_SPLIT_PREFETCH
#define CFFT_SPLIT_PREFETCH 1
#endif
template <class G, bool INV>
__global__ __launch_bounds__ (G::T) void c2c_split_kernel (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ tw,
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float sp_plane[];
    const int t0 = threadIdx.x;
    constexpr int PF = (CFFT_SPLIT_PREFETCH && G::T == 1024) ? 16 : 0;
    float2 nx[PF > 0 ? PF : 1];
    if constexpr (PF > 0)
    {
        if (blockIdx.x < batch)
        {
            const float2* __restrict__ src0 = in + (size_t) blockIdx.x * in_stride;
#pragma unroll
            for (int a = 0; a < PF; ++a)
                nx[a] = gload (&src0[(unsigned) (t0 + G::T * a)]);
        }
    }
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t); // keep the ~200 loop-invariant LDS / table offsets from being hoisted and held (stockham.hpp)
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        float2 v[32];
#pragma unroll
        for (int a = PF; a < 32; ++a)
        {
            const float2 x = gload (&src[(unsigned) (t + G::T * a)]);
            v[a] = INV ? make_float2 (x.y, x.x) : x;
        }
        if constexpr (PF > 0)
        {
#pragma unroll
            for (int a = 0; a < PF; ++a)
            {
                // (the swap at the consumer behind an opaque statement: built at the request it would wait for the prefetch there)
                float2 x = nx[a];
                asm volatile ("" : "+v"(x.x), "+v"(x.y));
                v[a] = INV ? make_float2 (x.y, x.x) : x;
            }
            const unsigned gn = g + gridDim.x;
            if (gn < batch)
            {
                const float2* __restrict__ srcn = in + (size_t) gn * in_stride;
#pragma unroll
                for (int a = 0; a < PF; ++a)
                    nx[a] = gload (&srcn[(unsigned) (t + G::T * a)]);
            }
        }
        sp_core<G> (v, tw, sp_plane, t);
        int ts = t0;
        CFFT_LAUNDER_INDEX (ts); // the store addresses are formed here, not at the top of the loop (where they would be spilled)
#pragma unroll
        for (int b = 0; b < 32; ++b)
            gstore (&dst[(unsigned) (ts + G::T * b)], INV ? make_float2 (v[b].y, v[b].x) : v[b]);
    }
}

// ---- real, N = 2 M (65536 or 32768 samples) -----------------------------------------------------------------------------
// Work-item t owns Z[t + T b], b < 32.  The pair stage handles k = t + T q, q < 16 (k < M/2) together with M - k,
// which is the partner work-item's slot 31 - q (work-item T - t; for t = 0: slot 32 - q of work-item 0 itself): the
// UPPER 16 slots of every work-item travel through the first half of the plane, one component at a time, at their natural
// index k' - M/2.
template <class G, bool TO_PARTNER>
__device__ __forceinline__ void sp_pair_exchange (float2 (&up)[16], float* __restrict__ plane, int t)
{
    // TO_PARTNER == false: `up` holds my own upper slots (k' = t + T (16 + q)); on return the values at M - k, k = t + T q
    // TO_PARTNER == true : `up` holds the values I computed for M - k; on return my own upper slots
    // Two address families, each ONE base + compile-time offsets (H = M/2 = 16 T, RS = T + T/32):
    //   own side     index k' - H = t + T q            -> phys = t + t/32 + RS q
    //   mirror side  index (M - k) - H = H - t - T q    -> phys = (-t) + floor(-t / 32) + 16 RS - RS q
    // The one irregular pair, k = 0 <-> M/2 (work-item 0, q = 0), lives at own-side index 0; the mirror formula gives H
    // for it, so work-item 0 keeps a copy there.
    float* __restrict__ own = plane + (t + (t >> 5));
    float* __restrict__ mir = plane + ((-t) + ((-t) >> 5) + 16 * G::RS);
    float* __restrict__ wr = TO_PARTNER ? mir : own;
    const float* __restrict__ rd = TO_PARTNER ? own : mir;
    constexpr int ws = TO_PARTNER ? -G::RS : G::RS, rs = -ws;
    float* __restrict__ patch = TO_PARTNER ? plane : plane + 16 * G::RS;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q)
        wr[q * ws] = up[q].x;
    if (t == 0)
        *patch = up[0].x;
    __syncthreads();
    float rx[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
        rx[q] = rd[q * rs];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q)
        wr[q * ws] = up[q].y;
    if (t == 0)
        *patch = up[0].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q)
        up[q] = make_float2 (rx[q], rd[q * rs]);
}

// forward: N real samples (as M complex points z[n] = x[2n] + i x[2n+1]) -> packed spectrum (real_kernels.hpp)
// LAYOUT 0: the packed spectrum.  LAYOUT 1 / 2 (round 3): the JUCE engine's interleaved layout written by the pair stage itself
// (reference chowdsp_fft_juce/chowdsp_fft_juce.cpp:57-66), as r2c_kernel<..., LAYOUT> does for the shorter frames: frames of 2 M
// complex slots, c[0] = (X0, 0), c[k] = X_k, c[M] = (X_{N/2}, 0), and for LAYOUT 2 also c[2 M - k] = conj X_k.  In place on the
// 2 N-float buffers: the whole frame is in registers before the first store.
template <class G, int LAYOUT = 0>
__global__ __launch_bounds__ (G::T) void r2c_split_kernel (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ tw,
                                                            const float2* __restrict__ wtab,
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride)
{
    extern __shared__ __attribute__ ((aligned (16))) float sp_plane[];
    constexpr int M = G::N;
    const int t0 = threadIdx.x;
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v[a] = gload (&src[(unsigned) (t + G::T * a)]);
        sp_core<G> (v, tw, sp_plane, t);
        int ts = t0;
        CFFT_LAUNDER_INDEX (ts);
        float2 up[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            up[q] = v[16 + q];
        sp_pair_exchange<G, false> (up, sp_plane, ts);
        CFFT_PHASE_FENCE();
        // four rounds of four pairs: the pair twiddles of a round (8 VGPRs) next to the 64 that hold Z[k] and Z[M - k]
#pragma unroll
        for (int h = 0; h < 4; ++h)
        {
            float2 pw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pw[q] = wtab[ts + G::T * (4 * h + q)];
#pragma unroll
            for (int q8 = 0; q8 < 4; ++q8)
            {
                const int q = 4 * h + q8;
                const int k = ts + G::T * q;
                const float2 A = v[q], Zp = up[q];
                float2 Xk, Xp;
                real_split_pair (A, Zp, pw[q8], Xk, Xp);
                if (k == 0)
                {
                    Xk = make_float2 (A.x + A.y, A.x - A.y); // [X0.re, X(N/2).re]
                    Xp = make_float2 (Zp.x, -Zp.y);          // X[M/2] = conj(Z[M/2])
                }
                const unsigned kp = (unsigned) (k == 0 ? M / 2 : M - k);
                if constexpr (LAYOUT == 0)
                {
                    gstore (&dst[(unsigned) k], Xk);
                    gstore (&dst[kp], Xp);
                }
                else
                {
                    gstore (&dst[(unsigned) k], k == 0 ? make_float2 (Xk.x, 0.f) : Xk);
                    gstore (&dst[kp], Xp);
                    if (k == 0)
                        gstore (&dst[(unsigned) M], make_float2 (Xk.y, 0.f));
                    if constexpr (LAYOUT == 2)
                    {
                        if (k != 0)
                            gstore (&dst[(unsigned) (2 * M - k)], make_float2 (Xk.x, -Xk.y));
                        gstore (&dst[(unsigned) (2 * M) - kp], make_float2 (Xp.x, -Xp.y));
                    }
                }
            }
            CFFT_PHASE_FENCE();
        }
    }
}

// backward: packed spectrum -> N real samples (unscaled).  LAYOUT 1 (round 3): the JUCE engine's layout in (X0 = c[0].re,
// X_{N/2} = c[M].re, reference chowdsp_fft_juce.cpp:74), the signal times `scale` out, in place on the 2 N-float buffers
template <class G, int LAYOUT = 0>
__global__ __launch_bounds__ (G::T) void c2r_split_kernel (const float2* __restrict__ in,
                                                            float2* __restrict__ out,
                                                            const float2* __restrict__ tw,
                                                            const float2* __restrict__ wtab,
                                                            unsigned batch,
                                                            unsigned in_stride,
                                                            unsigned out_stride,
                                                            float scale = 1.f)
{
    extern __shared__ __attribute__ ((aligned (16))) float sp_plane[];
    constexpr int M = G::N;
    const int t0 = threadIdx.x;
    // round 4 (CFFT_SPLIT_PREFETCH, T = 1024, packed layout): the first round's sixteen operands of the NEXT frame wait in registers
    constexpr bool PFC = CFFT_SPLIT_PREFETCH && G::T == 1024 && LAYOUT == 0;
    constexpr int PFQ = 6; // pairs of the first round that are prefetched (all eight: 20 bytes of scratch)
    float2 na[PFC ? PFQ : 1], nxp[PFC ? PFQ : 1];
    if constexpr (PFC)
    {
        if (blockIdx.x < batch)
        {
            const float2* __restrict__ src0 = in + (size_t) blockIdx.x * in_stride;
#pragma unroll
            for (int q8 = 0; q8 < PFQ; ++q8)
            {
                const int k = t0 + G::T * q8;
                na[q8] = gload (&src0[(unsigned) k]);
                nxp[q8] = gload (&src0[(unsigned) (k == 0 ? M / 2 : M - k)]);
            }
        }
    }
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * in_stride;
        float2* __restrict__ dst = out + (size_t) g * out_stride;
        float2 v[32], up[16];
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            // two rounds of eight pairs (48 VGPRs of operands each)
            float2 a[8], xp[8], pw[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8)
            {
                const int k = t + G::T * (8 * h + q8);
                if (PFC && h == 0 && q8 < PFQ)
                {
                    a[q8] = na[PFC ? q8 : 0];
                    xp[q8] = nxp[PFC ? q8 : 0];
                }
                else
                {
                    a[q8] = gload (&src[(unsigned) k]);
                    xp[q8] = gload (&src[(unsigned) (k == 0 ? M / 2 : M - k)]);
                }
                pw[q8] = wtab[k];
                if constexpr (LAYOUT != 0)
                {
                    if (k == 0)
                        a[q8].y = gload (&src[(unsigned) M]).x; // packed slot 0 = (X0.re, X_{N/2}.re)
                }
            }
            if constexpr (PFC)
            {
                if (h == 1 && g + gridDim.x < batch) // behind this frame's second round of loads
                {
                    const float2* __restrict__ srcn = in + (size_t) (g + gridDim.x) * in_stride;
#pragma unroll
                    for (int q8 = 0; q8 < PFQ; ++q8)
                    {
                        const int k = t + G::T * q8;
                        na[q8] = gload (&srcn[(unsigned) k]);
                        nxp[q8] = gload (&srcn[(unsigned) (k == 0 ? M / 2 : M - k)]);
                    }
                }
            }
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8)
            {
                const int q = 8 * h + q8;
                const int k = t + G::T * q;
                float2 zk, zp;
                real_merge_pair (a[q8], xp[q8], pw[q8], zk, zp);
                if (k == 0)
                {
                    zk = make_float2 (a[q8].x + a[q8].y, a[q8].x - a[q8].y); // Z[0] = (X0 + XM) + i (X0 - XM)
                    zp = make_float2 (2.f * xp[q8].x, -2.f * xp[q8].y);      // Z[M/2] = 2 conj(X[M/2])
                }
                // re <-> im swapped: the inverse DFT is swap(DFT(swap(.)))
                v[q] = make_float2 (zk.y, zk.x);
                up[q] = make_float2 (zp.y, zp.x);
            }
        }
        sp_pair_exchange<G, true> (up, sp_plane, t);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[16 + q] = up[q];
        sp_core<G> (v, tw, sp_plane, t);
        int ts = t0;
        CFFT_LAUNDER_INDEX (ts);
#pragma unroll
        for (int b = 0; b < 32; ++b)
            gstore (&dst[(unsigned) (ts + G::T * b)], LAYOUT != 0 ? make_float2 (v[b].y * scale, v[b].x * scale) : make_float2 (v[b].y, v[b].x));
    }
}
// ---- fused convolution frames: y = BACKWARD (FORWARD (x) * H) * scaling in ONE pass -----------------------------------------------
// (the call sequence of test/test.cpp:161-166 per frame; conv_kernels.hpp holds the LDS-resident plans' form).  The frame
// never leaves the registers between the two cores: the spectrum bins of work-item t are slots t + T b, exactly where the
// inverse core wants its input.  The filter bins are the same for every frame but 64 VGPRs of them do not fit next to the
// frame: they are re-read from L2 per frame, in rounds, requested before the multiplications that use them.
template <class G>
__global__ __launch_bounds__ (G::T, 4) void conv_frames_c2c_split_kernel (const float2* __restrict__ in,
                                                                        const float2* __restrict__ H,
                                                                        float2* __restrict__ out,
                                                                        const float2* __restrict__ tw,
                                                                        float scaling,
                                                                        unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float sp_plane[];
    const int t0 = threadIdx.x;
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * G::N;
        float2* __restrict__ dst = out + (size_t) g * G::N;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v[a] = gload (&src[(unsigned) (t + G::T * a)]);
        sp_core<G> (v, tw, sp_plane, t);
        int tm = t0;
        CFFT_LAUNDER_INDEX (tm);
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            float2 h[8];
#pragma unroll
            for (int b = 0; b < 8; ++b)
                h[b] = H[tm + G::T * (8 * r + b)];
#pragma unroll
            for (int b = 0; b < 8; ++b)
            {
                const float2 y = cmul (v[8 * r + b], h[b]);
                v[8 * r + b] = make_float2 (y.y * scaling, y.x * scaling); // re <-> im: the inverse DFT is swap (DFT (swap (.)))
            }
            CFFT_PHASE_FENCE();
        }
        sp_core<G> (v, tw, sp_plane, tm);
        int ts = t0;
        CFFT_LAUNDER_INDEX (ts);
#pragma unroll
        for (int b = 0; b < 32; ++b)
            gstore (&dst[(unsigned) (ts + G::T * b)], make_float2 (v[b].y, v[b].x));
    }
}

// REAL frames of N = 2 M samples, H a packed real spectrum (M float2, element 0 = DC / Nyquist): forward core, the pair
// exchange of r2c_split_kernel, split -> * H -> merge on every (k, M - k) pair in registers, the pair exchange of
// c2r_split_kernel, inverse core.
template <class G>
__global__ __launch_bounds__ (G::T, 4) void conv_frames_real_split_kernel (const float2* __restrict__ in,
                                                                         const float2* __restrict__ H,
                                                                         float2* __restrict__ out,
                                                                         const float2* __restrict__ tw,
                                                                         const float2* __restrict__ wtab,
                                                                         float scaling,
                                                                         unsigned batch)
{
    extern __shared__ __attribute__ ((aligned (16))) float sp_plane[];
    constexpr int M = G::N;
    const int t0 = threadIdx.x;
    for (unsigned g = blockIdx.x; g < batch; g += gridDim.x)
    {
        int t = t0;
        CFFT_LAUNDER_INDEX (t);
        const float2* __restrict__ src = in + (size_t) g * M;
        float2* __restrict__ dst = out + (size_t) g * M;
        float2 v[32];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            v[a] = gload (&src[(unsigned) (t + G::T * a)]);
        sp_core<G> (v, tw, sp_plane, t);
        int tm = t0;
        CFFT_LAUNDER_INDEX (tm);
        float2 up[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            up[q] = v[16 + q];
        sp_pair_exchange<G, false> (up, sp_plane, tm);
        CFFT_PHASE_FENCE();
        constexpr int PR = (G::T == 64) ? 1 : 2;
        // rounds of PR pairs: pair twiddle + the two filter bins of a pair are 6 VGPRs next to the 96 of v / up
#pragma unroll
        for (int r = 0; r < 16 / PR; ++r)
        {
            float2 pw[PR], hk[PR], hp[PR];
#pragma unroll
            for (int j = 0; j < PR; ++j)
            {
                const int k = tm + G::T * (PR * r + j);
                pw[j] = wtab[k];
                hk[j] = H[k];
                hp[j] = H[k == 0 ? M / 2 : M - k];
            }
#pragma unroll
            for (int j = 0; j < PR; ++j)
            {
                const int q = PR * r + j;
                const int k = tm + G::T * q;
                const float2 A = v[q], Zp = up[q];
                float2 Xk, Xp;
                real_split_pair (A, Zp, pw[j], Xk, Xp);
                float2 Yk = cmul (Xk, hk[j]), Yp = cmul (Xp, hp[j]);
                if (k == 0)
                {
                    // packed DC / Nyquist: real products (simd/chowdsp_fft_impl_sse.cpp:1640-1644); X[M/2] = conj (Z[M/2])
                    Yk = make_float2 ((A.x + A.y) * hk[j].x, (A.x - A.y) * hk[j].y);
                    Yp = cmul (make_float2 (Zp.x, -Zp.y), hp[j]);
                }
                Yk = make_float2 (Yk.x * scaling, Yk.y * scaling);
                Yp = make_float2 (Yp.x * scaling, Yp.y * scaling);
                float2 zk, zp;
                real_merge_pair (Yk, Yp, pw[j], zk, zp);
                if (k == 0)
                {
                    zk = make_float2 (Yk.x + Yk.y, Yk.x - Yk.y);
                    zp = make_float2 (2.f * Yp.x, -2.f * Yp.y);
                }
                v[q] = make_float2 (zk.y, zk.x);
                up[q] = make_float2 (zp.y, zp.x);
            }
            CFFT_PHASE_FENCE();
        }
        sp_pair_exchange<G, true> (up, sp_plane, tm);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[16 + q] = up[q];
        sp_core<G> (v, tw, sp_plane, tm);
        int ts = t0;
        CFFT_LAUNDER_INDEX (ts);
#pragma unroll
        for (int b = 0; b < 32; ++b)
            gstore (&dst[(unsigned) (ts + G::T * b)], make_float2 (v[b].y, v[b].x));
    }
}
} // namespace cfft
