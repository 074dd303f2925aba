/*
 * oracle/pffft_restate.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar restatement of the algorithm the reference runs on its SSE path (V = 4 lanes),
 * used only as the parity checker for the HIP kernels (tests/, __graft_entry__.smoke(), and the
 * "port" CPU baseline of bench.py).  The product library never links, loads or calls this file.
 *
 * What is restated and where it comes from (paths relative to /root/reference):
 *   plan / factorisation / twiddle tables   simd/chowdsp_fft_impl_common.hpp:51-228
 *   complex radix-2/4 passes + driver       simd/chowdsp_fft_impl_sse.cpp:148-180, 221-296, 372-432
 *   complex cross-lane finalize/preprocess  simd/chowdsp_fft_impl_sse.cpp:434-549
 *   real forward radix-2/4 passes + driver  simd/chowdsp_fft_impl_sse.cpp:552-586, 627-702, 788-846
 *   real cross-lane finalize / preprocess   simd/chowdsp_fft_impl_sse.cpp:849-1078
 *   real backward radix-2/4 passes + driver simd/chowdsp_fft_impl_sse.cpp:1081-1124, 1172-1260, 1358-1416
 *   z-domain reorder                        simd/chowdsp_fft_impl_sse.cpp:1420-1515
 *   transform driver                        simd/chowdsp_fft_impl_sse.cpp:1518-1601
 *   multiply-accumulate, accumulate         simd/chowdsp_fft_impl_sse.cpp:1603-1660
 *   radix-3 / radix-5 passes (sizes 2^a 3^b 5^c) simd/chowdsp_fft_impl_sse.cpp:182-219, 298-370 (complex),
 *                                            :588-625, :704-786 (real forward), :1126-1170, :1262-1356 (real backward)
 * The restatement is faithful to the point of reproducing a defect of the reference: its REAL FORWARD transform
 * is wrong for every size divisible by 25 (radf5_ps with ido > 1, :747-782) -- see tests/test_gpu_other_sizes.py.
 *
 * How it is restated.  The reference treats a length-N transform as V = 4 interleaved transforms of
 * length N/4: element e of lane l lives at float index 4*e + l of a buffer, every butterfly pass is
 * the scalar FFTPACK pass applied to all four lanes at once with SSE arithmetic, and only the
 * finalize / preprocess / reorder steps mix lanes.  Here each pass is written ONCE as a scalar routine
 * over a strided view (stride 4 floats) and simply called for l = 0..3; the lane-mixing steps are
 * written as explicit loops over (vector, lane).  The order of floating point operations of the SSE
 * code (no FMA: separate multiply and add, `cplx_mul` = mul, mul, sub, mul, add) is kept, and the file
 * is compiled with -ffp-contract=off, so results are expected to be bit-identical to the reference's
 * SSE path; tests/test_oracle_vs_reference.py pins exactly that against oracle/_ref.
 *
 * PARITY PINNED: against the reference compiled in place (oracle/_ref/libchowdsp_fft_ref.so) and against
 * the golden vectors under tests/golden/ generated from that build by tests/golden/make_golden.py.
 */
#include "pffft_restate.h"

#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_SQRT2
#define M_SQRT2 1.41421356237309504880
#endif

#define LANES 4
#define AT(p, e) ((p)[LANES * (e)]) /* element e of the lane view p */

struct oracle_setup
{
    int N;
    int transform;
    int Ncvec;       /* complex 4-vectors per frame: N/4 (complex) or N/8 (real) */
    int nfac;        /* number of passes */
    int fac[15];     /* radix of each pass, in FFTPACK order */
    float* e;        /* cross-lane twiddles, 6*Ncvec floats */
    float* twiddle;  /* per-pass twiddles */
};

/* ---- plan (common.hpp:51-82 decompose, :84-160 rffti1_ps / cffti1_ps, :162-228 fft_new_setup) ---- */

/* common.hpp:51-82 decompose: divide out the candidate radices in the given order (complex {5,3,4,2},
   real {4,2,3,5}); a factor 2 found after other factors is moved to the front.  Returns the number of
   factors, or -1 when n has a prime factor other than 2, 3, 5 (common.hpp:216-225 rejects the size). */
static int factor_like_reference (int n, int* fac, const int* tries)
{
    int nf = 0, left = n;
    for (int t = 0; tries[t] != 0; ++t)
    {
        const int r = tries[t];
        while (left != 1 && left % r == 0)
        {
            if (nf >= 15)
                return -1;
            fac[nf++] = r;
            left /= r;
            if (r == 2 && nf != 1)
            {
                for (int i = nf - 1; i > 0; --i)
                    fac[i] = fac[i - 1];
                fac[0] = 2;
            }
        }
    }
    return left == 1 ? nf : -1;
}

static void fill_complex_pass_twiddles (int n, float* wa, const int* fac, int nf)
{
    const float argh = (float) (2 * M_PI) / (float) n;
    int i = 1, l1 = 1;
    for (int k1 = 0; k1 < nf; ++k1)
    {
        const int ip = fac[k1];
        const int l2 = l1 * ip;
        const int ido = n / l2;
        int ld = 0;
        for (int j = 1; j < ip; ++j)
        {
            wa[i - 1] = 1.0f;
            wa[i] = 0.0f;
            ld += l1;
            const float argld = (float) ld * argh;
            for (int fi = 1; fi <= ido; ++fi)
            {
                i += 2;
                wa[i - 1] = cosf ((float) fi * argld);
                wa[i] = sinf ((float) fi * argld);
            }
        }
        l1 = l2;
    }
}

static void fill_real_pass_twiddles (int n, float* wa, const int* fac, int nf)
{
    const float argh = (float) (2 * M_PI) / (float) n;
    int is = 0, l1 = 1;
    for (int k1 = 0; k1 + 1 < nf; ++k1)
    {
        const int ip = fac[k1];
        const int l2 = l1 * ip;
        const int ido = n / l2;
        int ld = 0;
        for (int j = 1; j < ip; ++j)
        {
            int i = is, fi = 0;
            ld += l1;
            const float argld = (float) ld * argh;
            for (int ii = 3; ii <= ido; ii += 2)
            {
                i += 2;
                fi += 1;
                wa[i - 2] = cosf ((float) fi * argld);
                wa[i - 1] = sinf ((float) fi * argld);
            }
            is += ido;
        }
        l1 = l2;
    }
}

oracle_setup* oracle_new_setup (int N, int transform)
{
    static const int tries_complex[] = { 5, 3, 4, 2, 0 }; /* common.hpp:122 */
    static const int tries_real[] = { 4, 2, 3, 5, 0 };    /* common.hpp:86 */
    if (N <= 0)
        return NULL;
    if (transform == ORACLE_REAL && N % (2 * LANES * LANES) != 0)
        return NULL; /* common.hpp:168-172 */
    if (transform == ORACLE_COMPLEX && N % (LANES * LANES) != 0)
        return NULL; /* common.hpp:173-177 */

    oracle_setup* s = (oracle_setup*) calloc (1, sizeof (oracle_setup));
    if (! s)
        return NULL;
    s->N = N;
    s->transform = transform;
    s->Ncvec = (transform == ORACLE_REAL ? N / 2 : N) / LANES;
    s->e = (float*) calloc ((size_t) 6 * s->Ncvec + 8, sizeof (float));
    s->twiddle = (float*) calloc ((size_t) 2 * s->Ncvec + 16, sizeof (float));
    s->nfac = factor_like_reference (N / LANES, s->fac, transform == ORACLE_REAL ? tries_real : tries_complex);
    if (! s->e || ! s->twiddle || s->nfac < 0)
    {
        oracle_destroy_setup (s);
        return NULL;
    }
    /* cross-lane twiddles, common.hpp:195-205: block i = k/4 holds, for m = 0..2, a cos-vector and a
       sin-vector whose lane j = k%4 is the angle -2*pi*(m+1)*k/N (angle rounded to float first) */
    for (int k = 0; k < s->Ncvec; ++k)
    {
        const int blk = k / LANES, lane = k % LANES;
        for (int m = 0; m < LANES - 1; ++m)
        {
            const float A = (float) (-2 * M_PI * (m + 1) * k / N);
            s->e[(2 * (blk * 3 + m) + 0) * LANES + lane] = cosf (A);
            s->e[(2 * (blk * 3 + m) + 1) * LANES + lane] = sinf (A);
        }
    }
    if (transform == ORACLE_REAL)
        fill_real_pass_twiddles (N / LANES, s->twiddle, s->fac, s->nfac);
    else
        fill_complex_pass_twiddles (N / LANES, s->twiddle, s->fac, s->nfac);
    return s;
}

void oracle_destroy_setup (oracle_setup* s)
{
    if (! s)
        return;
    free (s->e);
    free (s->twiddle);
    free (s);
}

int oracle_num_passes (const oracle_setup* s) { return s->nfac; }
int oracle_pass_radix (const oracle_setup* s, int pass) { return s->fac[pass]; }

/* ---- complex arithmetic in the SSE rounding order (sse.cpp:102-140) --------------------------- */
#define CMUL(ar, ai, br, bi)                     \
    do                                           \
    {                                            \
        const float t_ = (ar) * (bi);            \
        float r_ = (ar) * (br);                  \
        r_ = r_ - (ai) * (bi);                   \
        float i_ = (ai) * (br);                  \
        i_ = i_ + t_;                            \
        (ar) = r_;                               \
        (ai) = i_;                               \
    } while (0)
#define CMUL_CONJ(ar, ai, br, bi)                \
    do                                           \
    {                                            \
        const float t_ = (ar) * (bi);            \
        float r_ = (ar) * (br);                  \
        r_ = r_ + (ai) * (bi);                   \
        float i_ = (ai) * (br);                  \
        i_ = i_ - t_;                            \
        (ar) = r_;                               \
        (ai) = i_;                               \
    } while (0)

/* ---- complex passes on ONE lane (sse.cpp:148-180 passf2_ps, :221-296 passf4_ps) -------------------
   `ido` is the reference's idot = 2 * (complex butterflies per block): re at even, im at odd positions. */
static void cpass2_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, float fsign)
{
    const int span = l1 * ido;
    for (int blk = 0; blk < l1; ++blk)
    {
        const float* c = cc + LANES * (2 * ido * blk);
        float* h = ch + LANES * (ido * blk);
        if (ido <= 2)
        {
            AT (h, 0) = AT (c, 0) + AT (c, ido);
            AT (h, span) = AT (c, 0) - AT (c, ido);
            AT (h, 1) = AT (c, 1) + AT (c, ido + 1);
            AT (h, span + 1) = AT (c, 1) - AT (c, ido + 1);
            continue;
        }
        for (int i = 0; i < ido - 1; i += 2)
        {
            float dr = AT (c, i) - AT (c, i + ido);
            float di = AT (c, i + 1) - AT (c, i + ido + 1);
            const float wr = wa1[i], wi = fsign * wa1[i + 1];
            AT (h, i) = AT (c, i) + AT (c, i + ido);
            AT (h, i + 1) = AT (c, i + 1) + AT (c, i + ido + 1);
            CMUL (dr, di, wr, wi);
            AT (h, i + span) = dr;
            AT (h, i + span + 1) = di;
        }
    }
}

static void cpass4_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, const float* wa3, float fsign)
{
    const int span = l1 * ido;
    for (int blk = 0; blk < l1; ++blk)
    {
        const float* c = cc + LANES * (4 * ido * blk);
        float* h = ch + LANES * (ido * blk);
        for (int i = 0; i < ido - 1; i += 2)
        {
            const float tr1 = AT (c, i) - AT (c, i + 2 * ido);
            const float tr2 = AT (c, i) + AT (c, i + 2 * ido);
            const float ti1 = AT (c, i + 1) - AT (c, i + 2 * ido + 1);
            const float ti2 = AT (c, i + 1) + AT (c, i + 2 * ido + 1);
            const float tr4 = (AT (c, i + 3 * ido + 1) - AT (c, i + ido + 1)) * fsign;
            const float ti4 = (AT (c, i + ido) - AT (c, i + 3 * ido)) * fsign;
            const float tr3 = AT (c, i + ido) + AT (c, i + 3 * ido);
            const float ti3 = AT (c, i + ido + 1) + AT (c, i + 3 * ido + 1);
            float cr2, ci2, cr3, ci3, cr4, ci4;

            AT (h, i) = tr2 + tr3;
            AT (h, i + 1) = ti2 + ti3;
            cr3 = tr2 - tr3;
            ci3 = ti2 - ti3;
            cr2 = tr1 + tr4;
            cr4 = tr1 - tr4;
            ci2 = ti1 + ti4;
            ci4 = ti1 - ti4;
            if (ido != 2) /* the ido == 2 special case of the reference (:228-250) has no twiddles */
            {
                CMUL (cr2, ci2, wa1[i], fsign * wa1[i + 1]);
                CMUL (cr3, ci3, wa2[i], fsign * wa2[i + 1]);
                CMUL (cr4, ci4, wa3[i], fsign * wa3[i + 1]);
            }
            AT (h, i + span) = cr2;
            AT (h, i + span + 1) = ci2;
            AT (h, i + 2 * span) = cr3;
            AT (h, i + 2 * span + 1) = ci3;
            AT (h, i + 3 * span) = cr4;
            AT (h, i + 3 * span + 1) = ci4;
        }
    }
}

/* sse.cpp:182-219 passf3_ps (only reached for sizes with a factor 3) */
static void cpass3_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, float fsign)
{
    const float taur = -0.5f;
    const float taui = 0.866025403784439f * fsign;
    const int span = l1 * ido;
    for (int blk = 0; blk < l1; ++blk)
    {
        const float* c = cc + LANES * (3 * ido * blk);
        float* h = ch + LANES * (ido * blk);
        for (int i = 0; i < ido - 1; i += 2)
        {
            const float tr2 = AT (c, i + ido) + AT (c, i + 2 * ido);
            const float cr2 = AT (c, i) + tr2 * taur;
            AT (h, i) = AT (c, i) + tr2;
            const float ti2 = AT (c, i + ido + 1) + AT (c, i + 2 * ido + 1);
            const float ci2 = AT (c, i + 1) + ti2 * taur;
            AT (h, i + 1) = AT (c, i + 1) + ti2;
            const float cr3 = (AT (c, i + ido) - AT (c, i + 2 * ido)) * taui;
            const float ci3 = (AT (c, i + ido + 1) - AT (c, i + 2 * ido + 1)) * taui;
            float dr2 = cr2 - ci3, dr3 = cr2 + ci3;
            float di2 = ci2 + cr3, di3 = ci2 - cr3;
            CMUL (dr2, di2, wa1[i], fsign * wa1[i + 1]);
            AT (h, i + span) = dr2;
            AT (h, i + span + 1) = di2;
            CMUL (dr3, di3, wa2[i], fsign * wa2[i + 1]);
            AT (h, i + 2 * span) = dr3;
            AT (h, i + 2 * span + 1) = di3;
        }
    }
}

/* sse.cpp:298-370 passf5_ps (only reached for sizes with a factor 5) */
static void cpass5_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, const float* wa3, const float* wa4, float fsign)
{
    const float tr11 = .309016994374947f, tr12 = -.809016994374947f;
    const float ti11 = .951056516295154f * fsign, ti12 = .587785252292473f * fsign;
    const int span = l1 * ido;
    for (int blk = 0; blk < l1; ++blk)
    {
        const float* c = cc + LANES * (5 * ido * blk);
        float* h = ch + LANES * (ido * blk);
        for (int i = 0; i < ido - 1; i += 2)
        {
#define RE(j) AT (c, (j) * ido + i)
#define IM(j) AT (c, (j) * ido + i + 1)
            const float ti5 = IM (1) - IM (4), ti2 = IM (1) + IM (4);
            const float ti4 = IM (2) - IM (3), ti3 = IM (2) + IM (3);
            const float tr5 = RE (1) - RE (4), tr2 = RE (1) + RE (4);
            const float tr4 = RE (2) - RE (3), tr3 = RE (2) + RE (3);
            AT (h, i) = RE (0) + (tr2 + tr3);
            AT (h, i + 1) = IM (0) + (ti2 + ti3);
            const float cr2 = RE (0) + (tr2 * tr11 + tr3 * tr12);
            const float ci2 = IM (0) + (ti2 * tr11 + ti3 * tr12);
            const float cr3 = RE (0) + (tr2 * tr12 + tr3 * tr11);
            const float ci3 = IM (0) + (ti2 * tr12 + ti3 * tr11);
            const float cr5 = tr5 * ti11 + tr4 * ti12;
            const float ci5 = ti5 * ti11 + ti4 * ti12;
            const float cr4 = tr5 * ti12 - tr4 * ti11;
            const float ci4 = ti5 * ti12 - ti4 * ti11;
#undef RE
#undef IM
            float dr3 = cr3 - ci4, dr4 = cr3 + ci4;
            float di3 = ci3 + cr4, di4 = ci3 - cr4;
            float dr5 = cr2 + ci5, dr2 = cr2 - ci5;
            float di5 = ci2 - cr5, di2 = ci2 + cr5;
            CMUL (dr2, di2, wa1[i], fsign * wa1[i + 1]);
            AT (h, i + span) = dr2;
            AT (h, i + span + 1) = di2;
            CMUL (dr3, di3, wa2[i], fsign * wa2[i + 1]);
            AT (h, i + 2 * span) = dr3;
            AT (h, i + 2 * span + 1) = di3;
            CMUL (dr4, di4, wa3[i], fsign * wa3[i + 1]);
            AT (h, i + 3 * span) = dr4;
            AT (h, i + 3 * span + 1) = di4;
            CMUL (dr5, di5, wa4[i], fsign * wa4[i + 1]);
            AT (h, i + 4 * span) = dr5;
            AT (h, i + 4 * span + 1) = di5;
        }
    }
}

/* pass driver, sse.cpp:372-432 (cfftf1_ps).  n = Ncvec.  Returns the buffer holding the result. */
static float* complex_passes (const oracle_setup* s, float* a, float* b, float fsign)
{
    const int n = s->Ncvec;
    float *in = a, *out = b;
    int l1 = 1, iw = 0;
    for (int p = 0; p < s->nfac; ++p)
    {
        const int ip = s->fac[p];
        const int l2 = ip * l1;
        const int ido = n / l2;
        const int idot = ido + ido;
        const float* wa = s->twiddle;
        for (int lane = 0; lane < LANES; ++lane)
        {
            if (ip == 4)
                cpass4_lane (idot, l1, in + lane, out + lane, wa + iw, wa + iw + idot, wa + iw + 2 * idot, fsign);
            else if (ip == 3)
                cpass3_lane (idot, l1, in + lane, out + lane, wa + iw, wa + iw + idot, fsign);
            else if (ip == 5)
                cpass5_lane (idot, l1, in + lane, out + lane, wa + iw, wa + iw + idot, wa + iw + 2 * idot, wa + iw + 3 * idot, fsign);
            else
                cpass2_lane (idot, l1, in + lane, out + lane, wa + iw, fsign);
        }
        l1 = l2;
        iw += (ip - 1) * idot;
        float* t = in;
        in = out;
        out = t;
    }
    return in;
}

/* ---- complex cross-lane step (sse.cpp:434-496 pffft_cplx_finalize, :498-549 pffft_cplx_preprocess) --
   A block is 8 vectors r0 i0 r1 i1 r2 i2 r3 i3.  The reference transposes the 4x4 matrices of re and
   of im, so that output lane l sees the four values that input VECTOR l held in its lanes 0..3. */
static void complex_finalize (int Ncvec, const float* in, float* out, const float* e)
{
    for (int blk = 0; blk < Ncvec / LANES; ++blk)
    {
        const float* I = in + 8 * LANES * blk;
        float* O = out + 8 * LANES * blk;
        const float* E = e + 6 * LANES * blk;
        for (int l = 0; l < LANES; ++l)
        {
            float r[4], im[4];
            for (int j = 0; j < 4; ++j)
            {
                r[j] = I[(2 * l) * LANES + j];      /* transposed: row j, column l */
                im[j] = I[(2 * l + 1) * LANES + j];
            }
            for (int j = 1; j < 4; ++j)
                CMUL (r[j], im[j], E[(2 * (j - 1)) * LANES + l], E[(2 * (j - 1) + 1) * LANES + l]);
            const float sr0 = r[0] + r[2], dr0 = r[0] - r[2];
            const float sr1 = r[1] + r[3], dr1 = r[1] - r[3];
            const float si0 = im[0] + im[2], di0 = im[0] - im[2];
            const float si1 = im[1] + im[3], di1 = im[1] - im[3];
            O[0 * LANES + l] = sr0 + sr1;
            O[1 * LANES + l] = si0 + si1;
            O[2 * LANES + l] = dr0 + di1;
            O[3 * LANES + l] = di0 - dr1;
            O[4 * LANES + l] = sr0 - sr1;
            O[5 * LANES + l] = si0 - si1;
            O[6 * LANES + l] = dr0 - di1;
            O[7 * LANES + l] = di0 + dr1;
        }
    }
}

static void complex_preprocess (int Ncvec, const float* in, float* out, const float* e)
{
    for (int blk = 0; blk < Ncvec / LANES; ++blk)
    {
        const float* I = in + 8 * LANES * blk;
        float* O = out + 8 * LANES * blk;
        const float* E = e + 6 * LANES * blk;
        for (int l = 0; l < LANES; ++l)
        {
            float r[4], im[4];
            for (int j = 0; j < 4; ++j)
            {
                r[j] = I[(2 * j) * LANES + l];
                im[j] = I[(2 * j + 1) * LANES + l];
            }
            const float sr0 = r[0] + r[2], dr0 = r[0] - r[2];
            const float sr1 = r[1] + r[3], dr1 = r[1] - r[3];
            const float si0 = im[0] + im[2], di0 = im[0] - im[2];
            const float si1 = im[1] + im[3], di1 = im[1] - im[3];
            r[0] = sr0 + sr1;
            im[0] = si0 + si1;
            r[1] = dr0 - di1;
            im[1] = di0 + dr1;
            r[2] = sr0 - sr1;
            im[2] = si0 - si1;
            r[3] = dr0 + di1;
            im[3] = di0 - dr1;
            for (int j = 1; j < 4; ++j)
                CMUL_CONJ (r[j], im[j], E[(2 * (j - 1)) * LANES + l], E[(2 * (j - 1) + 1) * LANES + l]);
            /* transpose on the way out: value j of lane l goes to vector l, lane j */
            for (int j = 0; j < 4; ++j)
            {
                O[(2 * l) * LANES + j] = r[j];
                O[(2 * l + 1) * LANES + j] = im[j];
            }
        }
    }
}

/* ---- real forward passes on ONE lane (sse.cpp:552-586 radf2_ps, :627-702 radf4_ps) ---------------- */
static void rfwd2_lane (int ido, int l1, const float* cc, float* ch, const float* wa1)
{
    const int span = l1 * ido;
    for (int k = 0; k < span; k += ido)
    {
        const float a = AT (cc, k), b = AT (cc, k + span);
        AT (ch, 2 * k) = a + b;
        AT (ch, 2 * (k + ido) - 1) = a - b;
    }
    if (ido < 2)
        return;
    if (ido != 2)
    {
        for (int k = 0; k < span; k += ido)
            for (int i = 2; i < ido; i += 2)
            {
                float tr2 = AT (cc, i - 1 + k + span), ti2 = AT (cc, i + k + span);
                const float br = AT (cc, i - 1 + k), bi = AT (cc, i + k);
                CMUL_CONJ (tr2, ti2, wa1[i - 2], wa1[i - 1]);
                AT (ch, i + 2 * k) = bi + ti2;
                AT (ch, 2 * (k + ido) - i) = ti2 - bi;
                AT (ch, i - 1 + 2 * k) = br + tr2;
                AT (ch, 2 * (k + ido) - i - 1) = br - tr2;
            }
        if (ido % 2 == 1)
            return;
    }
    for (int k = 0; k < span; k += ido)
    {
        AT (ch, 2 * k + ido) = -AT (cc, ido - 1 + k + span);
        AT (ch, 2 * k + ido - 1) = AT (cc, k + ido - 1);
    }
}

static void rfwd4_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, const float* wa3)
{
    const float minus_hsqt2 = -0.7071067811865475f;
    const int span = l1 * ido;
    for (int blk = 0; blk < l1; ++blk)
    {
        const float* c = cc + LANES * (ido * blk);
        float* h = ch + LANES * (4 * ido * blk);
        const float a0 = AT (c, 0), a1 = AT (c, span), a2 = AT (c, 2 * span), a3 = AT (c, 3 * span);
        const float tr1 = a1 + a3, tr2 = a0 + a2;
        AT (h, 2 * ido - 1) = a0 - a2;
        AT (h, 2 * ido) = a3 - a1;
        AT (h, 0) = tr1 + tr2;
        AT (h, 4 * ido - 1) = tr2 - tr1;
    }
    if (ido < 2)
        return;
    if (ido != 2)
    {
        for (int k = 0; k < span; k += ido)
            for (int i = 2; i < ido; i += 2)
            {
                const int ic = ido - i;
                const int p = i - 1 + k; /* the reference's running pointer pc = cc + 1 + k, advanced by 2 */
                float cr2 = AT (cc, p + span), ci2 = AT (cc, p + span + 1);
                float cr3 = AT (cc, p + 2 * span), ci3 = AT (cc, p + 2 * span + 1);
                float cr4 = AT (cc, p + 3 * span), ci4 = AT (cc, p + 3 * span + 1);
                CMUL_CONJ (cr2, ci2, wa1[i - 2], wa1[i - 1]);
                CMUL_CONJ (cr3, ci3, wa2[i - 2], wa2[i - 1]);
                CMUL_CONJ (cr4, ci4, wa3[i - 2], wa3[i - 1]);
                const float tr1 = cr2 + cr4, tr4 = cr4 - cr2;
                const float tr2 = AT (cc, p) + cr3, tr3 = AT (cc, p) - cr3;
                AT (ch, i - 1 + 4 * k) = tr1 + tr2;
                AT (ch, ic - 1 + 4 * k + 3 * ido) = tr2 - tr1;
                const float ti1 = ci2 + ci4, ti4 = ci2 - ci4;
                AT (ch, i - 1 + 4 * k + 2 * ido) = ti4 + tr3;
                AT (ch, ic - 1 + 4 * k + 1 * ido) = tr3 - ti4;
                const float ti2 = AT (cc, p + 1) + ci3, ti3 = AT (cc, p + 1) - ci3;
                AT (ch, i + 4 * k) = ti1 + ti2;
                AT (ch, ic + 4 * k + 3 * ido) = ti1 - ti2;
                AT (ch, i + 4 * k + 2 * ido) = tr4 + ti3;
                AT (ch, ic + 4 * k + 1 * ido) = tr4 - ti3;
            }
        if (ido % 2 == 1)
            return;
    }
    for (int k = 0; k < span; k += ido)
    {
        const float a = AT (cc, ido - 1 + k + span), b = AT (cc, ido - 1 + k + 3 * span);
        const float c = AT (cc, ido - 1 + k), d = AT (cc, ido - 1 + k + 2 * span);
        const float ti1 = (a + b) * minus_hsqt2;
        const float tr1 = (b - a) * minus_hsqt2;
        AT (ch, ido - 1 + 4 * k) = tr1 + c;
        AT (ch, ido - 1 + 4 * k + 2 * ido) = c - tr1;
        AT (ch, 4 * k + 1 * ido) = ti1 - d;
        AT (ch, 4 * k + 3 * ido) = ti1 + d;
    }
}

/* sse.cpp:588-625 radf3_ps (sizes with a factor 3) */
static void rfwd3_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2)
{
    const float taur = -0.5f, taui = 0.866025403784439f;
    for (int k = 0; k < l1; ++k)
    {
        const float cr2 = AT (cc, (k + l1) * ido) + AT (cc, (k + 2 * l1) * ido);
        AT (ch, 3 * k * ido) = AT (cc, k * ido) + cr2;
        AT (ch, (3 * k + 2) * ido) = (AT (cc, (k + l1 * 2) * ido) - AT (cc, (k + l1) * ido)) * taui;
        AT (ch, ido - 1 + (3 * k + 1) * ido) = AT (cc, k * ido) + cr2 * taur;
    }
    if (ido == 1)
        return;
    for (int k = 0; k < l1; ++k)
        for (int i = 2; i < ido; i += 2)
        {
            const int ic = ido - i;
            float dr2 = AT (cc, i - 1 + (k + l1) * ido), di2 = AT (cc, i + (k + l1) * ido);
            float dr3 = AT (cc, i - 1 + (k + l1 * 2) * ido), di3 = AT (cc, i + (k + l1 * 2) * ido);
            CMUL_CONJ (dr2, di2, wa1[i - 2], wa1[i - 1]);
            CMUL_CONJ (dr3, di3, wa2[i - 2], wa2[i - 1]);
            const float cr2 = dr2 + dr3, ci2 = di2 + di3;
            AT (ch, i - 1 + 3 * k * ido) = AT (cc, i - 1 + k * ido) + cr2;
            AT (ch, i + 3 * k * ido) = AT (cc, i + k * ido) + ci2;
            const float tr2 = AT (cc, i - 1 + k * ido) + cr2 * taur;
            const float ti2 = AT (cc, i + k * ido) + ci2 * taur;
            const float tr3 = (di2 - di3) * taui;
            const float ti3 = (dr3 - dr2) * taui;
            AT (ch, i - 1 + (3 * k + 2) * ido) = tr2 + tr3;
            AT (ch, ic - 1 + (3 * k + 1) * ido) = tr2 - tr3;
            AT (ch, i + (3 * k + 2) * ido) = ti2 + ti3;
            AT (ch, ic + (3 * k + 1) * ido) = ti3 - ti2;
        }
}

/* sse.cpp:704-786 radf5_ps (sizes with a factor 5).  0-based views of the reference's 1-based macros:
   CC(a, k, j) = cc[(j*l1 + k)*ido + a],  CH(a, j, k) = ch[(k*5 + j)*ido + a] */
static void rfwd5_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, const float* wa3, const float* wa4)
{
    const float tr11 = .309016994374947f, ti11 = .951056516295154f;
    const float tr12 = -.809016994374947f, ti12 = .587785252292473f;
#define CC(a, k, j) AT (cc, ((j) * l1 + (k)) * ido + (a))
#define CH(a, j, k) AT (ch, ((k) * 5 + (j)) * ido + (a))
    for (int k = 0; k < l1; ++k)
    {
        const float cr2 = CC (0, k, 4) + CC (0, k, 1), ci5 = CC (0, k, 4) - CC (0, k, 1);
        const float cr3 = CC (0, k, 3) + CC (0, k, 2), ci4 = CC (0, k, 3) - CC (0, k, 2);
        CH (0, 0, k) = CC (0, k, 0) + (cr2 + cr3);
        CH (ido - 1, 1, k) = CC (0, k, 0) + (cr2 * tr11 + cr3 * tr12);
        CH (0, 2, k) = ci5 * ti11 + ci4 * ti12;
        CH (ido - 1, 3, k) = CC (0, k, 0) + (cr2 * tr12 + cr3 * tr11);
        CH (0, 4, k) = ci5 * ti12 - ci4 * ti11;
    }
    if (ido == 1)
        return;
    for (int k = 0; k < l1; ++k)
        for (int ii = 2; ii < ido; ii += 2) /* ii = the reference's i - 1 (imaginary slot), ii - 1 the real slot */
        {
            const int icc = ido - ii;
            float dr2 = CC (ii - 1, k, 1), di2 = CC (ii, k, 1);
            float dr3 = CC (ii - 1, k, 2), di3 = CC (ii, k, 2);
            float dr4 = CC (ii - 1, k, 3), di4 = CC (ii, k, 3);
            float dr5 = CC (ii - 1, k, 4), di5 = CC (ii, k, 4);
            CMUL_CONJ (dr2, di2, wa1[ii - 2], wa1[ii - 1]);
            CMUL_CONJ (dr3, di3, wa2[ii - 2], wa2[ii - 1]);
            CMUL_CONJ (dr4, di4, wa3[ii - 2], wa3[ii - 1]);
            CMUL_CONJ (dr5, di5, wa4[ii - 2], wa4[ii - 1]);
            const float cr2 = dr2 + dr5, ci5 = dr5 - dr2;
            const float cr5 = di2 - di5, ci2 = di2 + di5;
            const float cr3 = dr3 + dr4, ci4 = dr4 - dr3;
            const float cr4 = di3 - di4, ci3 = di3 + di4;
            CH (ii - 1, 0, k) = CC (ii - 1, k, 0) + (cr2 + cr3);
            CH (ii, 0, k) = CC (ii, k, 0) - (ci2 + ci3);
            const float tr2 = CC (ii - 1, k, 0) + (cr2 * tr11 + cr3 * tr12);
            const float ti2 = CC (ii, k, 0) - (ci2 * tr11 + ci3 * tr12);
            const float tr3 = CC (ii - 1, k, 0) + (cr2 * tr12 + cr3 * tr11);
            const float ti3 = CC (ii, k, 0) - (ci2 * tr12 + ci3 * tr11);
            const float tr5 = cr5 * ti11 + cr4 * ti12;
            const float ti5 = ci5 * ti11 + ci4 * ti12;
            const float tr4 = cr5 * ti12 - cr4 * ti11;
            const float ti4 = ci5 * ti12 - ci4 * ti11;
            CH (ii - 1, 2, k) = tr2 - tr5;
            CH (icc - 1, 1, k) = tr2 + tr5;
            CH (ii, 2, k) = ti2 + ti5;
            CH (icc, 1, k) = ti5 - ti2;
            CH (ii - 1, 4, k) = tr3 - tr4;
            CH (icc - 1, 3, k) = tr3 + tr4;
            CH (ii, 4, k) = ti3 + ti4;
            CH (icc, 3, k) = ti4 - ti3;
        }
#undef CC
#undef CH
}

/* sse.cpp:788-846 rfftf1_ps: factors consumed last to first.  n = 2*Ncvec. */
static float* real_forward_passes (const oracle_setup* s, float* a, float* b)
{
    const int n = 2 * s->Ncvec;
    float *in = a, *out = b;
    int l2 = n, iw = n - 1;
    for (int p = s->nfac - 1; p >= 0; --p)
    {
        const int ip = s->fac[p];
        const int l1 = l2 / ip;
        const int ido = n / l2;
        iw -= (ip - 1) * ido;
        const float* wa = s->twiddle;
        for (int lane = 0; lane < LANES; ++lane)
        {
            if (ip == 4)
                rfwd4_lane (ido, l1, in + lane, out + lane, wa + iw, wa + iw + ido, wa + iw + 2 * ido);
            else if (ip == 3)
                rfwd3_lane (ido, l1, in + lane, out + lane, wa + iw, wa + iw + ido);
            else if (ip == 5)
                rfwd5_lane (ido, l1, in + lane, out + lane, wa + iw, wa + iw + ido, wa + iw + 2 * ido, wa + iw + 3 * ido);
            else
                rfwd2_lane (ido, l1, in + lane, out + lane, wa + iw);
        }
        l2 = l1;
        float* t = in;
        in = out;
        out = t;
    }
    return in;
}

/* ---- real backward passes on ONE lane (sse.cpp:1081-1124 radb2_ps, :1172-1260 radb4_ps) ------------ */
static void rbwd2_lane (int ido, int l1, const float* cc, float* ch, const float* wa1)
{
    const float minus_two = -2.0f;
    const int span = l1 * ido;
    for (int k = 0; k < span; k += ido)
    {
        const float a = AT (cc, 2 * k), b = AT (cc, 2 * (k + ido) - 1);
        AT (ch, k) = a + b;
        AT (ch, k + span) = a - b;
    }
    if (ido < 2)
        return;
    if (ido != 2)
    {
        for (int k = 0; k < span; k += ido)
            for (int i = 2; i < ido; i += 2)
            {
                const float a = AT (cc, i - 1 + 2 * k), b = AT (cc, 2 * (k + ido) - i - 1);
                const float c = AT (cc, i + 2 * k), d = AT (cc, 2 * (k + ido) - i);
                AT (ch, i - 1 + k) = a + b;
                float tr2 = a - b;
                AT (ch, i + k) = c - d;
                float ti2 = c + d;
                CMUL (tr2, ti2, wa1[i - 2], wa1[i - 1]);
                AT (ch, i - 1 + k + span) = tr2;
                AT (ch, i + k + span) = ti2;
            }
        if (ido % 2 == 1)
            return;
    }
    for (int k = 0; k < span; k += ido)
    {
        const float a = AT (cc, 2 * k + ido - 1), b = AT (cc, 2 * k + ido);
        AT (ch, k + ido - 1) = a + a;
        AT (ch, k + ido - 1 + span) = b * minus_two;
    }
}

static void rbwd4_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, const float* wa3)
{
    const float minus_sqrt2 = -1.414213562373095f;
    const int span = l1 * ido;
    for (int blk = 0; blk < l1; ++blk)
    {
        const float* c4 = cc + LANES * (4 * ido * blk);
        float* h = ch + LANES * (ido * blk);
        const float a = AT (c4, 0), b = AT (c4, 4 * ido - 1);
        const float c = AT (c4, 2 * ido), d = AT (c4, 2 * ido - 1);
        const float tr3 = d + d, tr2 = a + b, tr1 = a - b, tr4 = c + c;
        AT (h, 0 * span) = tr2 + tr3;
        AT (h, 2 * span) = tr2 - tr3;
        AT (h, 1 * span) = tr1 - tr4;
        AT (h, 3 * span) = tr1 + tr4;
    }
    if (ido < 2)
        return;
    if (ido != 2)
    {
        for (int k = 0; k < span; k += ido)
        {
#define PC(x) AT (cc, 4 * k - 1 + (x)) /* reference: pc = cc - 1 + 4k */
            for (int i = 2; i < ido; i += 2)
            {
                float* ph = ch + LANES * (k + i - 1); /* reference: ph = ch + k + 1, advanced by 2 */
                const float tr1 = PC (i) - PC (4 * ido - i);
                const float tr2 = PC (i) + PC (4 * ido - i);
                const float ti4 = PC (2 * ido + i) - PC (2 * ido - i);
                const float tr3 = PC (2 * ido + i) + PC (2 * ido - i);
                AT (ph, 0) = tr2 + tr3;
                float cr3 = tr2 - tr3;
                const float ti3 = PC (2 * ido + i + 1) - PC (2 * ido - i + 1);
                const float tr4 = PC (2 * ido + i + 1) + PC (2 * ido - i + 1);
                float cr2 = tr1 - tr4;
                float cr4 = tr1 + tr4;
                const float ti1 = PC (i + 1) + PC (4 * ido - i + 1);
                const float ti2 = PC (i + 1) - PC (4 * ido - i + 1);
                AT (ph, 1) = ti2 + ti3;
                float ci3 = ti2 - ti3;
                float ci2 = ti1 + ti4;
                float ci4 = ti1 - ti4;
                CMUL (cr2, ci2, wa1[i - 2], wa1[i - 1]);
                AT (ph, span) = cr2;
                AT (ph, span + 1) = ci2;
                CMUL (cr3, ci3, wa2[i - 2], wa2[i - 1]);
                AT (ph, 2 * span) = cr3;
                AT (ph, 2 * span + 1) = ci3;
                CMUL (cr4, ci4, wa3[i - 2], wa3[i - 1]);
                AT (ph, 3 * span) = cr4;
                AT (ph, 3 * span + 1) = ci4;
            }
        }
#undef PC
        if (ido % 2 == 1)
            return;
    }
    for (int k = 0; k < span; k += ido)
    {
        const int i0 = 4 * k + ido;
        const float c = AT (cc, i0 - 1), d = AT (cc, i0 + 2 * ido - 1);
        const float a = AT (cc, i0), b = AT (cc, i0 + 2 * ido);
        const float tr1 = c - d, tr2 = c + d, ti1 = b + a, ti2 = b - a;
        AT (ch, ido - 1 + k + 0 * span) = tr2 + tr2;
        AT (ch, ido - 1 + k + 1 * span) = (ti1 - tr1) * minus_sqrt2;
        AT (ch, ido - 1 + k + 2 * span) = ti2 + ti2;
        AT (ch, ido - 1 + k + 3 * span) = (ti1 + tr1) * minus_sqrt2;
    }
}

/* sse.cpp:1126-1170 radb3_ps (sizes with a factor 3) */
static void rbwd3_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2)
{
    const float taur = -0.5f, taui = 0.866025403784439f;
    const float taui_2 = 0.866025403784439f * 2;
    for (int k = 0; k < l1; ++k)
    {
        float tr2 = AT (cc, ido - 1 + (3 * k + 1) * ido);
        tr2 = tr2 + tr2;
        const float cr2 = AT (cc, 3 * k * ido) + tr2 * taur;
        AT (ch, k * ido) = AT (cc, 3 * k * ido) + tr2;
        const float ci3 = AT (cc, (3 * k + 2) * ido) * taui_2;
        AT (ch, (k + l1) * ido) = cr2 - ci3;
        AT (ch, (k + 2 * l1) * ido) = cr2 + ci3;
    }
    if (ido == 1)
        return;
    for (int k = 0; k < l1; ++k)
        for (int i = 2; i < ido; i += 2)
        {
            const int ic = ido - i;
            const float tr2 = AT (cc, i - 1 + (3 * k + 2) * ido) + AT (cc, ic - 1 + (3 * k + 1) * ido);
            const float cr2 = AT (cc, i - 1 + 3 * k * ido) + tr2 * taur;
            AT (ch, i - 1 + k * ido) = AT (cc, i - 1 + 3 * k * ido) + tr2;
            const float ti2 = AT (cc, i + (3 * k + 2) * ido) - AT (cc, ic + (3 * k + 1) * ido);
            const float ci2 = AT (cc, i + 3 * k * ido) + ti2 * taur;
            AT (ch, i + k * ido) = AT (cc, i + 3 * k * ido) + ti2;
            const float cr3 = (AT (cc, i - 1 + (3 * k + 2) * ido) - AT (cc, ic - 1 + (3 * k + 1) * ido)) * taui;
            const float ci3 = (AT (cc, i + (3 * k + 2) * ido) + AT (cc, ic + (3 * k + 1) * ido)) * taui;
            float dr2 = cr2 - ci3, dr3 = cr2 + ci3;
            float di2 = ci2 + cr3, di3 = ci2 - cr3;
            CMUL (dr2, di2, wa1[i - 2], wa1[i - 1]);
            AT (ch, i - 1 + (k + l1) * ido) = dr2;
            AT (ch, i + (k + l1) * ido) = di2;
            CMUL (dr3, di3, wa2[i - 2], wa2[i - 1]);
            AT (ch, i - 1 + (k + 2 * l1) * ido) = dr3;
            AT (ch, i + (k + 2 * l1) * ido) = di3;
        }
}

/* sse.cpp:1262-1356 radb5_ps (sizes with a factor 5).  0-based views of the reference's 1-based macros:
   CC(a, j, k) = cc[(k*5 + j)*ido + a],  CH(a, k, j) = ch[(j*l1 + k)*ido + a] */
static void rbwd5_lane (int ido, int l1, const float* cc, float* ch, const float* wa1, const float* wa2, const float* wa3, const float* wa4)
{
    const float tr11 = .309016994374947f, ti11 = .951056516295154f;
    const float tr12 = -.809016994374947f, ti12 = .587785252292473f;
#define CC(a, j, k) AT (cc, ((k) * 5 + (j)) * ido + (a))
#define CH(a, k, j) AT (ch, ((j) * l1 + (k)) * ido + (a))
    for (int k = 0; k < l1; ++k)
    {
        const float ti5 = CC (0, 2, k) + CC (0, 2, k);
        const float ti4 = CC (0, 4, k) + CC (0, 4, k);
        const float tr2 = CC (ido - 1, 1, k) + CC (ido - 1, 1, k);
        const float tr3 = CC (ido - 1, 3, k) + CC (ido - 1, 3, k);
        CH (0, k, 0) = CC (0, 0, k) + (tr2 + tr3);
        const float cr2 = CC (0, 0, k) + (tr2 * tr11 + tr3 * tr12);
        const float cr3 = CC (0, 0, k) + (tr2 * tr12 + tr3 * tr11);
        const float ci5 = ti5 * ti11 + ti4 * ti12;
        const float ci4 = ti5 * ti12 - ti4 * ti11;
        CH (0, k, 1) = cr2 - ci5;
        CH (0, k, 2) = cr3 - ci4;
        CH (0, k, 3) = cr3 + ci4;
        CH (0, k, 4) = cr2 + ci5;
    }
    if (ido == 1)
        return;
    for (int k = 0; k < l1; ++k)
        for (int ii = 2; ii < ido; ii += 2)
        {
            const int icc = ido - ii;
            const float ti5 = CC (ii, 2, k) + CC (icc, 1, k), ti2 = CC (ii, 2, k) - CC (icc, 1, k);
            const float ti4 = CC (ii, 4, k) + CC (icc, 3, k), ti3 = CC (ii, 4, k) - CC (icc, 3, k);
            const float tr5 = CC (ii - 1, 2, k) - CC (icc - 1, 1, k), tr2 = CC (ii - 1, 2, k) + CC (icc - 1, 1, k);
            const float tr4 = CC (ii - 1, 4, k) - CC (icc - 1, 3, k), tr3 = CC (ii - 1, 4, k) + CC (icc - 1, 3, k);
            CH (ii - 1, k, 0) = CC (ii - 1, 0, k) + (tr2 + tr3);
            CH (ii, k, 0) = CC (ii, 0, k) + (ti2 + ti3);
            const float cr2 = CC (ii - 1, 0, k) + (tr2 * tr11 + tr3 * tr12);
            const float ci2 = CC (ii, 0, k) + (ti2 * tr11 + ti3 * tr12);
            const float cr3 = CC (ii - 1, 0, k) + (tr2 * tr12 + tr3 * tr11);
            const float ci3 = CC (ii, 0, k) + (ti2 * tr12 + ti3 * tr11);
            const float cr5 = tr5 * ti11 + tr4 * ti12;
            const float ci5 = ti5 * ti11 + ti4 * ti12;
            const float cr4 = tr5 * ti12 - tr4 * ti11;
            const float ci4 = ti5 * ti12 - ti4 * ti11;
            float dr3 = cr3 - ci4, dr4 = cr3 + ci4;
            float di3 = ci3 + cr4, di4 = ci3 - cr4;
            float dr5 = cr2 + ci5, dr2 = cr2 - ci5;
            float di5 = ci2 - cr5, di2 = ci2 + cr5;
            CMUL (dr2, di2, wa1[ii - 2], wa1[ii - 1]);
            CMUL (dr3, di3, wa2[ii - 2], wa2[ii - 1]);
            CMUL (dr4, di4, wa3[ii - 2], wa3[ii - 1]);
            CMUL (dr5, di5, wa4[ii - 2], wa4[ii - 1]);
            CH (ii - 1, k, 1) = dr2;
            CH (ii, k, 1) = di2;
            CH (ii - 1, k, 2) = dr3;
            CH (ii, k, 2) = di3;
            CH (ii - 1, k, 3) = dr4;
            CH (ii, k, 3) = di4;
            CH (ii - 1, k, 4) = dr5;
            CH (ii, k, 4) = di5;
        }
#undef CC
#undef CH
}

/* sse.cpp:1358-1416 rfftb1_ps */
static float* real_backward_passes (const oracle_setup* s, float* a, float* b)
{
    const int n = 2 * s->Ncvec;
    float *in = a, *out = b;
    int l1 = 1, iw = 0;
    for (int p = 0; p < s->nfac; ++p)
    {
        const int ip = s->fac[p];
        const int l2 = ip * l1;
        const int ido = n / l2;
        const float* wa = s->twiddle;
        for (int lane = 0; lane < LANES; ++lane)
        {
            if (ip == 4)
                rbwd4_lane (ido, l1, in + lane, out + lane, wa + iw, wa + iw + ido, wa + iw + 2 * ido);
            else if (ip == 3)
                rbwd3_lane (ido, l1, in + lane, out + lane, wa + iw, wa + iw + ido);
            else if (ip == 5)
                rbwd5_lane (ido, l1, in + lane, out + lane, wa + iw, wa + iw + ido, wa + iw + 2 * ido, wa + iw + 3 * ido);
            else
                rbwd2_lane (ido, l1, in + lane, out + lane, wa + iw);
        }
        l1 = l2;
        iw += (ip - 1) * ido;
        float* t = in;
        in = out;
        out = t;
    }
    return in;
}

/* ---- real cross-lane steps (sse.cpp:849-966 finalize, :969-1078 preprocess) ----------------------- */
/* One 4x4 block of the forward combine (pffft_real_finalize_4x4).  v0r / v0i are the vectors that play
   the role of "r0" / "i0", rest points at the six following vectors r1 i1 r2 i2 r3 i3. */
static void real_finalize_block (const float* v0r, const float* v0i, const float* rest, const float* E, float* O)
{
    for (int l = 0; l < LANES; ++l)
    {
        /* after the 4x4 transposes lane l of (r_j, i_j) holds element j of the l-th (re, im) vector pair */
        const float* src_r = (l == 0) ? v0r : rest + (2 * (l - 1)) * LANES;
        const float* src_i = (l == 0) ? v0i : rest + (2 * (l - 1) + 1) * LANES;
        float r[4], im[4];
        for (int j = 0; j < 4; ++j)
        {
            r[j] = src_r[j];
            im[j] = src_i[j];
        }
        for (int j = 1; j < 4; ++j)
            CMUL (r[j], im[j], E[(2 * (j - 1)) * LANES + l], E[(2 * (j - 1) + 1) * LANES + l]);
        const float sr0 = r[0] + r[2], dr0 = r[0] - r[2];
        const float sr1 = r[1] + r[3], dr1 = r[3] - r[1];
        const float si0 = im[0] + im[2], di0 = im[0] - im[2];
        const float si1 = im[1] + im[3], di1 = im[3] - im[1];
        O[0 * LANES + l] = sr0 + sr1;
        O[1 * LANES + l] = si0 + si1;
        O[2 * LANES + l] = dr0 + di1;
        O[3 * LANES + l] = dr1 - di0;
        O[4 * LANES + l] = dr0 - di1;
        O[5 * LANES + l] = dr1 + di0;
        O[6 * LANES + l] = sr0 - sr1;
        O[7 * LANES + l] = si1 - si0;
    }
}

static void real_finalize (int Ncvec, const float* in, float* out, const float* e)
{
    const float s = (float) (M_SQRT2 / 2);
    const float zero[LANES] = { 0, 0, 0, 0 };
    const int nblk = Ncvec / LANES;
    const float* cr = in;                             /* vector 0 */
    const float* ci = in + (Ncvec * 2 - 1) * LANES;   /* last vector */

    real_finalize_block (zero, zero, in + 1 * LANES, e, out);
    for (int k = 1; k < nblk; ++k)
        real_finalize_block (in + (8 * k - 1) * LANES, in + (8 * k) * LANES, in + (8 * k + 1) * LANES, e + 6 * LANES * k, out + 8 * LANES * k);

    /* lane 0 of the first block: the bins that FFTPACK keeps purely real / at the quarter points (:943-958) */
    out[0 * LANES] = (cr[0] + cr[2]) + (cr[1] + cr[3]);
    out[1 * LANES] = (cr[0] + cr[2]) - (cr[1] + cr[3]);
    out[4 * LANES] = (cr[0] - cr[2]);
    out[5 * LANES] = (cr[3] - cr[1]);
    out[2 * LANES] = ci[0] + s * (ci[1] - ci[3]);
    out[3 * LANES] = -ci[2] - s * (ci[1] + ci[3]);
    out[6 * LANES] = ci[0] - s * (ci[1] - ci[3]);
    out[7 * LANES] = ci[2] - s * (ci[1] + ci[3]);
}

/* One 4x4 block of the backward split (pffft_real_preprocess_4x4); writes 8 vectors to O8 in the order
   r0 i0 r1 i1 r2 i2 r3 i3 (transposed back); the caller scatters them. */
static void real_preprocess_block (const float* I, const float* E, float* O8)
{
    for (int l = 0; l < LANES; ++l)
    {
        float r[4], im[4];
        for (int j = 0; j < 4; ++j)
        {
            r[j] = I[(2 * j) * LANES + l];
            im[j] = I[(2 * j + 1) * LANES + l];
        }
        const float sr0 = r[0] + r[3], dr0 = r[0] - r[3];
        const float sr1 = r[1] + r[2], dr1 = r[1] - r[2];
        const float si0 = im[0] + im[3], di0 = im[0] - im[3];
        const float si1 = im[1] + im[2], di1 = im[1] - im[2];
        r[0] = sr0 + sr1;
        r[2] = sr0 - sr1;
        r[1] = dr0 - si1;
        r[3] = dr0 + si1;
        im[0] = di0 - di1;
        im[2] = di0 + di1;
        im[1] = si0 - dr1;
        im[3] = si0 + dr1;
        for (int j = 1; j < 4; ++j)
            CMUL_CONJ (r[j], im[j], E[(2 * (j - 1)) * LANES + l], E[(2 * (j - 1) + 1) * LANES + l]);
        for (int j = 0; j < 4; ++j)
        {
            O8[(2 * l) * LANES + j] = r[j];
            O8[(2 * l + 1) * LANES + j] = im[j];
        }
    }
}

static void real_preprocess (int Ncvec, const float* in, float* out, const float* e)
{
    const float s = (float) M_SQRT2;
    const int nblk = Ncvec / LANES;
    float Xr[LANES], Xi[LANES], blk[8 * LANES];
    for (int k = 0; k < LANES; ++k)
    {
        Xr[k] = in[8 * k];
        Xi[k] = in[8 * k + 4];
    }
    /* block 0 ("first"): r0 / i0 are dropped, the other six vectors land at out[1..6] */
    real_preprocess_block (in, e, blk);
    memcpy (out + 1 * LANES, blk + 2 * LANES, 6 * LANES * sizeof (float));
    /* block k >= 1: eight vectors land at out[8k-1 .. 8k+6] */
    for (int k = 1; k < nblk; ++k)
    {
        real_preprocess_block (in + 8 * LANES * k, e + 6 * LANES * k, blk);
        memcpy (out + (8 * k - 1) * LANES, blk, 8 * LANES * sizeof (float));
    }
    /* :1062-1077 */
    float* first = out;
    float* last = out + (2 * Ncvec - 1) * LANES;
    first[0] = (Xr[0] + Xi[0]) + 2 * Xr[2];
    first[1] = (Xr[0] - Xi[0]) - 2 * Xi[2];
    first[2] = (Xr[0] + Xi[0]) - 2 * Xr[2];
    first[3] = (Xr[0] - Xi[0]) + 2 * Xi[2];
    last[0] = 2 * (Xr[1] + Xr[3]);
    last[1] = s * (Xr[1] - Xr[3]) - s * (Xi[1] + Xi[3]);
    last[2] = 2 * (Xi[3] - Xi[1]);
    last[3] = -s * (Xr[1] - Xr[3]) - s * (Xi[1] + Xi[3]);
}

/* ---- reorder between the SSE "unordered" layout and canonical order (sse.cpp:1420-1515) ------------ */
static void zip2 (const float* a, const float* b, float* lo, float* hi) /* interleave2, :85-89 */
{
    const float t[8] = { a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3] };
    memcpy (lo, t, 4 * sizeof (float));
    memcpy (hi, t + 4, 4 * sizeof (float));
}
static void unzip2 (const float* a, const float* b, float* even, float* odd) /* uninterleave2, :91-95 */
{
    const float t[8] = { a[0], a[2], b[0], b[2], a[1], a[3], b[1], b[3] };
    memcpy (even, t, 4 * sizeof (float));
    memcpy (odd, t + 4, 4 * sizeof (float));
}
static void low_high (const float* a, const float* b, float* o) /* _mm_shuffle_ps(a, b, 3210): a0 a1 b2 b3 */
{
    const float t[4] = { a[0], a[1], b[2], b[3] };
    memcpy (o, t, sizeof t);
}

/* :1420-1441.  `out` points one past the destination; vectors are written downwards. */
static void reversed_copy (int n, const float* in, int in_stride, float* out)
{
    float g0[4], g1[4], h0[4], h1[4];
    zip2 (in, in + LANES, g0, g1);
    in += in_stride * LANES;
    out -= LANES;
    low_high (g1, g0, out);
    for (int k = 1; k < n; ++k)
    {
        zip2 (in, in + LANES, h0, h1);
        in += in_stride * LANES;
        out -= LANES;
        low_high (h0, g1, out);
        out -= LANES;
        low_high (h1, h0, out);
        memcpy (g1, h1, sizeof g1);
    }
    out -= LANES;
    low_high (g0, g1, out);
}

/* :1443-1464 */
static void unreversed_copy (int n, const float* in, float* out, int out_stride)
{
    float g0[4], g1[4], h0[4], h1[4];
    memcpy (g0, in, sizeof g0);
    memcpy (g1, in, sizeof g1);
    in += LANES;
    for (int k = 1; k < n; ++k)
    {
        memcpy (h0, in, sizeof h0);
        memcpy (h1, in + LANES, sizeof h1);
        in += 2 * LANES;
        low_high (h0, g1, g1);
        low_high (h1, h0, h0);
        unzip2 (h0, g1, out, out + LANES);
        out += out_stride * LANES;
        memcpy (g1, h1, sizeof g1);
    }
    memcpy (h0, in, sizeof h0);
    memcpy (h1, g0, sizeof h1);
    low_high (h0, g1, g1);
    low_high (h1, h0, h0);
    unzip2 (h0, g1, out, out + LANES);
}

static void zreorder (const oracle_setup* s, const float* in, float* out, int direction)
{
    const int N = s->N, Ncvec = s->Ncvec;
    if (s->transform == ORACLE_REAL)
    {
        const int dk = N / 32;
        if (direction == ORACLE_FORWARD)
        {
            for (int k = 0; k < dk; ++k)
            {
                zip2 (in + (k * 8 + 0) * LANES, in + (k * 8 + 1) * LANES, out + (2 * (0 * dk + k) + 0) * LANES, out + (2 * (0 * dk + k) + 1) * LANES);
                zip2 (in + (k * 8 + 4) * LANES, in + (k * 8 + 5) * LANES, out + (2 * (2 * dk + k) + 0) * LANES, out + (2 * (2 * dk + k) + 1) * LANES);
            }
            reversed_copy (dk, in + 2 * LANES, 8, out + N / 2);
            reversed_copy (dk, in + 6 * LANES, 8, out + N);
        }
        else
        {
            for (int k = 0; k < dk; ++k)
            {
                unzip2 (in + (2 * (0 * dk + k) + 0) * LANES, in + (2 * (0 * dk + k) + 1) * LANES, out + (k * 8 + 0) * LANES, out + (k * 8 + 1) * LANES);
                unzip2 (in + (2 * (2 * dk + k) + 0) * LANES, in + (2 * (2 * dk + k) + 1) * LANES, out + (k * 8 + 4) * LANES, out + (k * 8 + 5) * LANES);
            }
            unreversed_copy (dk, in + N / 4, out + N - 6 * LANES, -8);
            unreversed_copy (dk, in + 3 * N / 4, out + N - 2 * LANES, -8);
        }
    }
    else
    {
        for (int k = 0; k < Ncvec; ++k)
        {
            const int kk = (k / 4) + (k % 4) * (Ncvec / 4);
            if (direction == ORACLE_FORWARD)
                zip2 (in + (k * 2) * LANES, in + (k * 2 + 1) * LANES, out + (kk * 2) * LANES, out + (kk * 2 + 1) * LANES);
            else
                unzip2 (in + (kk * 2) * LANES, in + (kk * 2 + 1) * LANES, out + (k * 2) * LANES, out + (k * 2 + 1) * LANES);
        }
    }
}

/* ---- transform driver (sse.cpp:1518-1601) --------------------------------------------------------
   The reference ping-pongs between `output` and `work` and arranges the parity so that the result
   lands in `output`; that bookkeeping does not change any value, so two private buffers are used and
   the result is copied out at the end (which also makes in == out trivially legal). */
static void transform_with_scratch (const oracle_setup* s, const float* in, float* out, int direction, int ordered, float* w0, float* w1)
{
    const int Ncvec = s->Ncvec;
    const size_t frame = (size_t) 2 * Ncvec * LANES;
    float* res;
    if (direction == ORACLE_FORWARD)
    {
        if (s->transform == ORACLE_REAL)
        {
            memcpy (w0, in, frame * sizeof (float));
            float* r = real_forward_passes (s, w0, w1);
            float* o = (r == w0 ? w1 : w0);
            real_finalize (Ncvec, r, o, s->e);
            res = o;
        }
        else
        {
            for (int k = 0; k < Ncvec; ++k) /* :1546-1549 de-interleave */
                unzip2 (in + (2 * k) * LANES, in + (2 * k + 1) * LANES, w0 + (2 * k) * LANES, w0 + (2 * k + 1) * LANES);
            float* r = complex_passes (s, w0, w1, -1.0f);
            float* o = (r == w0 ? w1 : w0);
            complex_finalize (Ncvec, r, o, s->e);
            res = o;
        }
        if (ordered)
        {
            float* o = (res == w0 ? w1 : w0);
            zreorder (s, res, o, ORACLE_FORWARD);
            res = o;
        }
    }
    else
    {
        const float* src = in;
        if (ordered)
        {
            zreorder (s, in, w0, ORACLE_BACKWARD);
            src = w0;
        }
        float* pre = (src == w0 ? w1 : w0);
        float* other = (pre == w0 ? w1 : w0);
        if (s->transform == ORACLE_REAL)
        {
            real_preprocess (Ncvec, src, pre, s->e);
            res = real_backward_passes (s, pre, other);
        }
        else
        {
            complex_preprocess (Ncvec, src, pre, s->e);
            res = complex_passes (s, pre, other, +1.0f);
            for (int k = 0; k < Ncvec; ++k) /* :1581-1584 re-interleave */
                zip2 (res + (2 * k) * LANES, res + (2 * k + 1) * LANES, res + (2 * k) * LANES, res + (2 * k + 1) * LANES);
        }
    }
    memcpy (out, res, frame * sizeof (float));
}

void oracle_transform (const oracle_setup* s, const float* in, float* out, int direction, int ordered)
{
    const size_t frame = (size_t) 2 * s->Ncvec * LANES;
    float* w = (float*) malloc (2 * frame * sizeof (float));
    if (! w)
        return;
    transform_with_scratch (s, in, out, direction, ordered, w, w + frame);
    free (w);
}

void oracle_transform_batch (const oracle_setup* s, const float* in, float* out, size_t batch, int direction)
{
    const size_t frame = (size_t) 2 * s->Ncvec * LANES;
    float* w = (float*) malloc (2 * frame * sizeof (float));
    if (! w)
        return;
    for (size_t f = 0; f < batch; ++f)
        transform_with_scratch (s, in + f * frame, out + f * frame, direction, 1, w, w + frame);
    free (w);
}

/* ---- sse.cpp:1603-1645 pffft_convolve_internal ---------------------------------------------------- */
void oracle_convolve (const oracle_setup* s, const float* a, const float* b, float* ab, float scaling)
{
    const int Ncvec = s->Ncvec;
    const float ar0 = a[0], ai0 = a[LANES], br0 = b[0], bi0 = b[LANES], abr0 = ab[0], abi0 = ab[LANES];
    for (int v = 0; v < Ncvec; ++v)
        for (int l = 0; l < LANES; ++l)
        {
            float ar = a[(2 * v) * LANES + l], ai = a[(2 * v + 1) * LANES + l];
            const float br = b[(2 * v) * LANES + l], bi = b[(2 * v + 1) * LANES + l];
            CMUL (ar, ai, br, bi);
            ab[(2 * v) * LANES + l] = ab[(2 * v) * LANES + l] + ar * scaling;
            ab[(2 * v + 1) * LANES + l] = ab[(2 * v + 1) * LANES + l] + ai * scaling;
        }
    if (s->transform == ORACLE_REAL)
    {
        ab[0] = abr0 + ar0 * br0 * scaling;
        ab[LANES] = abi0 + ai0 * bi0 * scaling;
    }
}

/* ---- sse.cpp:1647-1660 fft_accumulate_internal ---------------------------------------------------- */
void oracle_accumulate (const float* a, const float* b, float* ab, int N)
{
    for (int i = 0; i < N; ++i)
        ab[i] = a[i] + b[i];
}
