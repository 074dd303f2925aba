/*
  reference_sequence.c -- the reference's C test (reference test/test.c:31-139) as a program against the drop-in
  library, with the comparator it cannot have here (upstream pffft, not available offline) replaced by vectors
  read from files: the outputs of the reference's SSE path (tests/golden/reference_sse.npz where it holds the size,
  else the bit-identical oracle restatement), dumped to raw float32 by tests/test_c_program_gpu.py.

  For every case file given on the command line, twice (fft_new_setup and fft_new_setup_preallocated, as the
  reference's "SSE/NEON" and "pre-allocated" loops do, test.c:141-178):
      in-place ordered FORWARD with a work buffer          -> compare with the reference's forward result
      in-place ordered BACKWARD, both sides scaled by 1/N  -> compare with the reference's backward result
  Two bounds per comparison: the reference test's own absolute tolerance (test.c:11-16: 1e-6 * len / 8) and the
  parity bound of this repository (relative L2 over the frame <= 1e-6).

  case file layout (little endian): int32 N, int32 is_real, then 3 * frame floats: input, forward, backward
  (frame = N real / 2N complex; backward = BACKWARD(forward), unscaled).

  exit code 0 and a final line "gpu-ok" = every case passed on a GPU.  Any failure prints the case and exits non-zero;
  fft_new_setup returning NULL (no usable device) is a FAILURE here: this program is the GPU leg of the C ABI test.
*/
#include "chowdsp_fft.h"
#include "chowdsp_fft_gpu.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int compare (const float* ref, const float* test, size_t len, double scale, const char* what, const char* path)
{
    /* reference test.c:11-16 */
    const float tol = 1.0e-6f * (float) len / 8.0f;
    double num = 0.0, den = 0.0;
    for (size_t n = 0; n < len; ++n)
    {
        const double r = (double) ref[n] * scale, t = (double) test[n] * scale;
        if (! (fabs (r - t) < (double) tol))
        {
            printf ("FAIL %s %s: element %zu: %g vs %g (abs tol %g)\n", path, what, n, t, r, (double) tol);
            return 1;
        }
        num += (r - t) * (r - t);
        den += r * r;
    }
    const double rel = den > 0.0 ? sqrt (num / den) : sqrt (num);
    if (! (rel <= 1.0e-6))
    {
        printf ("FAIL %s %s: relative L2 %.3e > 1e-6\n", path, what, rel);
        return 1;
    }
    return 0;
}

static int run_case (const char* path, int preallocate)
{
    FILE* f = fopen (path, "rb");
    if (f == NULL)
    {
        printf ("FAIL cannot open %s\n", path);
        return 1;
    }
    int32_t hdr[2];
    if (fread (hdr, sizeof (int32_t), 2, f) != 2)
    {
        fclose (f);
        return 1;
    }
    const int N = hdr[0];
    const fft_transform_t tr = hdr[1] ? FFT_REAL : FFT_COMPLEX;
    const size_t len = hdr[1] ? (size_t) N : 2 * (size_t) N;
    float* data = (float*) aligned_malloc (sizeof (float) * len);
    float* work = (float*) aligned_malloc (sizeof (float) * len);
    float* fwd = (float*) malloc (sizeof (float) * len);
    float* bwd = (float*) malloc (sizeof (float) * len);
    int bad = (data == NULL || work == NULL || fwd == NULL || bwd == NULL);
    if (! bad)
        bad = fread (data, sizeof (float), len, f) != len || fread (fwd, sizeof (float), len, f) != len
              || fread (bwd, sizeof (float), len, f) != len;
    fclose (f);
    if (bad)
    {
        printf ("FAIL short read / allocation %s\n", path);
        return 1;
    }

    void* setup;
    void* prealloc = NULL;
    if (preallocate)
    {
        const size_t bytes_required = fft_bytes_required (N, tr, false);
        prealloc = aligned_malloc (bytes_required);
        setup = fft_new_setup_preallocated (N, tr, prealloc, false);
    }
    else
        setup = fft_new_setup (N, tr, false);
    if (setup == NULL)
    {
        printf ("FAIL %s: fft_new_setup returned NULL (%s)\n", path, fft_gpu_last_error());
        return 1;
    }
    if (fft_simd_width_bytes (setup) != 16) /* reference test/test.cpp:44-45 without AVX */
    {
        printf ("FAIL %s: fft_simd_width_bytes != 16\n", path);
        return 1;
    }

    int rc = 0;
    fft_transform (setup, data, data, work, FFT_FORWARD);
    rc |= compare (fwd, data, len, 1.0, "forward", path);
    memcpy (data, fwd, sizeof (float) * len); /* the stored backward result is BACKWARD of the reference's own spectrum */
    fft_transform (setup, data, data, work, FFT_BACKWARD);
    rc |= compare (bwd, data, len, 1.0 / (double) N, "backward * 1/N", path);

    if (preallocate)
        aligned_free (prealloc); /* caller-owned: no fft_destroy_setup (reference chowdsp_fft.h:110-112) */
    else
        fft_destroy_setup (setup);
    aligned_free (data);
    aligned_free (work);
    free (fwd);
    free (bwd);
    if (rc == 0)
        printf ("ok %s N=%d %s%s\n", hdr[1] ? "real" : "complex", N, preallocate ? "pre-allocated " : "", path);
    return rc;
}

int main (int argc, char** argv)
{
    if (argc < 2)
    {
        printf ("usage: %s case.bin ...\n", argv[0]);
        return 2;
    }
    if (fft_gpu_device_count() < 1)
    {
        printf ("FAIL no HIP device visible to libchowdsp_fft_hip.so\n");
        return 3;
    }
    int failures = 0;
    printf ("Running SSE/NEON Tests\n");
    for (int i = 1; i < argc; ++i)
        failures += run_case (argv[i], 0);
    printf ("Running pre-allocated Tests\n");
    for (int i = 1; i < argc; ++i)
        failures += run_case (argv[i], 1);
    if (failures != 0)
    {
        printf ("%d failure(s)\n", failures);
        return 1;
    }
    printf ("Testing complete!\ngpu-ok\n");
    return 0;
}
