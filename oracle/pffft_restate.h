/*
 * oracle/pffft_restate.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's SSE (4-lane) algorithm for the hot path; see pffft_restate.c.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef ORACLE_PFFFT_RESTATE_H
#define ORACLE_PFFFT_RESTATE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_FORWARD = 0, ORACLE_BACKWARD = 1 }; /* = fft_direction_t */
enum { ORACLE_REAL = 0, ORACLE_COMPLEX = 1 };     /* = fft_transform_t */

typedef struct oracle_setup oracle_setup;

/* NULL exactly when the reference's SSE path returns NULL: N % 16 (complex) / N % 32 (real), N/4 = 2^a 3^b 5^c. */
oracle_setup* oracle_new_setup (int N, int transform);
void oracle_destroy_setup (oracle_setup* s);

/* One frame; ordered != 0 gives fft_transform, ordered == 0 gives fft_transform_unordered (SSE layout).
   in and out may alias.  Unscaled in both directions. */
void oracle_transform (const oracle_setup* s, const float* in, float* out, int direction, int ordered);

/* ab += (a*b)*scaling on SSE-unordered spectra (fft_convolve_unordered). */
void oracle_convolve (const oracle_setup* s, const float* a, const float* b, float* ab, float scaling);

/* ab = a + b (fft_accumulate). */
void oracle_accumulate (const float* a, const float* b, float* ab, int N);

/* convenience for timing / tests: `batch` contiguous frames, ordered transform */
void oracle_transform_batch (const oracle_setup* s, const float* in, float* out, size_t batch, int direction);

/* introspection used by the tests that pin the plan against the reference */
int oracle_num_passes (const oracle_setup* s);
int oracle_pass_radix (const oracle_setup* s, int pass);

#ifdef __cplusplus
}
#endif
#endif
