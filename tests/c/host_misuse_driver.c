/* Host-side walk over the C ABI WITHOUT a GPU, for the sanitizer build of the library's host code
   (`make -C chowdsp_fft_amd/csrc san`; tests/test_host_sanitizers.py): the size rule and the planner's description for every length
   up to a bound, byte counts, aligned allocation, options, and every entry point with NULL / bad handles, a fake handle that is not
   a plan, and (no device) the plan constructors -- all must come back with codes, never touch freed or foreign memory.
   Reference behaviour for the legacy names: /root/reference/chowdsp_fft.h:75-134 (setup creation returns NULL for unsupported
   sizes; the transform calls are void). */
#include "chowdsp_fft.h"
#include "chowdsp_fft_gpu.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond)                                                  \
    do                                                               \
    {                                                                \
        if (! (cond))                                                \
        {                                                            \
            fprintf (stderr, "line %d: %s\n", __LINE__, #cond);      \
            return 1;                                                \
        }                                                            \
    } while (0)

int main (int argc, char** argv)
{
    const int top = argc > 1 ? atoi (argv[1]) : 70000;
    long accepted = 0;
    for (int tr = 0; tr < 2; ++tr)
        for (int n = -3; n <= top; ++n)
        {
            const int ok = fft_gpu_size_supported (n, (fft_transform_t) tr);
            int d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            const int rc = fft_gpu_plan_describe (n, (fft_transform_t) tr, d);
            CHECK ((ok == 1) == (rc == 0));
            if (ok == 1)
            {
                ++accepted;
                CHECK (fft_bytes_required (n, (fft_transform_t) tr, true) >= (size_t) n * 4);
                CHECK (d[0] >= 1);
            }
        }
    /* long and huge lengths: the planner's table of splits */
    static const int longs[] = { 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 24, 1 << 27, 41472, 96000, 122880, 245760, 655360, 1310720,
                                 34992, 48000, 1327104, 6291456, 3 << 20, 5 << 19, 15 << 17 };
    for (unsigned i = 0; i < sizeof (longs) / sizeof (longs[0]); ++i)
        for (int tr = 0; tr < 2; ++tr)
        {
            int d[8];
            (void) fft_gpu_plan_describe (longs[i], (fft_transform_t) tr, d);
            (void) fft_bytes_required (longs[i], (fft_transform_t) tr, false);
        }
    CHECK (fft_gpu_plan_describe (4096, FFT_COMPLEX, NULL) != 0);

    void* p = aligned_malloc (1000);
    CHECK (p != NULL && ((size_t) p % 64) == 0);
    memset (p, 0, 1000);
    aligned_free (p);
    aligned_free (NULL);

    /* plans cannot be made without a device: NULL, with a message */
    const int ndev = fft_gpu_device_count();
    if (ndev <= 0)
    {
        CHECK (fft_new_setup (4096, FFT_COMPLEX, true) == NULL);
        CHECK (fft_new_setup (1024, FFT_REAL, false) == NULL);
        void* mem = aligned_malloc (fft_bytes_required (1024, FFT_REAL, true));
        CHECK (fft_new_setup_preallocated (1024, FFT_REAL, mem, true) == NULL);
        aligned_free (mem);
        CHECK (fft_conv_engine_create (256, 2, NULL, 0, 1, NULL) == NULL);
        CHECK (fft_gpu_malloc (64) == NULL);
        CHECK (fft_gpu_set_device (0) != 0);
    }
    CHECK (fft_new_setup (24, FFT_COMPLEX, true) == NULL); /* unsupported size */
    CHECK (fft_new_setup (16, FFT_REAL, true) == NULL);
    CHECK (fft_new_setup_preallocated (1024, FFT_REAL, NULL, true) == NULL);

    /* NULL handles and a handle that is not a plan (the magic word is checked before anything is followed) */
    float buf[256];
    memset (buf, 0, sizeof (buf));
    void* fakes[2] = { NULL, (void*) buf };
    for (int k = 0; k < 2; ++k)
    {
        void* s = fakes[k];
        CHECK (fft_gpu_setup_size (s) < 0);
        CHECK (fft_gpu_setup_is_real (s) < 0);
        CHECK (fft_gpu_setup_device (s) == -1);
        CHECK (fft_gpu_workspace_bytes (s, 8) == 0);
        CHECK (fft_transform_batch (s, buf, buf, 1, FFT_FORWARD, NULL) != 0);
        CHECK (fft_transform_batch_strided (s, buf, 64, buf, 64, 1, FFT_FORWARD, NULL) != 0);
        CHECK (fft_transform_2d_batch (s, 4, buf, buf, 1, FFT_FORWARD, NULL) != 0);
        CHECK (fft_transform_unordered_batch (s, buf, buf, 1, FFT_BACKWARD, NULL) != 0);
        CHECK (fft_convolve_unordered_batch (s, buf, buf, buf, 1.f, 1, 0, NULL) != 0);
        CHECK (fft_accumulate_batch (s, buf, buf, buf, 4, NULL) != 0);
        CHECK (fft_convolve_frames_batch (s, buf, buf, buf, 1.f, 1, NULL) != 0);
        CHECK (fft_transform_host_batch (s, buf, buf, 1, FFT_FORWARD) != 0);
        CHECK (fft_juce_perform_batch (s, buf, buf, 1, 0, NULL) != 0);
        CHECK (fft_juce_real_forward_batch (s, buf, 1, 0, NULL) != 0);
        CHECK (fft_juce_real_inverse_batch (s, buf, 1, NULL) != 0);
        CHECK (fft_simd_width_bytes (s) <= 16);
        fft_transform (s, buf, buf, NULL, FFT_FORWARD);
        fft_transform_unordered (s, buf, buf, NULL, FFT_FORWARD);
        fft_convolve_unordered (s, buf, buf, buf, 1.f);
        fft_accumulate (s, buf, buf, buf, 16);
        if (s == NULL)
            fft_destroy_setup (s);
        CHECK (fft_conv_engine_process (s, buf, buf, NULL) != 0);
        CHECK (fft_conv_engine_reset (s, NULL) != 0);
        CHECK (fft_conv_engine_partitions (s) < 0);
    }
    fft_conv_engine_destroy (NULL);
    CHECK (fft_transform_batch (NULL, NULL, NULL, 4, FFT_FORWARD, NULL) != 0);

    /* shards: pure host arithmetic */
    size_t first = 7, count = 7, sum = 0;
    for (int r = 0; r < 8; ++r)
    {
        CHECK (fft_gpu_shard_range (1000003, 8, r, &first, &count) == 0);
        CHECK (first == sum);
        sum += count;
    }
    CHECK (sum == 1000003);
    CHECK (fft_gpu_shard_range (10, 0, 0, &first, &count) != 0);
    CHECK (fft_gpu_shard_range (10, 4, 4, &first, &count) != 0);
    CHECK (fft_gpu_shard_range (10, 4, 1, NULL, &count) != 0);
    CHECK (fft_transform_batch_multi_gpu (NULL, 2, NULL, NULL, NULL, FFT_FORWARD) != 0);
    CHECK (fft_convolve_frames_batch_multi_gpu (NULL, 1, NULL, NULL, NULL, NULL, 1.f) != 0);
    CHECK (fft_gather_multi_gpu (NULL, NULL, NULL, 1, NULL, NULL, 1) != 0);

    /* options: every documented name round-trips or is refused with a code; unknown names are refused */
    static const char* names[] = { "fused_two_pass", "fused_min_batch", "split16384", "fused_conv_split", "fused_conv_generic", "fused_conv_group", "group_fast",
                                   "group_real", "huge_columns", "real4_generic", "group_handoff", "table_cache_mib", "workspace_mib", "experiments" };
    for (unsigned i = 0; i < sizeof (names) / sizeof (names[0]); ++i)
    {
        const long v = fft_gpu_get_option (names[i]);
        if (v >= 0 && strcmp (names[i], "experiments") != 0)
            CHECK (fft_gpu_set_option (names[i], v) == 0 && fft_gpu_get_option (names[i]) == v);
    }
    CHECK (fft_gpu_set_option ("no_such_option", 1) != 0);
    CHECK (fft_gpu_set_option (NULL, 1) != 0);
    CHECK (fft_gpu_get_option ("no_such_option") == -1);
    CHECK (fft_gpu_set_option ("workspace_mib", 0) != 0);
    CHECK (fft_gpu_set_option ("real4_generic", 3) != 0);
    CHECK (strlen (fft_gpu_last_error()) > 0);

    size_t info[4] = { 9, 9, 9, 9 };
    (void) fft_gpu_table_cache_info (0, info);
    (void) fft_gpu_table_cache_info (0, NULL);
    (void) fft_gpu_trim_tables (-1);
    (void) fft_gpu_release_stream_resources (NULL);
    (void) fft_gpu_stream_synchronize (NULL);
    CHECK (fft_gpu_memcpy (NULL, NULL, 16, 0, NULL) != 0);
    fft_gpu_free (NULL);
    fft_gpu_host_free (NULL);
    /* the deferred-call window: opened and closed with nothing in it, closed twice */
    const int db = fft_gpu_defer_begin();
    (void) db;
    (void) fft_gpu_defer_flush();
    (void) fft_gpu_defer_flush();

    printf ("%ld accepted lengths up to %d, %d device(s)\n", accepted, top, ndev);
    return 0;
}
