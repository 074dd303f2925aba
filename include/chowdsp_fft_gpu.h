/*
  chowdsp_fft_gpu.h -- batched, device-pointer extension of the chowdsp_fft C interface for MI355X.

  The reference API transforms exactly one frame per call and has no batch argument: every caller
  loops over frames itself (reference test/test.cpp:47-62, bench/bench.cpp:36-45).  On a GPU the unit
  of work is a large batch of independent frames, so this header adds, next to the eleven legacy
  symbols of chowdsp_fft.h, batch variants of the same five operations plus a fused
  forward -> multiply-accumulate -> backward convolution kernel.  Each entry point names the legacy
  function (and reference file:line) whose per-frame semantics it repeats `batch` times.

  Conventions
    * `setup` is a handle from fft_new_setup / fft_new_setup_preallocated (chowdsp_fft.h).
    * all data pointers are DEVICE pointers (hipMalloc, fft_gpu_malloc, torch tensors ...);
      frames are contiguous: frame b starts at element b * frame_floats, where frame_floats is
      2*N (COMPLEX) or N (REAL).  in and out may alias.
    * `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
      with respect to the host; nothing in here synchronises unless its name says so.
    * return value: 0 on success, a negative FFT_GPU_E* code otherwise.  Nothing throws, nothing aborts.
*/
#pragma once

#include "chowdsp_fft.h"

#ifdef __cplusplus
#include <cstddef>
extern "C"
{
namespace chowdsp::fft
{
#else
#include <stddef.h>
#endif

enum
{
    FFT_GPU_OK = 0,
    FFT_GPU_EINVAL = -1,   /* NULL / malformed argument */
    FFT_GPU_ENODEV = -2,   /* no usable HIP device */
    FFT_GPU_ELAUNCH = -3,  /* kernel launch or runtime call failed */
    FFT_GPU_ENOMEM = -4,   /* device allocation failed */
    FFT_GPU_ENOTSUP = -5   /* operation not available for this plan */
};

/* ---- device / memory helpers (thin wrappers so that C callers need no HIP headers) ------------ */
int fft_gpu_device_count (void);
int fft_gpu_set_device (int ordinal);
void* fft_gpu_malloc (size_t nb_bytes);           /* device memory; NULL on failure */
void fft_gpu_free (void* p);
void* fft_gpu_host_malloc (size_t nb_bytes);      /* pinned host memory (counterpart of aligned_malloc) */
void fft_gpu_host_free (void* p);
/* kind: 0 = host->device, 1 = device->host, 2 = device->device; asynchronous on `stream` */
int fft_gpu_memcpy (void* dst, const void* src, size_t nb_bytes, int kind, void* stream);
int fft_gpu_stream_synchronize (void* stream);
/* Last HIP error string seen by this thread inside the library ("" if none). */
const char* fft_gpu_last_error (void);

/* Two-pass plans cache, per (device, stream) they are called on, their workspace (fft_gpu_workspace_bytes; with
   CHOWDSP_FFT_LANES=2 two of them plus two helper streams and three events).  This releases what is cached for `stream` (on every device): call it before
   destroying the stream -- a recycled handle would inherit the entry -- or to give the memory back.  Waits for the stream.
   Returns the number of entries released.  A workspace that would have to GROW while its stream is being captured into
   a hipGraph makes the call fail with FFT_GPU_ENOMEM (run the same call once outside the capture first). */
int fft_gpu_release_stream_resources (void* stream);

/* Device tables (twiddles, stage lists) are shared by every plan of one (device, N, transform) and counted: each plan holds
   one reference from fft_new_setup / fft_new_setup_preallocated until fft_destroy_setup (the reference frees everything a
   setup owns there).  Tables without a plan stay cached for the next fft_new_setup of the same size while the idle set fits
   option "table_cache_mib" (default 64; 0 = freed with their last plan), oldest out first.
   fft_gpu_trim_tables frees every idle table of `device` (-1: all devices) and returns the bytes of device memory given
   back; fft_gpu_release_stream_resources calls it too.  A plan made by fft_new_setup_preallocated whose owner follows the
   reference's rule (no fft_destroy_setup) keeps its tables pinned; calling fft_destroy_setup on it before freeing the
   caller's block is allowed here and drops the reference.
   fft_gpu_table_cache_info: info[0..3] = { table sets cached, their bytes, idle sets, idle bytes } for `device` (-1: all). */
size_t fft_gpu_trim_tables (int device);
int fft_gpu_table_cache_info (int device, size_t info[4]);

/* Run-time options (process-wide; initial values come from the environment variables named in INTEGRATION.md):
     "fused_two_pass"   which form runs two-pass sizes that have a one-launch form (N = 65536, 2^17, 2^18 complex), batches of at
                        least "fused_min_batch" frames:  2 (default) one launch, static groups of 16 workgroups per XCD, the
                        inter-pass hand-off inside that XCD's L2 (1.4 x the algorithmic bytes on the fabric);  1 one launch,
                        per-XCD ticket queues (1.5-1.6 x, slower);  0 two launches per chunk (2 x).
                        A one-launch transform whose bounded hand-off waits ran out (never observed) has written invalid
                        output: that is reported, with the launch's number on its stream, by the next one-launch call on
                        the stream and by fft_gpu_stream_synchronize / fft_gpu_release_stream_resources -- results of these
                        forms are validated by a LIBRARY synchronise, not by hipStreamSynchronize alone.
     "fused_min_batch"  smallest batch that takes a one-launch form (default 256)
     "fused_lag", "fused_slots"   queue depth of the ticket form (steps between a frame's two passes; workspace
                        frames per XCD, > lag)
     "group_force_solo" 1: the group form runs its fall-back mode (every workgroup transforms whole frames alone; what a
                        launch does by itself when its grid is not completely resident); for tests, default 0
     "group_reg_polls"  how many polls (about a microsecond each) a workgroup of the group form waits for the grid to be
                        complete before the launch falls back (default 200)
     "wave4096"         1: N = 4096 complex transforms run one wavefront per frame (two radix-64 stages in registers, one
                        wave-private exchange) instead of one 256-work-item workgroup per frame; same rate, default 0
     "split16384"       0: N = 16384 complex / real N = 32768 through the round-1 plan instead of the split-plane kernels
     "split8192"        1: N = 8192 complex / real N = 16384 through the split-plane kernels (256 work-items x 32 points)
                        instead of the 512-work-item LDS-image plan; same rate, default 0
     "fused_conv_split" 0: fft_convolve_frames_batch of the 8192 / 16384 / 32768-point cores as three launches instead of
                        the one-launch split-plane form (A/B switch; default 1)
     "fused_conv_generic" 0: fft_convolve_frames_batch of mixed-radix frames (up to 20000 complex points) as three launches
                        instead of the one-launch form (A/B switch; default 1)
     "group_fast"       0: N = 2^17 / 2^18 complex as two launches per chunk instead of the one-launch group form on the
                        register-direct tiles (one workgroup per CU; 0.35 -> 0.38 of 8 TB/s; batches of at least
                        "fused_min_batch" / 4 frames; A/B switch, default 1)
     "fused_conv_group" 0: fft_convolve_frames_batch of N = 65536 complex frames (batches of at least "fused_min_batch") as three
                        launches instead of one launch on the static groups of "fused_two_pass" = 2 (A/B switch; default 1)
     "conv_wave2048"    1: fft_convolve_frames_batch of 2048-point cores by one wavefront per frame; 12-14 % slower than
                        the workgroup plan, default 0
     "group_handoff"    how the one-launch forms pass the intermediate between workgroups.  0 (default): plain stores of payload and flag
                        (the lines stay in the L2 of the XCD whose workgroups read them back), two workgroups per CU -- the fast form,
                        validated by the stress tests of this repository but NOT one of the hand-off forms the CDNA4 programming guide
                        lists as measured-valid.  1: the listed form -- payload and flag stores `sc1` (write-through), every storing
                        wave's wait, a workgroup barrier, one lane's flag; `sc1` poll, barrier, `sc1` loads; ONE workgroup per CU
                        (enforced through the launch's LDS size) -- for complex N = 65536 and real N = 131072; the forms that do not
                        offer it (complex 2^17 / 2^18 groups, the N = 65536 fused convolution) then run their multi-launch composition,
                        which has no in-kernel hand-off.  Results are identical; what the listed form costs: INTEGRATION.md.
     "real4_generic"    how plans MADE from now on run long mixed-radix REAL frames (complex length beyond the one-pass kernels) that
                        split as N = R1 x R2 with R1 = 32 j, R2 even, both at most 639 points (41472, 96000, 245760 ...):
                        1 (default) the real four-step with generic cores -- two sweeps over the data instead of the complex
                        two-pass transform of N / 2 points + the pair stage (three) -- where both factors have a compile-time stage
                        list (0.25-0.30 of 8 TB/s against 0.21-0.24); 2 for every such split and with run-time stage lists only
                        (0.17-0.18, slower than the three sweeps: it keeps those kernels tested and measurable); 0 never.
     "table_cache_mib"  device memory (MiB) that tables of destroyed plans may keep for re-use (default 64; 0 = none)
     "workspace_mib"    intermediate (MiB) one chunk of a two-pass plan may occupy (default 256 = the Infinity Cache; initial value
                        CHOWDSP_FFT_WORKSPACE_MIB): calls made from now on work through their batch in chunks of at most this much
                        workspace, one after the other on the caller's stream.  A workspace that is already larger stays.  Results
                        do not depend on it; rates do (profiles/r05_lanes_sweep.txt: 64 -> 512 MiB).
   Returns 0, or FFT_GPU_EINVAL for an unknown name / a value out of range. */
int fft_gpu_set_option (const char* name, long value);
/* Current value of an option ("fused_two_pass", "fused_min_batch", "fused_lag", "fused_slots", "split16384", "split8192",
   "conv_wave2048", "fused_conv_split", "fused_conv_generic", "fused_conv_group", "group_fast"), or -1 for a name that cannot be read back. */
long fft_gpu_get_option (const char* name);

/* ---- plan introspection -------------------------------------------------------------------------- */
/* 1 if fft_new_setup would accept (N, transform) on a machine with a usable device, else 0.  Pure host logic (no GPU
   needed).  The rule is the reference's SSE size rule -- N % 16 == 0 (complex) / N % 32 == 0 (real), N = 2^a 3^b 5^c
   (reference simd/chowdsp_fft_impl_common.hpp:168-177, 216-225) -- for every such N up to a complex length of 2^21
   (complex N <= 2097152, real N <= 4194304).  The reference itself has no upper bound; beyond 2^21 complex points a size is
   accepted when it is the product of two accepted complex lengths of at most 2^21 each (every power of two, and every
   2^a 3^b 5^c with a >= 8), up to 2^27 complex points = 1 GiB per frame (real: N <= 2^25); longer frames return NULL.
   How a size runs: fft_gpu_plan_describe. */
int fft_gpu_size_supported (int N, fft_transform_t transform);
/* How (N, transform) is executed.  Fills desc[0..7] = { complex length transformed, number of passes through device
   memory (1 = the frame stays on chip, 2 = two-pass), n1, n2 (two-pass split, else 0), then up to four stage radices of
   the on-chip core (of the n2 core for two-pass plans; 0-padded; all 0 where both passes use run-time radix lists) }.
     one pass : powers of two up to 32768 complex points (16384: radix <= 16 stages in an LDS image; 32768: three radix-32
                stages, exchanges one component at a time), other sizes up to 20000 complex points (run-time radix list)
     two pass : 65536 ... 2^20: power-of-two columns x power-of-two rows (65536 in ONE launch for large batches);
                other lengths: generic columns x power-of-two rows where N = (odd * 2^j <= 1280, j >= 4) x (2^6..2^10)
                exists (24576 = 48 x 512), else generic columns x generic rows, both factors near sqrt(N)
                (20736 = 144 x 144, 46080 = 240 x 192, 2^21 = 2048 x 1024): two tile lengths with compile-time stage lists
                (every 2-3-5-smooth length 144 ... 1200 and nine up to 2048) where N factors into them, which also covers
                such products beyond 2^21, up to 2^22 complex points (2^22 = 2048 x 2048, 2304000 = 1600 x 1440)
     "3"      : beyond 2^21 complex points: n1 x n2 with both factors accepted lengths, n2 batched n1-point transforms and
                n1 batched n2-point transforms by those lengths' own plans, a twiddle and three transposes in between
                (3 x 2^21 = 3072 x 2048, 2^24 = 4096 x 4096): five or more sweeps over the frame -- completeness, not speed
   Returns 0, or FFT_GPU_EINVAL for an unsupported size.  Pure host logic. */
int fft_gpu_plan_describe (int N, fft_transform_t transform, int desc[8]);
int fft_gpu_setup_size (void* setup);              /* N, or negative error */
int fft_gpu_setup_is_real (void* setup);           /* 1 REAL, 0 COMPLEX */
int fft_gpu_setup_device (void* setup);            /* HIP device ordinal the plan's tables live on; -1 for a bad handle */
/* scratch the plan needs per launch for two-pass sizes (complex length > 16384), bytes for `batch` frames;
   0 for LDS-resident sizes.  The library allocates and caches it internally per (device, stream). */
size_t fft_gpu_workspace_bytes (void* setup, size_t batch);

/* ---- batched operations -------------------------------------------------------------------------- */

/* `batch` x fft_transform (chowdsp_fft.h; reference chowdsp_fft.h:138,
   simd/chowdsp_fft_impl_sse.cpp:1518-1601 with ordered = 1).  Buffers of every batched call must be 16-byte aligned
   (the reference's own requirement for its SIMD buffers, :1527-1528); a misaligned address is refused with
   FFT_GPU_EINVAL rather than left to fault inside a kernel. */
int fft_transform_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream);

/* Strided batch: the same transform on frames embedded in larger buffers, frame b
   starting at input + b*in_stride / output + b*out_stride (strides in FLOATS, multiples of 4, >= the frame
   length; buffers 16-byte aligned).  The reference's counterpart is the caller's loop over fft_transform with
   advancing pointers (e.g. chowdsp_fft_juce/chowdsp_fft_juce.cpp:47-55, one 2N-float JUCE buffer per frame).
   In place (input == output) needs in_stride == out_stride; any other overlap is FFT_GPU_EINVAL. */
int fft_transform_batch_strided (void* setup, const float* input, size_t in_stride, float* output, size_t out_stride, size_t batch, fft_direction_t direction, void* stream);

/* 2-D complex transform (the reference has no 2-D entry point -- its callers would loop
   fft_transform over the rows and over gathered columns): `batch` row-major [rows][cols] complex arrays, cols = the N of
   `setup` (a COMPLEX plan), rows = 2^a 3^b 5^c <= 2559.  A cols-point transform along every row (the ordinary batched
   transform of batch * rows frames), then a rows-point transform along every column (tiles of adjacent columns staged
   through LDS), in place on `output`; unscaled like every transform here: BACKWARD(FORWARD(x)) = rows * cols * x.
   input == output allowed. */
int fft_transform_2d_batch (void* setup, int rows, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream);

/* `batch` x fft_transform_unordered (reference chowdsp_fft.h:145; ordered = 0). */
int fft_transform_unordered_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction, void* stream);

/* `batch` x fft_convolve_unordered: ab[f] += (a[f] * b[f or 0]) * scaling
   (reference chowdsp_fft.h:154, simd/chowdsp_fft_impl_sse.cpp:1603-1645).
   b_is_shared != 0: `dft_b` is ONE frame (e.g. a filter spectrum) used for every frame of the batch. */
int fft_convolve_unordered_batch (void* setup, const float* dft_a, const float* dft_b, float* dft_ab, float scaling, size_t batch, int b_is_shared, void* stream);

/* ab[i] = a[i] + b[i] for i < count floats (reference chowdsp_fft.h:160,
   simd/chowdsp_fft_impl_sse.cpp:1647-1660); count = batch * N for a batch of frames. */
int fft_accumulate_batch (void* setup, const float* a, const float* b, float* ab, size_t count, void* stream);

/* Fused frequency-domain convolution of a batch of time-domain frames with one filter:
       y[f] = BACKWARD( FORWARD(x[f]) * H ) * scaling                       (H = dft_filter, one frame,
   i.e. the reference call sequence of test/test.cpp:161-166 / :212-215       unordered layout)
       fft_transform_unordered(x, FORWARD) -> fft_convolve_unordered(X, H, 0-initialised, scaling)
       -> fft_transform_unordered(BACKWARD)
   in ONE kernel for every power of two that runs as a single pass (N <= 32768 complex / N <= 65536 real): the
   spectrum never leaves the CU, HBM traffic is one read + one write per frame.  Every other supported size (two-pass
   and mixed-radix plans) runs the same three steps as separate launches with `y` holding the spectra in between
   (x == y allowed either way). */
int fft_convolve_frames_batch (void* setup, const float* x, const float* dft_filter, float* y, float scaling, size_t batch, void* stream);

/* ---- several devices in one process (SURVEY section 8 row e) ------------------------------------------------------
   Frames are independent: a batch is cut into contiguous blocks, one per device, plans are replicated (one setup created
   on each device after fft_gpu_set_device), and nothing is exchanged while transforming.
   fft_gpu_shard_range: block `rank` of `shards` = frames [*first, *first + *count); blocks differ by at most one frame. */
int fft_gpu_shard_range (size_t total_frames, int shards, int rank, size_t* first, size_t* count);
/* `shards` x fft_transform_batch, shard d on the device of setups[d] (device pointers of that device, counts[d] frames),
   each from its own host thread on that device's own stream, all at once.  Synchronous: every shard is complete on
   return.  Several shards may name the same device. */
int fft_transform_batch_multi_gpu (void* const* setups, int shards, const float* const* inputs, float* const* outputs, const size_t* counts, fft_direction_t direction);
/* `shards` x fft_convolve_frames_batch (BASELINE configs[3]: overlap-save frames sharded over the GPUs of a node): shard d on
   the device of setups[d], y[d] = BACKWARD (FORWARD (x[d]) * dft_filters[d] * scaling) for counts[d] frames.  The filter spectrum
   is REPLICATED: dft_filters[d] is a copy on shard d's device (one frame, fft_transform_unordered layout; the same pointer for
   shards that share a device) -- plans and filters are immutable and replicable (reference chowdsp_fft.h:87-91), so nothing is
   exchanged while convolving.  Same threading, streams and synchronous return as fft_transform_batch_multi_gpu; x[d] == y[d]
   allowed.  Replaces the caller-side loop test/test.cpp:161-166 over a sharded batch. */
int fft_convolve_frames_batch_multi_gpu (void* const* setups, int shards, const float* const* x, const float* const* dft_filters, float* const* y, const size_t* counts, float scaling);
/* Collect result shards: shard s (shard_floats[s] floats on device shard_devices[s]) is copied behind the shards before
   it into EVERY non-NULL destination dst_ptrs[d] (a buffer of sum(shard_floats) floats on device dst_devices[d]) --
   an all-gather when every device is a destination, a gather-to-root when one is.  Each (source, destination) pair is
   one direct peer copy on its own stream, all pairs in flight together.  Synchronous on both sides: the call first
   waits for every source and destination DEVICE (so shards produced by fft_transform_batch on a stream of the caller's, or a
   destination the caller has just filled, are safe), and every copy is complete on return. */
int fft_gather_multi_gpu (const float* const* shard_ptrs, const int* shard_devices, const size_t* shard_floats, int shards, float* const* dst_ptrs, const int* dst_devices, int ndst);

/* ---- host-memory streaming mode ------------------------------------------------
   `batch` x fft_transform for callers whose frames live in HOST memory (what every reference caller has:
   test/test.cpp:18-21, bench/bench.cpp:22-29, chowdsp_fft_juce.cpp:32-46) without paying one staged copy, one
   launch and one synchronisation per frame like the legacy symbols do.  The host buffers are pinned in place for
   the duration of the call, cut into chunks, and chunk i's host->device copy, transform and device->host copy
   run on stream i%2, so that the PCIe transfers of one chunk overlap the other chunk's.  Synchronous: the result
   is in `output` on return.  input == output is allowed.  PCIe-bound by construction. */
int fft_transform_host_batch (void* setup, const float* input, float* output, size_t batch, fft_direction_t direction);

/* Call coalescing for callers that can only make single-frame legacy calls with host pointers (second part of row f3).
   Between fft_gpu_defer_begin() and fft_gpu_defer_flush() the calling thread's fft_transform / fft_transform_unordered
   calls are recorded instead of executed; the flush gathers the recorded inputs into pinned memory, runs ONE batched
   transform per (setup, direction) and writes the outputs.  The observable result is that of the calls executed in order:
   a call whose input or output overlaps the output of a recorded call flushes first (so a forward / backward pair on one
   buffer is not coalesced -- independent frames are); fft_convolve_unordered / fft_accumulate and calls with device
   pointers flush and then run as usual.  Until the flush returns the recorded inputs must stay valid and unmodified and
   the outputs are not written.  Per thread; 128 MiB of recorded frames flush by themselves. */
int fft_gpu_defer_begin (void);
int fft_gpu_defer_flush (void);

/* ---- JUCE FFT-engine layouts on the device (second half of row f3) ------------------------------------------
   Batched counterparts of the three methods of the reference's JUCE adapter
   (chowdsp_fft_juce/chowdsp_fft_juce.cpp:32-86): same buffer shapes, layout and scaling, `batch` device buffers
   per call instead of one host buffer.  For power-of-two N <= 16384 the real methods are ONE launch: the layout is the
   transform's own store / load pattern; longer and mixed-radix frames run the transform and a layout kernel.
   perform: complex transform, output != or == input, the inverse scaled by 1/N (:32-45). */
int fft_juce_perform_batch (void* setup_complex, const float* input, float* output, size_t batch, int inverse, void* stream);
/* performRealOnlyForwardTransform (:47-67): `batch` buffers of 2N floats with the N samples at the start of each;
   on return each holds N/2+1 interleaved complex bins (Im of DC and Nyquist = 0) and, unless
   ignore_negative_freqs, the conjugate bins N/2+1 .. N-1. */
int fft_juce_real_forward_batch (void* setup_real, float* inout, size_t batch, int ignore_negative_freqs, void* stream);
/* performRealOnlyInverseTransform (:69-86): N/2+1 interleaved bins at the start of each 2N-float buffer in, the N
   real samples scaled by 1/N at the start of each buffer out. */
int fft_juce_real_inverse_batch (void* setup_real, float* inout, size_t batch, void* stream);

/* ---- streaming partitioned convolution engine -----------------------------------
   The caller-side structure that fft_convolve_unordered's "ab += a*b*scaling" and fft_accumulate exist for
   (reference chowdsp_fft.h:147-160, usage test/test.cpp:212-218), which the reference leaves to its callers:
   a uniformly partitioned overlap-save convolver for `channels` independent streams.  An impulse response of
   ir_len taps is cut into P = ceil(ir_len / block) partitions of `block` taps, each kept as the (unordered) REAL
   spectrum of a 2*block frame.  Every fft_conv_engine_process() call consumes `block` new samples per channel and
   produces `block` output samples per channel:
       frame = [previous block | new block]                      (2*block samples)
       X_t   = fft_transform_unordered (frame, FORWARD)            -> pushed into a frequency-domain delay line
       Y     = sum_{p<P} X_{t-p} * H_p * (1 / (2*block))          (P x fft_convolve_unordered)
       y     = fft_transform_unordered (Y, BACKWARD), last `block` samples are the output
   All state (delay line, filter spectra, previous block) lives on the device.  `block` must be a power of two,
   16 <= block <= 2^19.  ir: host or device pointer, [ir_len] if ir_is_shared else [channels][ir_len].
   in / out: device pointers, [channels][block] floats each.  Returns NULL / negative codes on failure.
   For block <= 4096 a block is one kernel launch and the engine's position is device-resident too, so one
   fft_conv_engine_process() call captured into a hipGraph can be replayed for every following block (refresh `in`,
   launch the graph); other block sizes advance their position on the host and must be called, not replayed. */
void* fft_conv_engine_create (int block, int channels, const float* ir, int ir_len, int ir_is_shared, void* stream);
int fft_conv_engine_process (void* engine, const float* in, float* out, void* stream);
int fft_conv_engine_reset (void* engine, void* stream);
int fft_conv_engine_partitions (void* engine);
void fft_conv_engine_destroy (void* engine);

#ifdef __cplusplus
} // namespace chowdsp::fft
} // extern "C"
#endif
