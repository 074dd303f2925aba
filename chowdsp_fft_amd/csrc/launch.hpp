// Kernel launchers (defined in kernels_*.hip) used by the C-ABI layer (capi.cpp).
#pragma once
#include "host_common.hpp"

namespace cfft::host
{
// All launchers: asynchronous on `stream`, return hipSuccess or the launch error.
// n = complex frame length handled by the LDS-resident core (16..4096).

// largest frame stride in float2: 64 teams * stride must stay a 32-bit element offset
static constexpr size_t kMaxFrameStride = (size_t) 1 << 24;

// Frame strides: distance between consecutive frames in float2 units, even, >= the frame length;
// 0 = packed.  With in == out both strides must be equal.

// batched complex FFT, interleaved in/out, in may equal out
hipError_t launch_c2c (int n, bool inverse, const float2* in, float2* out, const float2* stage_tw, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);

// CFFT_EXPERIMENTS (make EXPERIMENTS=1): superseded forms kept for A/B runs -- the ticket-queue one-launch form of N = 65536 (option
// "fused_two_pass" = 1, with "fused_lag" / "fused_slots"), one wavefront per 4096-point frame ("wave4096"), the split-plane geometry for
// plain 8192-point transforms ("split8192"), one wavefront per 2048-point fused-convolution frame ("conv_wave2048").  The shipped
// library carries ONE form per size: without the flag those kernels are not instantiated and their options are refused.
#ifndef CFFT_EXPERIMENTS
#define CFFT_EXPERIMENTS 0
#endif
void set_wave4096 (bool on); // N = 4096 complex: one wavefront per frame (wave_kernels.hpp) instead of the workgroup plan

// batched real FFT of N = 2*m real points through the m-point core (m = 16..4096)
hipError_t launch_r2c (int m, const float2* in, float2* out, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
hipError_t launch_c2r (int m, const float2* in, float2* out, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);

// N = 32768 complex points / real N = 65536 in one pass, the Stockham exchanges one component at a time through a 132 KiB
// LDS plane (split_plane_kernels.hpp); `tw` = the split-plane table (tables.cpp), strides in float2 (0 = packed)
// (n / m = 32768 or 16384 complex points)
bool split_plane_length (int n);
hipError_t launch_c2c_split (int n, bool inverse, const float2* in, float2* out, const float2* tw, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
// fused forward -> * H * scaling -> backward on the same kernels (n = complex core length 32768 or 16384; real: N = 2 n samples,
// H the packed spectrum)
hipError_t launch_conv_frames_split (int n, bool real, const float2* in, const float2* H, float2* out, const float2* tw, const float2* real_tw, float scaling, size_t batch, hipStream_t stream);
hipError_t launch_real_split (int m, bool forward, const float2* in, float2* out, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);

// ab += a*b*scaling on `batch` frames of `cplx_per_frame` complex values; real_packed: bin 0 is (DC, Nyquist)
hipError_t launch_conv_mac (bool real_packed, const float* a, const float* b, float* ab, float scaling, size_t cplx_per_frame, size_t batch, bool b_is_shared, hipStream_t stream, bool accumulate = true);
hipError_t launch_accumulate (const float* a, const float* b, float* ab, size_t count, hipStream_t stream);

// streaming partitioned convolution engine (conv_engine.cpp)
hipError_t launch_fdl_mac (const float* fdl, const float* H, float* Y, float scale, size_t frame_floats, unsigned channels, unsigned P, unsigned head, bool h_shared, hipStream_t stream);
hipError_t launch_ols_assemble (const float* in, float* prev, float* frame, size_t block, unsigned channels, hipStream_t stream);
hipError_t launch_ols_extract (const float* y, float* out, size_t block, unsigned channels, hipStream_t stream);

// fused per-block kernels of the engine (block = m >= 128 and LDS-resident): see conv_kernels.hpp
bool ols_fused_supported (int m);
hipError_t launch_ols_r2c (int m, float* buf0, float* buf1, const float* cur, float* fdl, const float2* stage_tw, const float2* real_tw, unsigned channels, const void* state, hipStream_t stream);
hipError_t launch_ols_mac_c2r (int m, const float* fdl, const float* H, float* out, const float2* stage_tw, const float2* real_tw, float scale, unsigned channels, unsigned P, void* state, bool h_shared, hipStream_t stream); // also moves the position on
// the whole block in one launch (ols_block_kernel): forward transform, delay-line sum, inverse transform per channel
hipError_t launch_ols_block (int m, float* buf0, float* buf1, const float* cur, float* fdl, const float* H, float* out, const float2* stage_tw, const float2* real_tw, float scale, unsigned channels, unsigned P, void* state, bool h_shared, hipStream_t stream);
hipError_t launch_ols_advance (void* state, unsigned partitions, hipStream_t stream); // the same step on its own
size_t ols_state_bytes();                                                             // device-resident engine position

// fused forward -> *H*scaling -> backward
hipError_t launch_conv_frames_c2c (int n, const float2* in, const float2* H, float2* out, const float2* stage_tw, float scaling, size_t batch, hipStream_t stream);
hipError_t launch_conv_frames_real (int m, const float2* in, const float2* H, float2* out, const float2* stage_tw, const float2* real_tw, float scaling, size_t batch, hipStream_t stream);

// multi-pass complex FFT of nc = n1*n2 points: in -> work -> out (work holds `batch` frames)
// stage_tw holds the n2-core table followed by the n1-core table.  grid_divisor: 2 when two launches are meant
// to share the GPU (two-lane pipeline), else 1.
hipError_t launch_big_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
// N = 2^17 .. 2^20 complex, n1 and n2 in {256, 512, 1024}: register-direct passes (tile_fast_kernels.hpp).  stage_tw = the
// factorised stage-1 rows of the n2 core, then of the n1 core (fast_tile_table_rows (n) x 32 entries each); big_tw and the
// workspace in plain [a][kb] order
int fast_tile_table_rows (int n);
// mul_h != nullptr (forward only): every output bin times mul_h[bin] * mul_scale (fft_convolve_frames_batch of these sizes)
hipError_t launch_fast_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0, const float2* mul_h = nullptr, float mul_scale = 1.f);
// ... and long REAL frames on the same tiles (real four-step; big_tw = forward table [a][kb < r2/2 + 16] followed by the
// backward table [kb <= r2/2][a]; work: real4_fast_workspace_frame_len() float2 per frame)
size_t real4_fast_workspace_frame_len (int r1, int r2);
hipError_t launch_real4_fast (int r1, int r2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0, const float2* mul_h = nullptr, float mul_scale = 1.f);
// long REAL frames (N = r1 * r2 samples, both powers of two in 256..1024): the four-step algorithm on the real data, two
// sweeps (real columns as pairs -> rows -> packed spectrum; backward the other way round).  in / out: the real frame and the
// packed spectrum, both as N/2 float2; work: real4_workspace_frame_len() float2 per frame.
size_t real4_workspace_frame_len (int r1, int r2);
// ... with generic cores, for long mixed-radix real frames (kernels_real4_generic.hip; the same workspace layout)
// complex frames on general two-pass plans whose factors are both at most 639 points: fft_convolve_frames_batch in three sweeps (conv4_generic_kernels.hpp)
bool conv4_generic_applies (const DeviceTables& t);
hipError_t launch_conv4_generic (const DeviceTables& t, const float2* in, const float2* H, float scaling, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
hipError_t launch_real4_generic_conv (const DeviceTables& t, const float2* in, const float2* H, float scaling, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
hipError_t launch_real4_generic (const DeviceTables& t, bool inverse, const float2* in, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
hipError_t launch_real4 (int r1, int r2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
// the same two passes in ONE launch with the intermediate handed over inside an XCD's L2 (fused_twopass_kernels.hpp):
// `state` = fused_state_bytes() of device memory (zeroed by the launcher on `stream`), `work` = fused_workspace_bytes()
bool fused_supported (int n1, int n2);
size_t fused_state_bytes();
size_t fused_workspace_bytes (int nc, int slots);
// host_err: a pinned, device-visible word that a launch whose bounded waits ran out sets to (seq << 8) | code; else untouched
hipError_t launch_fused_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, int lag, int slots, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
// ... and with static groups of 16 workgroups per XCD instead of ticket queues (group_twopass_kernels.hpp); same arguments,
// `state` = group_state_bytes(), `work` = group_workspace_bytes()
size_t group_state_bytes();
void set_group_options (int force_solo, int reg_polls); // -1 = leave as it is
size_t group_workspace_bytes();
// ... and for N = 512 x 256 / 512 x 512 on the register-direct tiles (group_fast_kernels.hpp): one workgroup per CU
bool groupfast_supported (int n1, int n2);
size_t groupfast_workspace_bytes (int n1, int n2);
hipError_t launch_groupfast_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
void get_group_options (int& force_solo, int& reg_polls);
// option "group_handoff": 1 = the hand-off form the CDNA4 guide lists (sc1 payload and flag stores, one workgroup per CU) for the
// one-launch forms that offer it (complex N = 65536, real N = 131072); group_lds_bytes: the dynamic LDS such a launch asks for
void set_group_handoff (int listed);
int get_group_handoff();
size_t group_lds_bytes (size_t kernel_lds);
void set_group_spin_limit (int polls); // developer option: 0 = built-in limit
int get_group_spin_limit();
int group_wg_per_cu_tenths();
unsigned group_conv_grid(); // workgroups of the fused-convolution form's launch (kernels_group_conv.hip)
hipError_t launch_group_conv (const float2* in, const float2* H, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, float scaling, size_t batch, hipStream_t stream);
// 2^19 .. 2^21 complex points: radix-2 decimation in time around the 1024 x n2 register-direct passes of the two half-length sub-frames
// (tile_fast_kernels.hpp); stage_tw = the two cores' rows, then W_N^kb (kb < n2), then W_2048^ms (ms < 32); big_tw = W_(N/2)^(a kb);
// mul_h (forward only): every bin times mul_h[bin] * mul_scale
hipError_t launch_fast_r2_c2c (int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0, const float2* mul_h = nullptr, float mul_scale = 1.f);
// real N = 131072 on the same groups, pair stage in registers (group_real_kernels.hpp, kernels_group_real.hip): stage_tw / big_tw of
// the 65536-point complex plan, real_tw = W_131072^k of the real plan
unsigned group_real_grid (bool inverse);
hipError_t launch_group_real (bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, const float2* real_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0, const float2* mul_h = nullptr, float mul_scale = 1.f);
hipError_t launch_group_c2c (int n1, int n2, bool inverse, const float2* in, float2* work, float2* out, const float2* stage_tw, const float2* big_tw, void* state, unsigned* host_err, unsigned seq, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
// long mixed-radix frames (t.mixed_n2 > 0): generic column pass + power-of-two row pass, in -> work -> out
hipError_t launch_mixed_big_c2c (const DeviceTables& t, bool inverse, const float2* in, float2* work, float2* out, size_t batch, int grid_divisor, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);
hipError_t launch_real_pair (bool forward, const float2* in, float2* out, const float2* real_tw, unsigned m, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);

// mixed-radix (non power of two) frames: mode 0 c2c, 1 real forward, 2 real backward; nc = complex length
hipError_t launch_generic (int mode, bool inverse, int nc, const DeviceTables& t, const float2* in, float2* out, size_t batch, hipStream_t stream, size_t in_stride = 0, size_t out_stride = 0);

// fused forward -> * H * scaling -> backward for mixed-radix frames up to 20000 complex points (generic_conv_kernels.hpp);
// hipErrorNotSupported where no one-launch form exists (the caller then composes the three launches)
hipError_t launch_generic_conv (bool real, int nc, const DeviceTables& t, const float2* in, const float2* H, float2* out, float scaling, size_t batch, hipStream_t stream);

// column pass of the 2-D transform (row f4): rows-point FFTs along the columns of `batch` row-major [rows][cols] complex arrays,
// in place; `t` = acquire_column_tables (device, rows)
// src (nullptr: in place): where the arrays are read from; tw_lo / tw_hi / tw_shift (nullptr: none): every result times W_(rows cols)^(k col)
hipError_t launch_column_fft (bool inverse, int rows, int cols, const DeviceTables& t, float2* data, size_t batch, hipStream_t stream, const float2* src = nullptr, const float2* tw_lo = nullptr, const float2* tw_hi = nullptr, unsigned tw_shift = 0);

// layout conversions of the JUCE-engine adapter (kernels_layout.hip); data = batch frames of 2*m float2
// real transform with the JUCE engine's layout fused into the pair stage, in place on frames of 4 m floats; hipErrorNotSupported
// for core lengths without a fused form (the caller then converts in a second launch)
hipError_t launch_real_juce (int m, bool forward, float2* inout, const float2* stage_tw, const float2* real_tw, size_t batch, hipStream_t stream, bool mirror, float scale);
// ... and for m = 16384 / 32768 (real N = 32768 / 65536) on the split-plane kernels; `tw` = the split-plane table of that length
hipError_t launch_real_split_juce (int m, bool forward, float2* inout, const float2* tw, const float2* real_tw, size_t batch, hipStream_t stream, bool mirror, float scale);
hipError_t launch_juce_unpack (float2* data, unsigned m, size_t batch, bool mirror, hipStream_t stream);
hipError_t launch_juce_pack_scale (float2* data, unsigned m, size_t batch, float scale, hipStream_t stream);
hipError_t launch_scale (float* data, size_t count, float scale, hipStream_t stream); // count % 4 == 0
// out[c][r] = in[r][c] for one [rows][cols] complex matrix (frames beyond 2^21 points); twiddle 0 none, +1 times W_N^(r c),
// -1 times its conjugate, N = rows * cols, W_N^j = tw_hi[j >> shift] * tw_lo[j & ((1 << shift) - 1)]; in != out
hipError_t launch_transpose (const float2* in, float2* out, unsigned rows, unsigned cols, int twiddle, const float2* tw_lo, const float2* tw_hi, unsigned shift, hipStream_t stream);

// shared by the kernels_*.hip files
int blocks_per_cu_cached (const void* kernel, int threads, size_t lds_bytes);
} // namespace cfft::host
