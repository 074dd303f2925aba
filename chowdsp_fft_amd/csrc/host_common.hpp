// Host-side plumbing shared by the C-ABI translation units: error capture, device properties,
// the process-wide device-table cache and the plan header.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace cfft::host
{
// ---- error capture (nothing throws across the C ABI) -------------------------------------------
void set_last_error (const char* what, hipError_t e);
void set_last_error (const char* what);
const char* last_error();

#define CFFT_HIP_TRY(expr, failret)                        \
    do                                                     \
    {                                                      \
        hipError_t cfft_e_ = (expr);                       \
        if (cfft_e_ != hipSuccess)                         \
        {                                                  \
            ::cfft::host::set_last_error (#expr, cfft_e_); \
            return failret;                                \
        }                                                  \
    } while (0)

// ---- per-device facts ----------------------------------------------------------------------------
struct DeviceInfo
{
    int ordinal = -1;
    int num_cus = 0;
    size_t lds_per_cu = 0;
};
// Info for the CURRENT device (cached); returns false if no device is usable.
bool current_device_info (DeviceInfo& out);

// ---- device tables shared by all plans of one (device, N, transform) --------------------------
struct DeviceTables
{
    float2* stage_twiddles = nullptr; // per-stage Stockham twiddles for the LDS-resident core
    float2* real_twiddles = nullptr;  // W_N^k, k < N/4+1, for the real <-> half-length complex split
    float2* big_twiddles = nullptr;   // W_N^{a*b} table for the multi-pass (N > 4096) decomposition
    int core_n = 0;                   // length handled by the LDS-resident core (N, N/2 for REAL, or N2)
    // mixed-radix (not a power of two) plans: run-time stage list of the generic kernel, else gen_ns == 0
    int gen_ns = 0;
    int gen_radix[10] = { 0 };
    int gen_tw_offset[10] = { 0 };
    // long mixed-radix plans: two passes, gen_* describe the n2-point column FFT, the n1-point row core's table starts at
    // stage_twiddles + mixed_tw1_offset; both 0 otherwise
    int mixed_n1 = 0, mixed_n2 = 0, mixed_tw1_offset = 0, mixed_ca = 0;
    // N = 2^17 .. 2^20 complex by the register-direct passes (tile_fast_kernels.hpp): fast_n1 x fast_n2, else 0
    int fast_n1 = 0, fast_n2 = 0;
    // 2^21 complex points: radix-2 around the r2_n x r2_n register-direct two-pass transform of the two half-length sub-frames, else 0
    int r2_n = 0;
    // long real frames by the real four-step (N = real4_r1 * real4_r2), else 0
    int real4_r1 = 0, real4_r2 = 0;
    int real4_fast = 0; // ... on the register-direct tiles (tile_fast_kernels.hpp) instead of the LDS-staged ones
    int real4_generic = 0; // ... mixed-radix: gen_* = the r2-point column core's stage list, gen2_* = the r1-point row core's (real4_generic_kernels.hpp)
    // real N = 131072: the 65536-point complex plan (256 x 256, block-ordered inter-pass twiddles) whose tables the one-launch group
    // form of the REAL transform uses (group_real_kernels.hpp), else nullptr
    const DeviceTables* group_core = nullptr;
    // general form (rows generic too): mixed_cb > 0, gen2_* = the n1-point row FFT's stage list
    int mixed_cb = 0;
    int gen2_ns = 0;
    int gen2_radix[10] = { 0 };
    int gen2_tw_offset[10] = { 0 };
    // frames beyond 2^21 complex points: huge_r x huge_c (tables.hpp split_huge); the two factors' own plans, and the two-level
    // table of W_N^j = huge_tw_hi[j >> huge_shift] * huge_tw_lo[j & mask]
    int huge_r = 0, huge_c = 0, huge_shift = 0;
    const DeviceTables* huge_plan_r = nullptr;
    const DeviceTables* huge_plan_c = nullptr;
    float2* huge_tw_lo = nullptr;
    float2* huge_tw_hi = nullptr;
    // ... the same frames as huge2_r-point COLUMNS (the 2-D column kernel, twiddle at its stores) x huge2_c-point rows + one transpose (round 4), else 0
    int huge2_r = 0, huge2_c = 0;
    const DeviceTables* huge2_cols = nullptr;   // column tables of huge2_r (acquire_column_tables)
    const DeviceTables* huge2_plan_c = nullptr; // complex plan of huge2_c
};

// The plan header.  Lives either in a block we allocated (fft_new_setup) or in caller memory
// (fft_new_setup_preallocated) -- the counterpart of the reference's FFT_Setup
// (simd/chowdsp_fft_impl_sse.cpp:56-65), minus the twiddle arrays, which are device-resident here.
struct Setup
{
    uint32_t magic;     // kMagic
    int N;
    int transform;      // fft_transform_t
    int device;         // HIP ordinal the tables live on
    void* owned_block;  // block to aligned_free in fft_destroy_setup, or nullptr (preallocated)
    const DeviceTables* tables;
};
static constexpr uint32_t kMagic = 0xC4F7A355u;

inline Setup* as_setup (void* p)
{
    auto* s = reinterpret_cast<Setup*> (p);
    return (s != nullptr && s->magic == kMagic) ? s : nullptr;
}

// hipStreamSynchronize + the hand-off report of the one-launch two-pass forms (capi.cpp): what EVERY synchronous entry point
// of the library uses to wait for a stream.  FFT_GPU_OK or FFT_GPU_ELAUNCH (reason recorded for fft_gpu_last_error).
int sync_stream_checked (hipStream_t stream, const char* what);

// ---- the device-table cache (host_common.cpp) -------------------------------------------------------------------------
// Every plan holds ONE counted reference to the tables of its (device, N, transform), taken by acquire_tables and given back by
// release_tables (fft_destroy_setup; the reference frees everything a setup owns there, simd/chowdsp_fft_impl_sse.cpp:79-82).
// Tables nobody references any more are IDLE: they stay cached, least recently released first out, while the idle set fits
// table_cache_limit() bytes of device memory (default 64 MiB; 0 = freed at the last release), and trim_tables() frees all of
// them.  Composite plans (frames beyond 2^21 points, real N = 131072, the 2-D column pass) reference the tables of their
// parts; those references go when the composite entry is freed.  hipFree waits for the device, so a kernel still reading a
// table when it is released completes first.
const DeviceTables* acquire_tables (int device, int N, int transform);
// stage list + tables of the 2-D transform's column pass (rows-point FFTs of a tile of columns); nullptr if `rows` is not
// 2-3-5-smooth or does not fit a column tile (tables.hpp kGeneralMaxFactor).  Counted like acquire_tables.
const DeviceTables* acquire_column_tables (int device, int rows);
void retain_tables (const DeviceTables* t);  // one more reference to tables the caller already holds (recorded deferred calls)
void release_tables (const DeviceTables* t); // nullptr and unknown pointers are ignored
// frees the idle tables of `device` (-1: every device); returns the bytes of device memory given back
size_t trim_tables (int device);
void set_table_cache_limit (size_t bytes);
size_t table_cache_limit();
struct TableCacheStats
{
    size_t entries = 0, idle_entries = 0;  // (device, N, transform) sets cached / of them unreferenced
    size_t bytes = 0, idle_bytes = 0;      // device memory they hold
};
TableCacheStats table_cache_stats (int device); // -1: every device
// RAII: run on the device the plan's tables live on, whatever device the calling thread has current
struct DeviceGuard
{
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard (int want)
    {
        if (hipGetDevice (&prev) == hipSuccess && prev != want)
            switched = hipSetDevice (want) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched)
            (void) hipSetDevice (prev);
    }
};
} // namespace cfft::host
