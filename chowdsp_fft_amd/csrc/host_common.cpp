// Error capture, device facts and the process-wide device-table cache (host side, no kernels).
#include "host_common.hpp"
#include "tables.hpp"
#include "launch.hpp"

#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace cfft::host
{
namespace
{
thread_local std::string g_last_error;
std::mutex g_cache_mutex;
std::map<std::tuple<int, int, int>, DeviceTables*> g_tables; // (device, N, transform) -> tables
std::map<int, DeviceInfo> g_devinfo;

float2* upload (const std::vector<float2>& h)
{
    if (h.empty())
        return nullptr;
    float2* d = nullptr;
    if (hipMalloc ((void**) &d, h.size() * sizeof (float2)) != hipSuccess)
        return nullptr;
    if (hipMemcpy (d, h.data(), h.size() * sizeof (float2), hipMemcpyHostToDevice) != hipSuccess)
    {
        (void) hipFree (d);
        return nullptr;
    }
    return d;
}
} // namespace

void set_last_error (const char* what, hipError_t e)
{
    g_last_error = std::string (what) + ": " + hipGetErrorString (e);
}
void set_last_error (const char* what) { g_last_error = what; }
const char* last_error() { return g_last_error.c_str(); }

bool current_device_info (DeviceInfo& out)
{
    int dev = -1;
    if (hipGetDevice (&dev) != hipSuccess || dev < 0)
    {
        set_last_error ("hipGetDevice failed (no usable HIP device)");
        return false;
    }
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    auto it = g_devinfo.find (dev);
    if (it == g_devinfo.end())
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties (&prop, dev) != hipSuccess)
        {
            set_last_error ("hipGetDeviceProperties failed");
            return false;
        }
        DeviceInfo info;
        info.ordinal = dev;
        info.num_cus = prop.multiProcessorCount;
        info.lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
        it = g_devinfo.emplace (dev, info).first;
    }
    out = it->second;
    return true;
}

int blocks_per_cu_cached (const void* kernel, int threads, size_t lds_bytes)
{
    int dev = 0;
    (void) hipGetDevice (&dev);
    static std::map<std::tuple<const void*, int, size_t>, int> cache;
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    const auto key = std::make_tuple (kernel, dev, lds_bytes);
    auto it = cache.find (key);
    if (it != cache.end())
        return it->second;
    // first use of this kernel on this device: raise the dynamic-LDS cap when the launch needs more than 64 KiB
    if (lds_bytes > 64 * 1024 && hipFuncSetAttribute (kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes) != hipSuccess)
        (void) hipGetLastError();
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, kernel, threads, lds_bytes) != hipSuccess || n < 1)
    {
        (void) hipGetLastError();
        n = 1;
    }
    cache.emplace (key, n);
    return n;
}

const DeviceTables* acquire_column_tables (int device, int rows)
{
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    const auto key = std::make_tuple (device, rows, 2 /* neither FFT_REAL nor FFT_COMPLEX: column pass */);
    auto it = g_tables.find (key);
    if (it != g_tables.end())
        return it->second;
    HostTables h;
    if (! build_column_tables (rows, h))
    {
        set_last_error ("unsupported column length");
        return nullptr;
    }
    auto* t = new DeviceTables();
    t->gen_ns = (int) h.generic_radix.size();
    for (int i = 0; i < t->gen_ns; ++i)
    {
        t->gen_radix[i] = h.generic_radix[(size_t) i];
        t->gen_tw_offset[i] = h.generic_tw_offset[(size_t) i];
    }
    t->stage_twiddles = upload (h.stage_twiddles);
    if (t->stage_twiddles == nullptr)
    {
        set_last_error ("device table upload failed");
        delete t;
        return nullptr;
    }
    g_tables.emplace (key, t);
    return t;
}

const DeviceTables* acquire_tables (int device, int N, int transform)
{
    // a frame beyond 2^21 complex points is transformed through the plans of its two factors: acquired first (the cache lock is
    // not recursive), complex plans both
    const DeviceTables *sub_r = nullptr, *sub_c = nullptr;
    {
        const int nc = (transform == 0) ? N / 2 : N;
        int hr = 0, hc = 0;
        if (runs_as_huge (nc) && size_supported (N, transform) && split_huge (nc, hr, hc))
        {
            sub_r = acquire_tables (device, hr, 1);
            sub_c = acquire_tables (device, hc, 1);
            if (sub_r == nullptr || sub_c == nullptr)
                return nullptr;
        }
    }
    // ... and, where the frame splits into columns the 2-D column kernel takes, that plan too (capi.cpp huge_transform prefers it)
    const DeviceTables *cols2 = nullptr, *plan2_c = nullptr;
    int h2r = 0, h2c = 0;
    {
        const int nc = (transform == 0) ? N / 2 : N;
        if (sub_r != nullptr && split_huge_columns (nc, h2r, h2c))
        {
            cols2 = acquire_column_tables (device, h2r);
            plan2_c = acquire_tables (device, h2c, 1);
            if (cols2 == nullptr || plan2_c == nullptr)
                cols2 = plan2_c = nullptr;
        }
    }
    // real N = 131072 can run in one launch on the complex 65536-point plan's tables (group_real_kernels.hpp)
    const DeviceTables* group_core = nullptr;
    if (transform == 0 && N == 131072)
    {
        group_core = acquire_tables (device, 65536, 1);
        if (group_core != nullptr && (group_core->fast_n1 != 0 || group_core->core_n != 0 || group_core->mixed_n2 != 0))
            group_core = nullptr; // (CHOWDSP_FFT_FAST_TILES_65536 experiment: not the 256 x 256 block-ordered plan)
    }
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    const auto key = std::make_tuple (device, N, transform);
    auto it = g_tables.find (key);
    if (it != g_tables.end())
        return it->second;

    HostTables h;
    if (! build_host_tables (N, transform, h))
    {
        set_last_error ("unsupported transform size");
        return nullptr;
    }
    auto* t = new DeviceTables();
    t->core_n = h.core_n;
    t->gen_ns = (int) h.generic_radix.size();
    for (int i = 0; i < t->gen_ns; ++i)
    {
        t->gen_radix[i] = h.generic_radix[(size_t) i];
        t->gen_tw_offset[i] = h.generic_tw_offset[(size_t) i];
    }
    t->mixed_n1 = h.mixed_n1;
    t->mixed_n2 = h.mixed_n2;
    t->mixed_tw1_offset = h.mixed_tw1_offset;
    t->mixed_ca = h.mixed_ca;
    t->mixed_cb = h.mixed_cb;
    t->fast_n1 = h.fast_n1;
    t->fast_n2 = h.fast_n2;
    t->r2_n = h.r2_n;
    t->real4_r1 = h.real4_r1;
    t->real4_r2 = h.real4_r2;
    t->real4_fast = h.real4_fast;
    t->gen2_ns = (int) h.generic2_radix.size();
    for (int i = 0; i < t->gen2_ns; ++i)
    {
        t->gen2_radix[i] = h.generic2_radix[(size_t) i];
        t->gen2_tw_offset[i] = h.generic2_tw_offset[(size_t) i];
    }
    t->huge_r = h.huge_r;
    t->huge_c = h.huge_c;
    t->huge_shift = h.huge_shift;
    t->huge_plan_r = sub_r;
    t->huge_plan_c = sub_c;
    if (cols2 != nullptr)
    {
        t->huge2_r = h2r;
        t->huge2_c = h2c;
        t->huge2_cols = cols2;
        t->huge2_plan_c = plan2_c;
    }
    t->group_core = group_core;
    t->huge_tw_lo = upload (h.huge_tw_lo);
    t->huge_tw_hi = upload (h.huge_tw_hi);
    t->stage_twiddles = upload (h.stage_twiddles);
    t->real_twiddles = upload (h.real_twiddles);
    t->big_twiddles = upload (h.big_twiddles);
    if ((! h.stage_twiddles.empty() && t->stage_twiddles == nullptr)
        || (! h.huge_tw_lo.empty() && (t->huge_tw_lo == nullptr || t->huge_tw_hi == nullptr))
        || (! h.real_twiddles.empty() && t->real_twiddles == nullptr)
        || (! h.big_twiddles.empty() && t->big_twiddles == nullptr))
    {
        set_last_error ("device table upload failed");
        (void) hipFree (t->stage_twiddles);
        (void) hipFree (t->real_twiddles);
        (void) hipFree (t->big_twiddles);
        (void) hipFree (t->huge_tw_lo);
        (void) hipFree (t->huge_tw_hi);
        delete t;
        return nullptr;
    }
    g_tables.emplace (key, t);
    return t;
}
} // namespace cfft::host
