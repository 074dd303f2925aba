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
using TableKey = std::tuple<int, int, int>; // (device, N, transform; 2 = the 2-D column pass of N rows)
struct TableEntry
{
    DeviceTables* t = nullptr;
    int refs = 0;          // plans, recorded deferred calls and composite entries holding these tables
    size_t bytes = 0;      // device memory of this entry's own arrays
    uint64_t idle_stamp = 0; // when refs last dropped to 0 (larger = later)
    std::vector<const DeviceTables*> parts; // tables of other entries this one references (one count each)
};
std::map<TableKey, TableEntry> g_tables;
std::map<const DeviceTables*, TableKey> g_table_keys;
uint64_t g_idle_clock = 0;
size_t g_table_cache_limit = (size_t) 64 << 20;
std::map<int, DeviceInfo> g_devinfo;

float2* upload (const std::vector<float2>& h, size_t& bytes)
{
    if (h.empty())
        return nullptr;
    bytes += h.size() * sizeof (float2);
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

void free_arrays (DeviceTables* t)
{
    (void) hipFree (t->stage_twiddles);
    (void) hipFree (t->real_twiddles);
    (void) hipFree (t->big_twiddles);
    (void) hipFree (t->huge_tw_lo);
    (void) hipFree (t->huge_tw_hi);
}

// (cache lock held)  one reference less; an entry without references becomes idle
void unref_locked (const DeviceTables* t)
{
    auto k = g_table_keys.find (t);
    if (k == g_table_keys.end())
        return;
    TableEntry& e = g_tables[k->second];
    if (e.refs > 0 && --e.refs == 0)
        e.idle_stamp = ++g_idle_clock;
}

// (cache lock held)  frees one idle entry; its parts lose the reference it held and may become idle themselves
size_t free_entry_locked (std::map<TableKey, TableEntry>::iterator it)
{
    TableEntry e = it->second;
    const int device = std::get<0> (it->first);
    g_table_keys.erase (e.t);
    g_tables.erase (it);
    {
        const DeviceGuard guard (device);
        free_arrays (e.t);
    }
    delete e.t;
    for (const DeviceTables* p : e.parts)
        unref_locked (p);
    return e.bytes;
}

// (cache lock held)  idle entries beyond the limit go, least recently released first
void enforce_limit_locked()
{
    for (;;)
    {
        size_t idle = 0;
        auto oldest = g_tables.end();
        for (auto it = g_tables.begin(); it != g_tables.end(); ++it)
            if (it->second.refs == 0)
            {
                idle += it->second.bytes;
                if (oldest == g_tables.end() || it->second.idle_stamp < oldest->second.idle_stamp)
                    oldest = it;
            }
        if (oldest == g_tables.end() || idle <= g_table_cache_limit)
            return;
        (void) free_entry_locked (oldest);
    }
}

// (cache lock held)  cached tables of `key` with one more reference, or nullptr
const DeviceTables* lookup_ref_locked (const TableKey& key)
{
    auto it = g_tables.find (key);
    if (it == g_tables.end())
        return nullptr;
    ++it->second.refs;
    return it->second.t;
}
} // namespace

void retain_tables (const DeviceTables* t)
{
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    auto k = g_table_keys.find (t);
    if (k != g_table_keys.end())
        ++g_tables[k->second].refs;
}

void release_tables (const DeviceTables* t)
{
    if (t == nullptr)
        return;
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    unref_locked (t);
    enforce_limit_locked();
}

size_t trim_tables (int device)
{
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    size_t freed = 0;
    // freeing a composite entry can make its parts idle: repeat until a sweep frees nothing
    for (bool again = true; again;)
    {
        again = false;
        for (auto it = g_tables.begin(); it != g_tables.end(); ++it)
            if (it->second.refs == 0 && (device < 0 || std::get<0> (it->first) == device))
            {
                freed += free_entry_locked (it);
                again = true;
                break; // (the erase invalidated the iteration)
            }
    }
    return freed;
}

void set_table_cache_limit (size_t bytes)
{
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    g_table_cache_limit = bytes;
    enforce_limit_locked();
}

size_t table_cache_limit()
{
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    return g_table_cache_limit;
}

TableCacheStats table_cache_stats (int device)
{
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    TableCacheStats st;
    for (const auto& kv : g_tables)
        if (device < 0 || std::get<0> (kv.first) == device)
        {
            ++st.entries;
            st.bytes += kv.second.bytes;
            if (kv.second.refs == 0)
            {
                ++st.idle_entries;
                st.idle_bytes += kv.second.bytes;
            }
        }
    return st;
}

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
    const TableKey key = std::make_tuple (device, rows, 2 /* neither FFT_REAL nor FFT_COMPLEX: column pass */);
    if (const DeviceTables* hit = lookup_ref_locked (key))
        return hit;
    HostTables h;
    if (! build_column_tables (rows, h))
    {
        set_last_error ("unsupported column length");
        return nullptr;
    }
    TableEntry e;
    auto* t = new DeviceTables();
    t->gen_ns = (int) h.generic_radix.size();
    for (int i = 0; i < t->gen_ns; ++i)
    {
        t->gen_radix[i] = h.generic_radix[(size_t) i];
        t->gen_tw_offset[i] = h.generic_tw_offset[(size_t) i];
    }
    {
        const DeviceGuard guard (device);
        t->stage_twiddles = upload (h.stage_twiddles, e.bytes);
    }
    if (t->stage_twiddles == nullptr)
    {
        set_last_error ("device table upload failed");
        delete t;
        return nullptr;
    }
    e.t = t;
    e.refs = 1;
    g_tables.emplace (key, e);
    g_table_keys.emplace (t, key);
    return t;
}

const DeviceTables* acquire_tables (int device, int N, int transform)
{
    // (a real plan made while option "real4_generic" is on is a different plan of the same length: a key of its own)
    int g1 = 0, g2 = 0;
    const TableKey key = std::make_tuple (device, N, (transform == 0 && split_real4_generic (N, g1, g2)) ? 16 + get_real4_generic() : transform);
    {
        std::lock_guard<std::mutex> lock (g_cache_mutex);
        if (const DeviceTables* hit = lookup_ref_locked (key))
            return hit;
    }
    // Not cached: the tables of the plan's parts first (each acquire takes the cache lock itself; the references belong to the
    // new entry and go back if it is not created after all).
    std::vector<const DeviceTables*> parts;
    const auto drop_parts = [&parts] {
        for (const DeviceTables* p : parts)
            release_tables (p);
        parts.clear();
    };
    // a frame beyond 2^21 complex points is transformed through the plans of its two factors, complex plans both
    const DeviceTables *sub_r = nullptr, *sub_c = nullptr;
    const int nc = (transform == 0) ? N / 2 : N;
    {
        int hr = 0, hc = 0;
        if (runs_as_huge (nc) && size_supported (N, transform) && split_huge (nc, hr, hc))
        {
            sub_r = acquire_tables (device, hr, 1);
            if (sub_r != nullptr)
                parts.push_back (sub_r);
            sub_c = acquire_tables (device, hc, 1);
            if (sub_c != nullptr)
                parts.push_back (sub_c);
            if (sub_r == nullptr || sub_c == nullptr)
            {
                drop_parts();
                return nullptr;
            }
        }
    }
    // ... and, where the frame splits into columns the 2-D column kernel takes, that plan too (capi.cpp huge_transform prefers it)
    const DeviceTables *cols2 = nullptr, *plan2_c = nullptr;
    int h2r = 0, h2c = 0;
    if (sub_r != nullptr && split_huge_columns (nc, h2r, h2c))
    {
        cols2 = acquire_column_tables (device, h2r);
        plan2_c = acquire_tables (device, h2c, 1);
        if (cols2 == nullptr || plan2_c == nullptr)
        {
            release_tables (cols2);
            release_tables (plan2_c);
            cols2 = plan2_c = nullptr;
        }
        else
        {
            parts.push_back (cols2);
            parts.push_back (plan2_c);
        }
    }
    // real N = 131072 can run in one launch on the complex 65536-point plan's tables (group_real_kernels.hpp)
    const DeviceTables* group_core = nullptr;
    if (transform == 0 && N == 131072)
    {
        group_core = acquire_tables (device, 65536, 1);
        if (group_core != nullptr && (group_core->fast_n1 != 0 || group_core->core_n != 0 || group_core->mixed_n2 != 0))
        {
            release_tables (group_core); // (CHOWDSP_FFT_FAST_TILES_65536 experiment: not the 256 x 256 block-ordered plan)
            group_core = nullptr;
        }
        if (group_core != nullptr)
            parts.push_back (group_core);
    }

    HostTables h;
    if (! build_host_tables (N, transform, h))
    {
        set_last_error ("unsupported transform size");
        drop_parts();
        return nullptr;
    }
    TableEntry e;
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
    t->real4_generic = h.real4_generic;
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
    {
        const DeviceGuard guard (device);
        t->huge_tw_lo = upload (h.huge_tw_lo, e.bytes);
        t->huge_tw_hi = upload (h.huge_tw_hi, e.bytes);
        t->stage_twiddles = upload (h.stage_twiddles, e.bytes);
        t->real_twiddles = upload (h.real_twiddles, e.bytes);
        t->big_twiddles = upload (h.big_twiddles, e.bytes);
    }
    if ((! h.stage_twiddles.empty() && t->stage_twiddles == nullptr)
        || (! h.huge_tw_lo.empty() && (t->huge_tw_lo == nullptr || t->huge_tw_hi == nullptr))
        || (! h.real_twiddles.empty() && t->real_twiddles == nullptr)
        || (! h.big_twiddles.empty() && t->big_twiddles == nullptr))
    {
        set_last_error ("device table upload failed");
        free_arrays (t);
        delete t;
        drop_parts();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock (g_cache_mutex);
    if (const DeviceTables* raced = lookup_ref_locked (key))
    {
        // another thread built the same entry meanwhile: keep that one
        free_arrays (t);
        delete t;
        for (const DeviceTables* p : parts)
            unref_locked (p);
        enforce_limit_locked();
        return raced;
    }
    e.t = t;
    e.refs = 1;
    e.parts = parts;
    g_tables.emplace (key, e);
    g_table_keys.emplace (t, key);
    return t;
}
} // namespace cfft::host
