// oracle/ref_bench.cpp -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Times the REAL reference (oracle/_ref/libchowdsp_fft_ref.so, compiled in place from /root/reference
// by oracle/Makefile) on the host CPU, for bench.py's `cpu_baseline` leg.  The library is opened with
// dlopen(RTLD_LOCAL) and driven only through the reference's public C API (chowdsp_fft.h:81-163), the
// way the reference's own bench does (bench/bench.cpp:20-71: in-place FORWARD + BACKWARD with a heap
// `work` buffer), with three regimes:
//   hot      one thread, one frame transformed over and over (cache-hot; what the "single-thread AVX
//            reference" of the north star means)
//   stream   one thread walking a large batch of distinct frames out of place (DRAM-bound, like the GPU run)
//   allcore  T threads sharing ONE read-only setup (chowdsp_fft.h:87-91), each streaming its own block
// Output: one JSON object on stdout.
//
// usage: ref_bench <lib.so> <N> <real|complex> <avx 0|1> <seconds per regime> <threads> <stream MiB>
#include <dlfcn.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

using clk = std::chrono::steady_clock;

struct Api
{
    void* (*new_setup) (int, int, bool);
    void (*destroy_setup) (void*);
    int (*simd_width) (void*);
    void (*transform) (void*, const float*, float*, float*, int);
    void* (*amalloc) (size_t);
    void (*afree) (void*);
};

template <class F>
static bool sym (void* h, const char* name, F& out)
{
    out = reinterpret_cast<F> (dlsym (h, name));
    return out != nullptr;
}

// CPUs this process may actually use: the affinity mask, cut down by a cgroup CPU quota where one is set (v2
// cpu.max = "<quota> <period>" or "max ..."; v1 cpu.cfs_quota_us / cpu.cfs_period_us).  hardware_concurrency()
// reports the machine, not the container: a 256-thread host with a 16-CPU quota runs 16 threads' worth of work.
static int affinity_cpus()
{
    cpu_set_t set;
    CPU_ZERO (&set);
    if (sched_getaffinity (0, sizeof (set), &set) == 0)
        return CPU_COUNT (&set);
    return (int) std::thread::hardware_concurrency();
}
static double cgroup_cpu_quota() // CPUs, or 0 when unlimited / unknown
{
    if (FILE* f = fopen ("/sys/fs/cgroup/cpu.max", "r"))
    {
        char q[64] = { 0 };
        long long period = 0;
        const int n = fscanf (f, "%63s %lld", q, &period);
        fclose (f);
        if (n == 2 && strcmp (q, "max") != 0 && period > 0)
            return (double) atoll (q) / (double) period;
        return 0;
    }
    long long quota = -1, period = 0;
    if (FILE* f = fopen ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))
    {
        if (fscanf (f, "%lld", &quota) != 1)
            quota = -1;
        fclose (f);
    }
    if (FILE* f = fopen ("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))
    {
        if (fscanf (f, "%lld", &period) != 1)
            period = 0;
        fclose (f);
    }
    return (quota > 0 && period > 0) ? (double) quota / (double) period : 0;
}

// Measured parallelism: some containers show every host CPU in the affinity mask and no cgroup quota, yet are
// throttled to a few CPUs' worth of time by a layer the process cannot read.  Spin a register-only loop on 1 thread and on
// `nthreads` threads for `secs` each and return (aggregate rate with nthreads) / (rate with one).
static double parallel_speedup_probe (int nthreads, double secs)
{
    auto run = [secs] (int nt) {
        std::atomic<unsigned long long> total { 0 };
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back ([&total, secs] {
                const auto t0 = clk::now();
                unsigned long long n = 0;
                unsigned x = 12345u;
                do
                {
                    for (int i = 0; i < 20000; ++i)
                        x = x * 1664525u + 1013904223u;
                    n += 20000;
                } while (std::chrono::duration<double> (clk::now() - t0).count() < secs);
                total += n + (x & 1u);
            });
        for (auto& th : pool)
            th.join();
        return (double) total.load();
    };
    const double one = run (1);
    const double many = run (nthreads);
    return one > 0 ? many / one : 1.0;
}

static double seconds_since (clk::time_point t0) { return std::chrono::duration<double> (clk::now() - t0).count(); }

int main (int argc, char** argv)
{
    if (argc < 8)
    {
        fprintf (stderr, "usage: %s lib.so N real|complex avx seconds threads stream_mib\n", argv[0]);
        return 2;
    }
    const char* libpath = argv[1];
    const int N = atoi (argv[2]);
    const bool is_real = std::string (argv[3]) == "real";
    const bool avx = atoi (argv[4]) != 0;
    const double secs = atof (argv[5]);
    int threads = atoi (argv[6]);
    const size_t stream_mib = (size_t) atol (argv[7]);
    const int aff = affinity_cpus();
    const double quota = cgroup_cpu_quota();
    int effective = aff;
    if (quota > 0 && quota < (double) effective)
        effective = (int) (quota + 0.5) > 0 ? (int) (quota + 0.5) : 1;
    const int probe_threads = effective < 64 ? effective : 64;
    const double probe = probe_threads > 1 ? parallel_speedup_probe (probe_threads, 0.25) : 1.0;
    if (probe_threads > 1 && probe < 0.75 * (double) probe_threads) // throttled below what the masks promise
        effective = (int) (probe + 0.5) > 0 ? (int) (probe + 0.5) : 1;
    if (threads <= 0)
        threads = effective; // the all-core regime runs one thread per CPU this process can really use

    void* h = dlopen (libpath, RTLD_NOW | RTLD_LOCAL);
    if (! h)
    {
        fprintf (stderr, "dlopen: %s\n", dlerror());
        return 3;
    }
    Api api {};
    if (! (sym (h, "fft_new_setup", api.new_setup) && sym (h, "fft_destroy_setup", api.destroy_setup)
           && sym (h, "fft_simd_width_bytes", api.simd_width) && sym (h, "fft_transform", api.transform)
           && sym (h, "aligned_malloc", api.amalloc) && sym (h, "aligned_free", api.afree)))
    {
        fprintf (stderr, "missing symbol\n");
        return 4;
    }
    void* setup = api.new_setup (N, is_real ? 0 : 1, avx);
    if ((reinterpret_cast<uintptr_t> (setup) & ~(uintptr_t) 3) == 0)
    {
        fprintf (stderr, "setup failed\n");
        return 5;
    }
    const int width = api.simd_width (setup);
    const size_t frame = is_real ? (size_t) N : 2 * (size_t) N;

    // ---- hot: one frame, forward + backward in place with a work buffer (bench/bench.cpp:36-45)
    double hot_tps = 0;
    {
        float* x = (float*) api.amalloc (frame * sizeof (float));
        float* w = (float*) api.amalloc (frame * sizeof (float));
        std::mt19937 rng (1234);
        std::uniform_real_distribution<float> u (-1.f, 1.f);
        for (size_t i = 0; i < frame; ++i)
            x[i] = u (rng);
        const float inv = 1.0f / (float) N;
        size_t done = 0;
        const auto t0 = clk::now();
        do
        {
            for (int r = 0; r < 64; ++r)
            {
                api.transform (setup, x, x, w, 0);
                api.transform (setup, x, x, w, 1);
                for (size_t i = 0; i < frame; i += 64) // keep values bounded (1 element per cache line: negligible cost)
                    x[i] *= inv;
            }
            done += 128;
        } while (seconds_since (t0) < secs);
        hot_tps = (double) done / seconds_since (t0);
        api.afree (x);
        api.afree (w);
    }

    // ---- stream / allcore: distinct frames, out of place, forward then backward sweep
    auto stream_run = [&] (int nthreads, double& tps) {
        const size_t frames_total = std::max<size_t> ((size_t) nthreads, (stream_mib << 20) / (frame * sizeof (float)));
        const size_t per = frames_total / (size_t) nthreads;
        std::vector<float*> in ((size_t) nthreads), out ((size_t) nthreads), work ((size_t) nthreads);
        for (int t = 0; t < nthreads; ++t)
        {
            in[(size_t) t] = (float*) api.amalloc (per * frame * sizeof (float));
            out[(size_t) t] = (float*) api.amalloc (per * frame * sizeof (float));
            work[(size_t) t] = (float*) api.amalloc (frame * sizeof (float));
            std::mt19937 rng (1234u + (unsigned) t);
            std::uniform_real_distribution<float> u (-1.f, 1.f);
            for (size_t i = 0; i < per * frame; ++i)
                in[(size_t) t][i] = u (rng);
            memset (out[(size_t) t], 0, per * frame * sizeof (float));
        }
        std::atomic<size_t> total { 0 };
        std::atomic<bool> stop { false };
        const auto t0 = clk::now();
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back ([&, t] {
                size_t mine = 0;
                while (! stop.load (std::memory_order_relaxed))
                {
                    for (size_t f = 0; f < per && ! stop.load (std::memory_order_relaxed); ++f)
                    {
                        api.transform (setup, in[(size_t) t] + f * frame, out[(size_t) t] + f * frame, work[(size_t) t], 0);
                        ++mine;
                    }
                    for (size_t f = 0; f < per && ! stop.load (std::memory_order_relaxed); ++f)
                    {
                        api.transform (setup, out[(size_t) t] + f * frame, out[(size_t) t] + f * frame, work[(size_t) t], 1);
                        ++mine;
                    }
                }
                total += mine;
            });
        while (seconds_since (t0) < secs)
            std::this_thread::sleep_for (std::chrono::milliseconds (20));
        stop = true;
        for (auto& th : pool)
            th.join();
        tps = (double) total.load() / seconds_since (t0);
        for (int t = 0; t < nthreads; ++t)
        {
            api.afree (in[(size_t) t]);
            api.afree (out[(size_t) t]);
            api.afree (work[(size_t) t]);
        }
    };
    double stream_tps = 0, all_tps = 0;
    stream_run (1, stream_tps);
    stream_run (threads, all_tps);

    printf ("{\"N\": %d, \"transform\": \"%s\", \"simd_width_bytes\": %d, \"seconds_per_regime\": %.2f, "
            "\"hot_1t_transforms_per_s\": %.1f, \"stream_1t_transforms_per_s\": %.1f, "
            "\"allcore_transforms_per_s\": %.1f, \"allcore_threads\": %d, \"stream_mib\": %zu, "
            "\"host_hardware_threads\": %u, \"affinity_cpus\": %d, \"cgroup_cpu_quota\": %.2f, \"parallel_speedup_probe\": %.2f, \"probe_threads\": %d, \"cores_effective\": %d}\n",
            N, is_real ? "real" : "complex", width, secs, hot_tps, stream_tps, all_tps, threads, stream_mib,
            std::thread::hardware_concurrency(), aff, quota, probe, probe_threads, effective);
    api.destroy_setup (setup);
    dlclose (h);
    return 0;
}
