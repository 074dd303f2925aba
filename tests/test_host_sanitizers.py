"""Sanitizer build of the library's HOST side, on the CPU (the GPU pool offers no device sanitizers): the seven .cpp units (C ABI,
planner, device-table cache, engine, host streaming, adapters, multi-GPU) compiled with AddressSanitizer +
UndefinedBehaviorSanitizer (`make -C chowdsp_fft_amd/csrc san` -> chowdsp_fft_amd/san/libchowdsp_fft_hip_san.so, linked with the
regular kernel objects) and driven without a device:

* tests/c/host_misuse_driver.c -- the size rule and the planner's description for every length up to 70000 and a list of long /
  huge ones, byte counts, aligned allocation, options, every entry point with NULL and with a handle that is not a plan, the
  plan constructors without a device, the shard arithmetic, the deferred-call window;
* a planner probe -- build_host_tables (csrc/tables.cpp: the stage lists, twiddle tables and inter-pass tables of every plan
  kind) for every accepted length up to 2^17 and the long lengths of every family, both transforms.

Leak detection is on: the host side frees what it allocates on every refused call.
(What the reference does at this level: simd/chowdsp_fft_impl_common.hpp:168-214 new_setup / decompose, no sanitizer
configuration of its own -- SURVEY section 5.)
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "chowdsp_fft_amd", "csrc")
SAN = os.path.join(ROOT, "chowdsp_fft_amd", "san")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")

PROBE = r"""
#include "tables.hpp"
#include <cstdio>
#include <cstdlib>
int main (int argc, char** argv)
{
    using namespace cfft::host;
    long built = 0;
    size_t bytes = 0;
    const int top = argc > 1 ? std::atoi (argv[1]) : (1 << 17);
    const auto one = [&] (int n, int tr) {
        HostTables t;
        if (build_host_tables (n, tr, t))
        {
            ++built;
            bytes += (t.stage_twiddles.size() + t.real_twiddles.size() + t.big_twiddles.size()) * sizeof (float2);
        }
    };
    for (int tr = 0; tr < 2; ++tr)
        for (int n = 1; n <= top; ++n)
            one (n, tr);
    static const int longs[] = { 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 245760, 655360, 1310720, 1327104, 153600, 184320, 3 << 19, 5 << 18 };
    for (int n : longs)
        for (int tr = 0; tr < 2; ++tr)
            one (n, tr);
    for (int mode = 0; mode <= 2; ++mode) // the real four-step's split modes (option "real4_generic")
    {
        set_real4_generic (mode);
        for (int n : { 41472, 96000, 122880, 245760 })
            one (n, 0);
    }
    for (int rows : { 2, 16, 144, 600, 1080, 4096 })
    {
        HostTables t;
        built += build_column_tables (rows, t) ? 1 : 0;
    }
    std::printf ("%ld plans, %zu table bytes\n", built, bytes);
    return built > 100 ? 0 : 1;
}
"""


@pytest.fixture(scope="module")
def san_lib():
    if shutil.which("hipcc") is None or not os.path.exists(CLANG):
        pytest.skip("needs the ROCm compilers (hipcc + its clang with the sanitizer runtimes)")
    r = subprocess.run(["make", "-C", CSRC, "san", "-j8"], capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lib = os.path.join(SAN, "libchowdsp_fft_hip_san.so")
    assert os.path.exists(lib)
    return lib


def _clean(r):
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    for word in ("AddressSanitizer", "LeakSanitizer", "runtime error", "UndefinedBehaviorSanitizer"):
        assert word not in out, out[-4000:]
    return out


def test_c_abi_walk_without_a_device_under_asan_ubsan(san_lib, tmp_path):
    exe = str(tmp_path / "driver")
    r = subprocess.run([CLANG, "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "c", "host_misuse_driver.c"), "-o", exe,
                        "-L", SAN, "-lchowdsp_fft_hip_san", "-Wl,-rpath," + SAN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _clean(subprocess.run([exe, "70000"], capture_output=True, text=True, timeout=900, env=ENV))
    assert "accepted lengths up to 70000" in out


def test_planner_tables_under_asan_ubsan(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc for the host compile")
    src, exe = str(tmp_path / "probe.cpp"), str(tmp_path / "probe")
    open(src, "w").write(PROBE)
    r = subprocess.run(["hipcc", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-gpu-sanitize", "-I", CSRC, src, os.path.join(CSRC, "tables.cpp"), "-o", exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _clean(subprocess.run([exe], capture_output=True, text=True, timeout=1500, env=ENV))
    assert "plans" in out
