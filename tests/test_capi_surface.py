"""The drop-in boundary without a GPU: the shared library loads, exports every symbol the headers
declare (and nothing else public), the header is consumable from C11 and from C++ under its
chowdsp::fft namespace, and the host-side plan logic (size rule, pass split, byte sizes) is right.
No compute call is made here.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
sys.path.insert(0, ROOT)

from chowdsp_fft_amd import _lib, api  # noqa: E402

LEGACY = ["fft_bytes_required", "fft_new_setup", "fft_new_setup_preallocated", "fft_destroy_setup",
          "fft_simd_width_bytes", "fft_transform", "fft_transform_unordered", "fft_convolve_unordered",
          "fft_accumulate", "aligned_malloc", "aligned_free"]  # reference chowdsp_fft.h:81-163


def _declared(header):
    txt = open(os.path.join(INCLUDE, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return re.findall(r"\b([a-z_0-9]+)\s*\(", " ".join(l for l in txt.splitlines() if not l.strip().startswith("#")))


def test_headers_declare_expected_symbols():
    legacy = [n for n in _declared("chowdsp_fft.h") if n.startswith(("fft_", "aligned_"))]
    assert sorted(set(legacy)) == sorted(LEGACY)
    ext = [n for n in _declared("chowdsp_fft_gpu.h") if n.startswith("fft_")]
    assert set(ext) <= set(_lib.PROTOTYPES), set(ext) - set(_lib.PROTOTYPES)
    assert set(_lib.PROTOTYPES) == set(legacy) | set(ext)


def test_library_exports_every_declared_symbol_and_only_those():
    lib = _lib.lib()
    for name in _lib.PROTOTYPES:
        assert getattr(lib, name) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(_lib.PROTOTYPES) <= exported
    extra = {s for s in exported - set(_lib.PROTOTYPES) if not s.startswith(("_init", "_fini", "__"))}
    assert not extra, f"unexpected public symbols: {sorted(extra)[:8]}"


def test_enum_values_match_reference():
    # reference chowdsp_fft.h:64-75
    assert (_lib.FFT_FORWARD, _lib.FFT_BACKWARD, _lib.FFT_REAL, _lib.FFT_COMPLEX) == (0, 1, 0, 1)


def test_aligned_malloc_is_64_byte_aligned():
    lib = _lib.lib()
    for nbytes in (1, 63, 64, 4096, 1 << 20):
        p = lib.aligned_malloc(nbytes)
        assert p and p % 64 == 0  # reference chowdsp_fft.cpp:63-72
        C.memset(p, 0xAB, nbytes)
        lib.aligned_free(p)
    lib.aligned_free(None)  # reference chowdsp_fft.cpp:74-78: NULL is a no-op


def test_bytes_required_is_never_below_the_reference_figure():
    # reference simd/chowdsp_fft_impl_sse.cpp:67-72: 8N+96 (complex) / 4N+96 (real)
    for n in (16, 64, 4096, 65536):
        assert api.fft_bytes_required(n, api.FFT_COMPLEX) >= 8 * n + 96
        assert api.fft_bytes_required(n, api.FFT_REAL, False) >= 4 * n + 96
    assert api.fft_bytes_required(16, api.FFT_REAL) >= 128  # room for the plan header even for tiny N


def _smooth(n):
    for f in (2, 3, 5):
        while n % f == 0:
            n //= f
    return n == 1


def test_size_rule_host_logic():
    """the reference's rule (common.hpp:168-177, 216-225): N % 16 (complex) / N % 32 (real) and no prime factor other than
    2, 3, 5 -- accepted for EVERY such N up to a complex length of 2^21 (real: N <= 2^22)"""
    for n in range(1, 2000):
        assert api.size_supported(n, api.FFT_COMPLEX) == (n % 16 == 0 and _smooth(n)), n
        assert api.size_supported(n, api.FFT_REAL) == (n % 32 == 0 and _smooth(n)), n
    for n in (0, -16, 24, 112, 176, 7 << 18, 1 << 28, 3 << 27):
        assert not api.size_supported(n, api.FFT_COMPLEX)
    # beyond 2^21 complex points: N = R x C with both factors accepted lengths (two plans + transposes), up to 2^27 complex
    # points (real: N <= 2^25)
    for n in (3 << 21, 5 << 21, 1 << 24, 1 << 27, 15 << 20, (1 << 22) - 2048):
        assert api.size_supported(n, api.FFT_COMPLEX) == _smooth(n), n
        if not _smooth(n):
            continue
        d = api.plan_describe(n, api.FFT_COMPLEX)
        assert d["passes"] == 3 and d["n1"] * d["n2"] == n and max(d["n1"], d["n2"]) <= 1 << 21 and d["n1"] % 16 == 0 and d["n2"] % 16 == 0, (n, d)
        # what executes by default (option huge_columns = 1): strided columns x contiguous rows + one transpose, reported in the radix slots
        if d["radices"]:
            cols, rows = d["radices"]
            assert cols * rows == n and cols <= 2559 and rows <= 1 << 21 and rows % 16 == 0, (n, d)
    for n in (1 << 23, 3 << 21, 1 << 27):
        assert len(api.plan_describe(n, api.FFT_COMPLEX)["radices"]) == 2, n
    api.set_option("huge_columns", 0)
    assert api.plan_describe(1 << 23, api.FFT_COMPLEX)["radices"] == []
    api.set_option("huge_columns", 1)
    assert api.size_supported(1 << 25, api.FFT_REAL) and not api.size_supported(1 << 26, api.FFT_REAL)
    # ... except, up to 2^22, the products of two tile lengths with compile-time stage lists (at most 2048 points each): two passes
    for n, pair in ((1 << 22, (2048, 2048)), (2304000, (1440, 1600)), (3 << 20, (1536, 2048)), (2560000, (1600, 1600))):
        d = api.plan_describe(n, api.FFT_COMPLEX)
        assert d["passes"] == 2 and tuple(sorted((d["n1"], d["n2"]))) == pair, (n, d)
    # the reference's "Other sizes" test list (test/test.cpp:279-285)
    for n in (96, 192, 384, 480, 640, 768, 9216):
        assert api.size_supported(n, api.FFT_COMPLEX) and api.size_supported(n, api.FFT_REAL)
    assert api.size_supported(1 << 20, api.FFT_REAL) and api.size_supported(1 << 21, api.FFT_COMPLEX) and api.size_supported(1 << 22, api.FFT_REAL)
    # mixed-radix frames: one pass (in-place stages in one LDS image) up to a complex length of 20000, longer ones in two
    # passes: generic columns x power-of-two rows where N = (odd part * 2^j <= 1280, j >= 4) * (power of two in 64..1024)
    # exists, else generic columns x generic rows with both factors near sqrt(N)
    for n in (10240, 10368, 11520, 12288, 13824, 19200):
        assert api.size_supported(n, api.FFT_COMPLEX), n
        assert api.plan_describe(n, api.FFT_COMPLEX)["passes"] == 1
    assert api.size_supported(2 * 245760, api.FFT_REAL)
    assert api.plan_describe(24576, api.FFT_COMPLEX) == {"complex_len": 24576, "passes": 2, "n1": 512, "n2": 48, "radices": [8, 8, 8]}
    assert api.size_supported(46080, api.FFT_COMPLEX)  # 720 x 64, 8-wide column tiles
    d = api.plan_describe(20736, api.FFT_COMPLEX)  # 81 * 256: no power-of-two row length -> 144 x 144, both passes generic
    assert (d["passes"], d["n1"], d["n2"]) == (2, 144, 144)
    d = api.plan_describe(1 << 21, api.FFT_COMPLEX)
    assert (d["passes"], d["n1"] * d["n2"]) == (2, 1 << 21) and max(d["n1"], d["n2"]) <= 2048
    assert api.size_supported(2 * 9216, api.FFT_REAL)


def test_accepted_size_set_equals_the_reference_rule_up_to_2_21():
    """every N <= 2^21: accepted <=> N % 16 == 0 (complex) / N % 32 == 0 (real) and N = 2^a 3^b 5^c -- the reference's
    accepted set (the oracle's new_setup is spot-checked against the same rule in test_oracle_accepts_the_same_sizes)"""
    accepted = 0
    for n in range(16, (1 << 21) + 1, 16):
        want = _smooth(n)
        assert api.size_supported(n, api.FFT_COMPLEX) == want, n
        assert api.size_supported(n, api.FFT_REAL) == (want and n % 32 == 0), n
        accepted += want
        if want:
            d = api.plan_describe(n, api.FFT_COMPLEX)
            assert d is not None and d["complex_len"] == n, n
    assert accepted == 332


def test_oracle_accepts_the_same_sizes(oracle):
    import oracle_lib as ol

    for n in (16, 32, 48, 80, 112, 20736, 41472, 62208, 34992, 50000, 100000, 18 * 16, 22 * 16, 7 * 64, 1 << 17):
        for tr, atr in ((ol.COMPLEX, api.FFT_COMPLEX), (ol.REAL, api.FFT_REAL)):
            so = oracle.new_setup(n, tr)
            assert (so is not None) == api.size_supported(n, atr), (n, tr)
            if so is not None:
                oracle.destroy_setup(so)


def test_plan_describe_host_logic():
    d = api.plan_describe(4096, api.FFT_COMPLEX)
    assert d == {"complex_len": 4096, "passes": 1, "n1": 0, "n2": 0, "radices": [16, 16, 16]}
    d = api.plan_describe(1024, api.FFT_REAL)  # real N=1024 rides the 512-point complex core
    assert d["complex_len"] == 512 and d["passes"] == 1
    d = api.plan_describe(65536, api.FFT_COMPLEX)  # BASELINE config 5: two passes, 256 x 256
    assert (d["passes"], d["n1"], d["n2"]) == (2, 256, 256)
    for lg in range(4, 21):
        d = api.plan_describe(1 << lg, api.FFT_COMPLEX)
        prod = 1
        for r in d["radices"]:
            assert r in (2, 4, 8, 16, 32)  # 32: the split-plane single pass of N = 32768
            prod *= r
        assert prod == (d["n2"] if d["passes"] == 2 else d["complex_len"])
        if d["passes"] == 2:
            # (2^19 and 2^20: one radix-2 split around 1024-point rows, reported as rows of 2048)
            assert d["n1"] * d["n2"] == 1 << lg and 64 <= d["n2"] <= d["n1"] <= (2048 if lg >= 19 else 1024)
    assert api.plan_describe(24, api.FFT_COMPLEX) is None
    # mixed-radix sizes: one pass of the generic kernel; fewest stages, composite radices (6, 9, 10, 12, 15) merge the odd
    # factors into their neighbours, stages with an odd factor first
    d = api.plan_describe(480, api.FFT_COMPLEX)
    assert (d["complex_len"], d["passes"], d["n1"], d["n2"]) == (480, 1, 0, 0)
    # ... ordered so that the last stage stores whole 128-byte lines: runs of 480 / 6 = 80 points (tables.cpp: generic_radices)
    assert d["radices"][:3] == [10, 8, 6] and (480 // d["radices"][2]) % 16 == 0
    # ... 160, 192 and 240 points: three stages of radix <= 8 by measurement (round 5, profiles/r05_generic_lists.txt), not the fewest stages
    assert api.plan_describe(192, api.FFT_COMPLEX)["radices"][:3] == [6, 8, 4] and api.plan_describe(160, api.FFT_COMPLEX)["radices"][:3] == [4, 8, 5]
    assert api.plan_describe(240, api.FFT_COMPLEX)["radices"][:3] == [6, 8, 5] and api.plan_describe(144, api.FFT_COMPLEX)["radices"][:2] == [12, 12]
    assert api.plan_describe(9216, api.FFT_COMPLEX)["radices"] == [12, 8, 8, 12]  # (order by measurement, round 5: +11.6 % over 12 12 8 8)
    assert api.plan_describe(5120, api.FFT_COMPLEX)["radices"] == [10, 8, 8, 8]
    assert api.plan_describe(96, api.FFT_REAL)["complex_len"] == 48


def test_mixed_radix_plans_multiply_out():
    """every accepted single-pass mixed-radix length: the stage radices are ones the kernels have butterflies for and their
    product is the frame length (fft_gpu_plan_describe reports the first four stages)"""
    allowed = {2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16}
    checked = 0
    for n in range(16, 20001, 16):
        if not _smooth(n) or n & (n - 1) == 0:
            continue
        d = api.plan_describe(n, api.FFT_COMPLEX)
        assert d is not None and d["passes"] == 1, n
        prod = 1
        for r in d["radices"]:
            assert r in allowed, (n, d["radices"])
            prod *= r
        assert n % prod == 0 and (prod == n or len(d["radices"]) == 4), (n, d["radices"])
        checked += 1
    assert checked == 82  # the 2-3-5-smooth multiples of 16 up to 20000 that are not powers of two


def test_misuse_returns_codes_not_crashes():
    lib = _lib.lib()
    assert lib.fft_transform_batch(None, None, None, 4, 0, None) == -1  # FFT_GPU_EINVAL
    assert lib.fft_convolve_frames_batch(None, None, None, None, 1.0, 1, None) == -1
    assert lib.fft_gpu_setup_size(None) == -1
    lib.fft_destroy_setup(None)  # tolerated (the reference dereferences NULL here)
    lib.fft_transform(None, None, None, None, 0)
    assert isinstance(lib.fft_gpu_last_error(), bytes)


C_PROGRAM = r"""
#include "chowdsp_fft.h"
#include "chowdsp_fft_gpu.h"
#include <stdio.h>
#include <string.h>
int main (void)
{
    float* x = (float*) aligned_malloc (sizeof (float) * 64 * 2);
    if (! x || ((size_t) x) % 64) return 2;
    memset (x, 0, sizeof (float) * 128);
    x[2] = 1.0f; /* unit impulse at n = 1 */
    if (fft_bytes_required (64, FFT_COMPLEX, false) < 8 * 64 + 96) return 3;
    if (fft_new_setup (24, FFT_COMPLEX, true) != NULL) return 4; /* unsupported size -> NULL */
    void* s = fft_new_setup (64, FFT_COMPLEX, false);
    if (s != NULL) /* only with a usable GPU */
    {
        if (fft_simd_width_bytes (s) != 16) return 5;
        fft_transform (s, x, x, NULL, FFT_FORWARD);
        /* |X[k]| = 1 for an impulse */
        for (int k = 0; k < 64; ++k)
        {
            float m = x[2 * k] * x[2 * k] + x[2 * k + 1] * x[2 * k + 1];
            if (m < 0.9999f || m > 1.0001f) return 6;
        }
        fft_destroy_setup (s);
        puts ("gpu-ok");
    }
    else
        puts ("no-gpu");
    aligned_free (x);
    return fft_gpu_size_supported (4096, FFT_COMPLEX) == 1 ? 0 : 7;
}
"""

CPP_PROGRAM = r"""
#include "chowdsp_fft.h"
#include "chowdsp_fft_gpu.h"
int main()
{
    // C++ callers use the namespace and the `= true` default argument (reference chowdsp_fft.h:82-85)
    auto bytes = chowdsp::fft::fft_bytes_required (1024, chowdsp::fft::FFT_REAL);
    void* p = chowdsp::fft::aligned_malloc (bytes);
    void* s = chowdsp::fft::fft_new_setup_preallocated (1024, chowdsp::fft::FFT_REAL, p);
    (void) s; // NULL without a GPU; caller-owned memory either way
    chowdsp::fft::aligned_free (p);
    return chowdsp::fft::fft_gpu_size_supported (1024, chowdsp::fft::FFT_REAL) == 1 ? 0 : 1;
}
"""


@pytest.mark.parametrize("lang,compiler,std,src", [("c", "gcc", "-std=c11", C_PROGRAM), ("cpp", "g++", "-std=c++17", CPP_PROGRAM)])
def test_header_is_consumable_and_links(tmp_path, lang, compiler, std, src):
    """counterpart of the reference's test/test.c: the header compiles as C11 and as C++, and a program
    linked against the drop-in library runs (with or without a GPU)"""
    f = tmp_path / f"smoke.{lang}"
    f.write_text(src)
    exe = tmp_path / f"smoke_{lang}"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [compiler, std, "-Wall", "-Werror", "-I", INCLUDE, str(f), "-o", str(exe), "-L", libdir,
           "-lchowdsp_fft_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_round3_planner_rules_host_logic():
    """Planner rules of round 3, through fft_gpu_plan_describe (pure host logic).
    (a) single-pass mixed-radix plans: where some order of the plan's radices lets the last stage store whole 128-byte lines
        (runs of n / r_last points, a multiple of 16), the plan has such an order (tables.cpp: generic_radices);
    (b) long plans: the pair of tile lengths and which of them is the column length n2 (tables.cpp: split_static_pair)."""
    for n in (160, 192, 240, 288, 320, 384, 400, 432, 480, 576, 640, 720, 768, 800, 864, 960, 1152, 1280, 1536, 1600, 1920, 2000):
        d = api.plan_describe(n, api.FFT_COMPLEX)
        rad = [r for r in d["radices"] if r]
        prod = 1
        for r in rad:
            prod *= r
        assert d["passes"] == 1 and prod == n, (n, d)
        if any((n // r) % 16 == 0 for r in rad):
            assert (n // rad[-1]) % 16 == 0, (n, rad)
    for n, want in ((34992, (243, 144)), (100000, (250, 400)), (48000, (240, 200)), (46080, (240, 192)), (38400, (240, 160)),
                    (20736, (144, 144)), (30000, (375, 80)), (1 << 22, (2048, 2048)), (245760, (1024, 240))):
        d = api.plan_describe(n, api.FFT_COMPLEX)
        assert d["passes"] == 2 and (d["n1"], d["n2"]) == want, (n, d)


def test_generated_inplace_lists_are_the_planners():
    """chowdsp_fft_amd/csrc/generic_static_inplace.hpp (tools/gen_static_inplace.py) covers EVERY 2-3-5-smooth multiple of 16 in
    2048 < n <= 20000 that is not a power of two, each with the stage list the planner gives the length today (a list that
    drifted would silently fall back to the run-time list at launch)."""
    import re

    from chowdsp_fft_amd import api

    path = os.path.join(ROOT, "chowdsp_fft_amd", "csrc", "generic_static_inplace.hpp")
    rows = {}
    for m in re.finditer(r"X \(([0-9, ]+)\)", open(path).read()):
        v = [int(t) for t in m.group(1).split(",")]
        rows[v[0]] = (v[1], v[2:])
    want = sorted(2 ** a * 3 ** b * 5 ** c for a in range(16) for b in range(10) for c in range(8)
                  if 2048 < 2 ** a * 3 ** b * 5 ** c <= 20000 and (2 ** a * 3 ** b * 5 ** c) % 16 == 0 and not (b == 0 and c == 0))
    assert sorted(rows) == want and len(want) == 52
    for n, (block, radices) in rows.items():
        d = api.plan_describe(n, api.FFT_COMPLEX)
        assert d["complex_len"] == n and d["passes"] == 1
        shown = d["radices"]
        assert radices[:len(shown)] == shown, f"n = {n}: header {radices}, planner {shown}"
        prod = 1
        for r in radices:
            prod *= r
        assert prod == n
        assert block == (256 if n <= 2592 else (512 if n <= 10240 else 1024))  # (tables.hpp generic_geometry; 512 from 2880 points on since round 5)


def test_developer_options_are_locked_without_the_environment_switch():
    """"group_spin_limit" can make every one-launch transform of the process time out on purpose: only reachable with
    CHOWDSP_FFT_DEVELOPER=1 (the test suite sets it, tests/conftest.py); resetting it to 0 is always allowed"""
    import subprocess
    import sys

    code = ("from chowdsp_fft_amd import api\n"
            "api.set_option('group_spin_limit', 0)\n"
            "try:\n"
            "    api.set_option('group_spin_limit', 1)\n"
            "    print('accepted')\n"
            "except api.FFTError:\n"
            "    print('refused')\n")
    for dev, want in (("0", "refused"), ("1", "accepted")):
        env = dict(os.environ, CHOWDSP_FFT_DEVELOPER=dev)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith(want), (dev, r.stdout, r.stderr[-500:])
