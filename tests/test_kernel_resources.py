"""Compile-time hygiene of the HIP kernels (no GPU needed: hipcc cross-compiles gfx950).

Every kernel must keep its working set in registers: a non-zero scratch size means the compiler spilled
or demoted an array to private memory (this happened once with a by-reference lambda, costing 2x), and
the hot kernels must stay within the occupancy their launch geometry assumes.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chowdsp_fft_amd", "csrc")
UNITS = {"kernels_c2c": 0, "kernels_real": 0, "kernels_conv": 0, "kernels_multipass": 1, "kernels_generic": 0, "kernels_split": 0,
         "kernels_tile_static_d": 1, "kernels_group_real": 0, "kernels_group_conv": 0}  # -> CFFT_PACKED as in the Makefile (of the four tile_static units the one with the longest frames)


def _resource_usage(unit, packed):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffp-contract=fast",
           f"-DCFFT_PACKED={packed}", "-c", os.path.join(CSRC, unit + ".hip"), "-o", os.devnull,
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            kernels.append(cur)
            continue
        for key, tag in (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "occ")):
            m = re.search(r" " + re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[tag] = int(m.group(1))
    return kernels


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
@pytest.mark.parametrize("unit", sorted(UNITS))
def test_no_scratch_and_bounded_registers(unit):
    kernels = _resource_usage(unit, UNITS[unit])
    assert len(kernels) >= 4 or unit in ("kernels_split", "kernels_group_real", "kernels_group_conv")
    for k in kernels:
        assert k.get("scratch", 0) == 0, f"{k['name']} uses {k['scratch']} B/lane of scratch"
        # groupfast_kernel (group_fast_kernels.hpp) is launched one 256-work-item workgroup per CU by design (__launch_bounds__ (256, 1):
        # one wave per SIMD owns the unified 512-entry file); every other kernel must leave room for a second wave
        limit = 512 if "groupfast_kernel" in k["name"] else 256
        assert k.get("vgpr", 0) + k.get("agpr", 0) <= limit, f"{k['name']}: {k.get('vgpr')} VGPR + {k.get('agpr')} AGPR"


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_headline_kernel_occupancy():
    """N=4096 c2c (software-pipelined): <= 256 registers -> two 256-thread workgroups (8 waves) per CU"""
    ks = [k for k in _resource_usage("kernels_c2c", 0) if "c2c_kernel_pf" in k["name"] and "Li4096ELi256" in k["name"]]
    assert len(ks) == 2
    for k in ks:
        assert k["occ"] >= 2 and k["vgpr"] <= 256
