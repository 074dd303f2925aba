"""A compiled C11 caller of the drop-in library, run on the GPU (the reference's test/test.c:31-178 end to end).

The reference proves its C ABI with a C program that runs complex and real transforms for 2^5 .. 2^19 against
upstream pffft.  pffft is not available offline, so the comparator here is the reference's SSE path itself: the
committed golden vectors where they hold the size (tests/golden/reference_sse.npz, generated from the compiled
reference), otherwise the oracle restatement (bit-identical to that path, tests/test_oracle_vs_reference.py).
This test dumps the vectors to raw float32 files, compiles tests/c/reference_sequence.c with gcc -std=c11 against
include/ and libchowdsp_fft_hip.so, runs it and requires the final "gpu-ok".

The file name sorts first in tests/ and the gpu-marked test makes no HIP call in the pytest process (no `gpu`
fixture): the C program is started as a child BEFORE this process has initialised the GPU (a process that has must
not fork + exec on the GPU pool).
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from chowdsp_fft_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCE = os.path.join(HERE, "c", "reference_sequence.c")


def build_program(tmp_path):
    exe = tmp_path / "reference_sequence"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", INCLUDE, SOURCE, "-o", str(exe), "-L", libdir,
           "-lchowdsp_fft_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def write_case(path, n, is_real, x, fwd, bwd):
    with open(path, "wb") as f:
        np.array([n, 1 if is_real else 0], dtype="<i4").tofile(f)
        for a in (x, fwd, bwd):
            np.ascontiguousarray(a, dtype="<f4").tofile(f)


def test_c_program_compiles_without_a_gpu(tmp_path):
    """CPU leg: the program builds -Wall -Werror as C11 against the public headers and reports the missing device"""
    exe = build_program(tmp_path)
    if _lib.lib().fft_gpu_device_count() >= 1:
        pytest.skip("a GPU is visible: the gpu-marked test below runs the program for real")
    r = subprocess.run([str(exe), "/nonexistent"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_reference_c_test_sequence_on_gpu(oracle, golden, tmp_path):
    exe = build_program(tmp_path)
    cases = []
    # the reference's loop: 2^5 .. 2^19, complex and real (test.c:141-151), tone input (test.c:38-42, 84-87)
    for lg in range(5, 20):
        n = 1 << lg
        for is_real in (False, True):
            tag = "r" if is_real else "c"
            key = f"{tag}{n}_tone"
            if f"{key}_fwd" in golden:  # outputs of the compiled reference itself
                x, fwd, bwd = golden[f"{key}_in"], golden[f"{key}_fwd"], golden[f"{key}_bwd"]
            else:
                x = ol.tone(n, not is_real)
                so = oracle.new_setup(n, ol.REAL if is_real else ol.COMPLEX)
                fwd = oracle.transform(so, x, ol.FORWARD)
                bwd = oracle.transform(so, fwd, ol.BACKWARD)
                oracle.destroy_setup(so)
            p = tmp_path / f"{tag}{n}.bin"
            write_case(p, n, is_real, x, fwd, bwd)
            cases.append(str(p))
    # the reference's "Other sizes" that the golden file holds (radix-3 / radix-5 passes)
    for n in (96, 480):
        for is_real in (False, True):
            tag = "r" if is_real else "c"
            key = f"{tag}{n}_tone"
            p = tmp_path / f"{tag}{n}.bin"
            write_case(p, n, is_real, golden[f"{key}_in"], golden[f"{key}_fwd"], golden[f"{key}_bwd"])
            cases.append(str(p))
    r = subprocess.run([str(exe)] + cases, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-2000:])
    assert r.stdout.rstrip().endswith("gpu-ok"), r.stdout[-2000:]
    assert sum(1 for line in r.stdout.splitlines() if line.startswith("ok ")) == 2 * len(cases)  # plain + pre-allocated
