"""The two-lane form of the chunked two-pass plans (CHOWDSP_FFT_LANES=2: two internal streams, two workspaces, half grids, events
to fork from / join back into the caller's stream -- the default of rounds 1-4, kept as a switch: profiles/r05_lanes_sweep.txt)
is read once per process, so it is exercised in a CHILD pytest process: the chunking-invariance cases and the graph-capture
cases run again under it.

Like tests/test_bench_two_ranks_gpu.py this file sorts before every test that makes a HIP call in the pytest process (a process
that has initialised the GPU must not fork + exec on the GPU pool).
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.gpu
def test_two_lane_form_in_a_child_process():
    env = dict(os.environ, CHOWDSP_FFT_LANES="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(HERE, "test_gpu_chunking.py"), os.path.join(HERE, "test_gpu_graph_capture.py"),
           os.path.join(HERE, "test_gpu_other_sizes.py") + "::test_workspace_bytes_reports_what_is_allocated"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-1000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
