#!/bin/bash
# round 3: the static-group one-launch form of N = 65536 -- parity tests, then the three forms side by side
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_fused_two_pass.py -x -q > gpurun_out/r3/fused_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r3/fused_tests.log
[ $rc -ne 0 ] && exit $rc
CHOWDSP_FFT_FUSED_DEBUG=1 timeout -k 10 120 python tools/group_time.py 512 2 > gpurun_out/r3/group_debug.txt 2>&1
timeout -k 10 300 python tools/group_time.py 8192 0,1,2 > gpurun_out/r3/group_time.txt 2>&1 || exit 1
cat gpurun_out/r3/group_time.txt
CHOWDSP_FFT_GROUP_WG_PER_CU=3 timeout -k 10 300 python tools/group_time.py 8192 2 > gpurun_out/r3/group_time_wpc3.txt 2>&1 || exit 1
cat gpurun_out/r3/group_time_wpc3.txt
CHOWDSP_FFT_GROUP_WG_PER_CU=1 timeout -k 10 300 python tools/group_time.py 8192 2 > gpurun_out/r3/group_time_wpc1.txt 2>&1 || exit 1
cat gpurun_out/r3/group_time_wpc1.txt
