#!/bin/bash
set -o pipefail
REPO=$PWD; OUT=$REPO/gpurun_out/r3; mkdir -p $OUT
for PF in 0 1 2; do
echo "== prefetch $PF"; CHOWDSP_FFT_GROUP_PREFETCH=$PF timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
done
echo "== prefetch 1, 2.5 per CU"; CHOWDSP_FFT_GROUP_PREFETCH=1 CHOWDSP_FFT_GROUP_WG_PER_CU=2.5 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
echo "== prefetch 0, 2.5 per CU"; CHOWDSP_FFT_GROUP_PREFETCH=0 CHOWDSP_FFT_GROUP_WG_PER_CU=2.5 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
export CHOWDSP_FFT_LIB=$REPO/chowdsp_fft_amd/variants/lib_gexp.so
for PF in 1 2; do
echo "== stamps (dbg 32) prefetch $PF"
CHOWDSP_FFT_GROUP_PREFETCH=$PF CHOWDSP_FFT_GROUP_DBG=32 CHOWDSP_FFT_FUSED_DEBUG=1 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep "stamps\|form" | tail -9
done
