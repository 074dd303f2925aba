#!/bin/bash
set -o pipefail
REPO=$PWD
for PF in 0 3 0 3; do
echo "== prefetch $PF"; CHOWDSP_FFT_GROUP_PREFETCH=$PF timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
done
export CHOWDSP_FFT_LIB=$REPO/chowdsp_fft_amd/variants/lib_gexp.so
echo "== stamps (dbg 32) prefetch 3"
CHOWDSP_FFT_GROUP_PREFETCH=3 CHOWDSP_FFT_GROUP_DBG=32 CHOWDSP_FFT_FUSED_DEBUG=1 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep "stamps\|form" | tail -9
