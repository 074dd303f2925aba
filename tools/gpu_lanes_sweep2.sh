#!/bin/bash
# one lane: workspace cap x real chunk divisor (all read once per process).  Output: gpurun_out/ws/lanes_sweep2.txt
mkdir -p gpurun_out/ws
OUT=gpurun_out/ws/lanes_sweep2.txt
: > $OUT
W=${1:-c524288,c1048576,r262144,r524288,r1048576,c20736,c46080,r41472,r96000,r32768,r65536}
for div in 2 1; do
  for mib in 192 256 320 384; do
    echo "## CHOWDSP_FFT_LANES=1 CHOWDSP_FFT_WORKSPACE_MIB=$mib CHOWDSP_FFT_REAL4_CHUNK_DIV=$div CHOWDSP_FFT_PAIR_CHUNK_DIV=$div" >> $OUT
    CHOWDSP_FFT_LANES=1 CHOWDSP_FFT_WORKSPACE_MIB=$mib CHOWDSP_FFT_REAL4_CHUNK_DIV=$div CHOWDSP_FFT_PAIR_CHUNK_DIV=$div timeout -k 10 300 python tools/ab_env.py --env DUMMY=0 --rounds 3 --iters 15 --workloads $W 2>&1 | grep -v amdgpu.ids | grep -A1 "^---" | grep -v "^--$" | paste - - | awk '{print $2, $(NF-3), $(NF-2), $(NF-1), $NF}' >> $OUT || exit 1
  done
done
cat $OUT
