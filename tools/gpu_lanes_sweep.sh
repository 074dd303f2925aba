#!/bin/bash
# lanes x workspace cap for the chunked two-pass plans (both read once per process: one process per cell).  Output: gpurun_out/ws/lanes_sweep.txt
mkdir -p gpurun_out/ws
OUT=gpurun_out/ws/lanes_sweep.txt
: > $OUT
W=${1:-c131072,c262144,c524288,c1048576,r131072,r262144,r524288,r1048576,c20736,c46080,r41472,r96000}
for lanes in 1 2; do
  for mib in 64 128 256 512; do
    echo "## CHOWDSP_FFT_LANES=$lanes CHOWDSP_FFT_WORKSPACE_MIB=$mib" >> $OUT
    CHOWDSP_FFT_LANES=$lanes CHOWDSP_FFT_WORKSPACE_MIB=$mib timeout -k 10 300 python tools/ab_env.py --env DUMMY=0 --rounds 3 --iters 15 --workloads $W 2>&1 | grep -v amdgpu.ids | grep -A1 "^---" | grep -v "^--$" | paste - - | awk '{print $2, $(NF-3), $(NF-2), $(NF-1), $NF}' >> $OUT || exit 1
  done
done
cat $OUT
