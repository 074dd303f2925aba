#!/bin/bash
# How do the BASELINE kernels respond to fewer persistent workgroups in flight?  (CHOWDSP_FFT_GRID_PERCENT: developer knob, launch_util.hpp)
OUT=$PWD/gpurun_out/grid_sweep; mkdir -p "$OUT"; : > "$OUT/sweep.txt"
for W in c2c_n4096_b65536_fwd_inv real_n1024_b262144_fwd_inv conv_real_n4096_b131072; do
  for P in 100 88 75 63 50 100; do
    CHOWDSP_FFT_GRID_PERCENT=$P timeout -k 10 90 python bench.py --workload $W --no-cpu-baseline --no-secondary --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$W grid $P %', 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a "$OUT/sweep.txt"
  done
done
