#!/bin/bash
# round 2: the one-launch (XCD-local) two-pass path for N = 65536: parity, then timing against the two-launch form
OUT=$PWD/gpurun_out/r02b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python -m pytest tests -q -m gpu -x -k "65536 or two_pass or fullsize or threads or strided" > "$OUT/pytest_fused.log" 2>&1
tail -5 "$OUT/pytest_fused.log"
W=c2c_n65536_b8192_fwd_inv
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 60 python bench.py --workload $W --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'ms/step', round(d['ms_per_step'],3), 'GB/s', round(d['roofline']['achieved']), 'frac', round(d['roofline']['frac'],3))" | tee -a "$OUT/timing.txt"
}
run two-launch CHOWDSP_FFT_FUSED=0
for WG in 2 3 4; do
  for LAG in 2 3 4 5; do
    run "fused lag=$LAG wg/cu=$WG" CHOWDSP_FFT_FUSED_LAG=$LAG CHOWDSP_FFT_FUSED_WG_PER_CU=$WG
  done
done
