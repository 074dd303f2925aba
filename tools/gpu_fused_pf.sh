#!/bin/bash
# one-launch two-pass form: parity first (short timeout: a broken queue ends by its bounded spins), then a lag / workgroup sweep.
# Written for the operand-prefetch experiment of session r02j (commit 06d0325, reverted; CHOWDSP_FFT_FUSED_PREFETCH exists only
# there: with the current library the prefetch=0 and prefetch=1 rows are the same kernel); profiles/r02_fused_prefetch_experiment.txt
OUT=$PWD/gpurun_out/fused_pf
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 240 python -m pytest tests/test_gpu_fused_two_pass.py -q -m gpu -x > "$OUT/pytest_fused.log" 2>&1 || { tail -15 "$OUT/pytest_fused.log"; exit 1; }
tail -2 "$OUT/pytest_fused.log"
W=c2c_n65536_b8192_fwd_inv
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 90 python bench.py --workload ${WL:-$W} --no-cpu-baseline --no-secondary --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'ms/step', round(d['ms_per_step'],3), 'GB/s', round(d['roofline']['achieved']), 'frac', round(d['roofline']['frac'],3))" | tee -a "$OUT/timing.txt"
}
: > "$OUT/timing.txt"
run two-launch CHOWDSP_FFT_FUSED=0 || exit 1
WL=${W}_one_launch
for PF in 0 1; do
  for LAG in 2 3 4; do
    run "fused prefetch=$PF lag=$LAG wg/cu=2" CHOWDSP_FFT_FUSED=1 CHOWDSP_FFT_FUSED_PREFETCH=$PF CHOWDSP_FFT_FUSED_LAG=$LAG || exit 1
  done
done
run "fused prefetch=1 lag=3 wg/cu=3" CHOWDSP_FFT_FUSED=1 CHOWDSP_FFT_FUSED_LAG=3 CHOWDSP_FFT_FUSED_WG_PER_CU=3
run "fused prefetch=1 lag=2 wg/cu=1" CHOWDSP_FFT_FUSED=1 CHOWDSP_FFT_FUSED_LAG=2 CHOWDSP_FFT_FUSED_WG_PER_CU=1
