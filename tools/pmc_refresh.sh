#!/bin/bash
# ONE command that regenerates every entry of profiles/pmc_traffic.json (run on the GPU box; copy the result into profiles/):
#   tools/pmc_refresh.sh [tag] ["where / when"]   ->  gpurun_out/<tag>/pmc_traffic.json + the per-workload summaries
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (they do not fit one pass), kernel trace only.
TAG=${1:-pmc}; NOTE=${2:-"tools/pmc_refresh.sh"}
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
STEPS=3; WARMUP=1
for W in c2c_n4096_b65536_fwd_inv real_n1024_b262144_fwd_inv conv_real_n4096_b131072 c2c_n65536_b8192_fwd_inv c2c_n65536_b8192_fwd_inv_two_launches; do
  for C in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$W/prof_pmc_$C" -- python3 "$REPO/bench.py" --steps $STEPS --warmup $WARMUP --prewarm-seconds 0 --no-cpu-baseline --no-secondary --workload $W > "$OUT/${W}_$C.log" 2>&1 || { echo "pmc pass failed: $W $C"; tail -5 "$OUT/${W}_$C.log"; exit 1; }
    cd "$REPO"
  done
done
python tools/pmc_refresh.py "$OUT" $STEPS $WARMUP "$OUT/pmc_traffic.json" "$NOTE" | tee "$OUT/pmc_traffic_summary.txt"
