#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats and HBM counters.
# usage (on the GPU box, from the repo root): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp

echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -1 "$OUT/smoke.log"

echo "== bench (default workload)"
timeout 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench_c2c4096.json" 2> "$OUT/bench_c2c4096.err"
cat "$OUT/bench_c2c4096.json"
for W in real_n1024_b262144_fwd_inv conv_real_n4096_b131072 c2c_n65536_b8192_fwd_inv ols_engine_b2048_p16_c8192; do
  timeout 600 python bench.py --steps 20 --warmup 3 --workload $W --no-cpu-baseline > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err"
  python - "$OUT/bench_$W.json" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"], "value", round(d["value"], 5), d["unit"], "ms/step", round(d["ms_per_step"], 3), "roofline", round(d["roofline"]["achieved"]), "GB/s frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print("bench parse failed", sys.argv[1], e)
EOF
done

echo "== legacy host-pointer call rate (PCIe inclusive)"
timeout 300 python tools/legacy_rate.py 2>/dev/null | tee "$OUT/legacy_rate.txt"

echo "== rocprofv3 kernel stats (same command as the bench line)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_stats.log" 2>&1
cd "$REPO"
find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-220

echo "== rocprofv3 HBM counters (separate passes; FETCH_SIZE and WRITE_SIZE do not fit one pass)"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/prof_pmc_$C" -- python3 "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/prof_pmc_$C.log" 2>&1
  # calibration of the counter on a known byte count in the same access widths
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/prof_cal_$C" -- "$REPO/tools/membench" 16384 > "$OUT/prof_cal_$C.log" 2>&1
done
cd "$REPO"
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1
cat "$OUT/pmc_summary.txt"
echo "== done"
