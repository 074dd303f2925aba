#!/bin/bash
# Round-2 evidence session on the GPU box: parity suite, smoke, the default bench line, rocprofv3 kernel stats of the same
# command, HBM counters (separate --pmc passes) for every BASELINE configuration, all-sizes rate table.
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp

echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
echo "== smoke"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/smoke.log"
echo "== bench (default: headline + secondary configs + cpu baseline)"
timeout 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], "value", round(d["value"], 5), "ms/step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3))
for s in d.get("secondary", []):
    print(" ", s["workload"], round(s["value"], 5), "ms", round(s["ms_per_step"], 3), "frac", round(s["roofline"]["frac"], 3))
cb = d.get("cpu_baseline", {})
print("  cpu:", cb.get("value"), "1t hot;", cb.get("allcore_GFFT_s"), "on", cb.get("allcore_threads"), "threads; cores_effective", cb.get("cores_effective"))
PY
echo "== rocprofv3 kernel stats of the bench command"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_stats.log" 2>&1
cd "$REPO"
find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1 | xargs -r cat > "$OUT/kernel_stats_bench_default.csv"
head -12 "$OUT/kernel_stats_bench_default.csv" | cut -c1-200
echo "== rocprofv3 HBM counters per BASELINE configuration (FETCH_SIZE and WRITE_SIZE in separate passes)"
for W in c2c_n4096_b65536_fwd_inv real_n1024_b262144_fwd_inv conv_real_n4096_b131072 c2c_n65536_b8192_fwd_inv; do
  for C in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_${W}/prof_pmc_$C" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > "$OUT/pmc_${W}_$C.log" 2>&1
    cd "$REPO"
  done
  echo "-- $W"
  python tools/pmc_summary.py "$OUT/pmc_${W}" | grep -E "cfft|---"
done
echo "== all-sizes rate table"
timeout 900 python tools/all_sizes_rate.py > "$OUT/all_sizes.txt" 2>/dev/null || true
tail -5 "$OUT/all_sizes.txt"
echo "== JUCE-engine adapter: fused layouts vs the two-launch form"
{ echo "# fused (default)"; timeout 200 python tools/juce_rate.py 5 7 9 10 12 14; echo "# CHOWDSP_FFT_JUCE_FUSED=0 (two launches)"; CHOWDSP_FFT_JUCE_FUSED=0 timeout 200 python tools/juce_rate.py 5 7 9 10 12 14; } > "$OUT/juce_rates.txt" 2>/dev/null || true
tail -4 "$OUT/juce_rates.txt"
echo "== done"
