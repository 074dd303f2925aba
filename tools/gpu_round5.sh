#!/bin/bash
# Round-5 evidence session on the GPU box: parity suite, smoke, the default bench line, rocprofv3 kernel stats of the same
# command, PMC traffic of every bench workload (tools/pmc_refresh.sh: profiles/pmc_traffic.json with the source hash it was
# taken on), all-sizes rate table, fused-convolution rates.
TAG=${1:-r05}
STAGE=${2:-all}   # a = tests, smoke, bench, rocprofv3 stats, PMC traffic; b = the rate tables; all = both (may exceed one gpurun call)
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp

if [ "$STAGE" != "b" ]; then
echo "== pytest -m gpu"
timeout -k 10 1200 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
echo "== smoke"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/smoke.log"
echo "== bench (default: headline + secondary configs + cpu baseline)"
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["config"]["workload"], "value", round(d["value"], 5), "ms/step", round(d["ms_per_step"], 3), "frac", round(r["frac"], 3), "traffic", r.get("traffic"), "stale", r.get("traffic_stale"))
for s in d.get("secondary", []):
    print(" ", s["workload"], round(s["value"], 5), "ms", round(s["ms_per_step"], 3), "frac", round(s["roofline"]["frac"], 3), "traffic", s["roofline"].get("traffic"))
cb = d.get("cpu_baseline", {})
print("  cpu:", cb.get("value"), "1t hot;", cb.get("allcore_GFFT_s"), "on", cb.get("allcore_threads"), "threads; cores_effective", cb.get("cores_effective"))
PY
echo "== rocprofv3 kernel stats of the bench command"
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_stats.log" 2>&1
cd "$REPO"
find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1 | xargs -r cat > "$OUT/kernel_stats_bench_default.csv"
head -12 "$OUT/kernel_stats_bench_default.csv" | cut -c1-200
echo "== PMC traffic of every bench workload"
bash tools/pmc_refresh.sh "$TAG/pmc" "round 5, MI355X gpurun session $TAG (tools/gpu_round5.sh -> tools/pmc_refresh.sh)"
PMC_RC=$?
echo "== bench again: the traffic figures now belong to these sources"
if [ $PMC_RC -eq 0 ] && [ -s "$OUT/pmc/pmc_traffic.json" ]; then cp "$OUT/pmc/pmc_traffic.json" profiles/pmc_traffic.json; else echo "pmc_refresh failed (rc $PMC_RC): profiles/pmc_traffic.json NOT replaced"; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench_default_with_traffic.json" 2> "$OUT/bench_default_with_traffic.err"
python - "$OUT/bench_default_with_traffic.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["config"]["workload"], "frac", round(r["frac"], 3), "traffic", r.get("traffic"), "stale", r.get("traffic_stale"))
for s in d.get("secondary", []):
    print(" ", s["workload"], "frac", round(s["roofline"]["frac"], 3), "traffic", s["roofline"].get("traffic"))
PY
fi
if [ "$STAGE" != "a" ]; then
echo "== all-sizes rate table"
timeout -k 10 900 python tools/all_sizes_rate.py > "$OUT/all_sizes.txt" 2>/dev/null || true
tail -3 "$OUT/all_sizes.txt"
echo "== 2-D rates"
timeout -k 10 300 python tools/rate_2d.py > "$OUT/rates_2d.txt" 2>/dev/null || true
tail -3 "$OUT/rates_2d.txt"
echo "== fused-convolution rates"
timeout -k 10 400 python tools/conv_frames_rate.py c1024 c2048 c4096 r2048 r4096 r8192 c8192 c16384 c32768 r65536 c96 c480 c768 c1536 c9216 c19200 r960 r9216 c65536 c131072 r131072 r41472 > "$OUT/conv_frames_rates.txt" 2>/dev/null || true
tail -4 "$OUT/conv_frames_rates.txt"
echo "== engine, JUCE layouts"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --workload ols_engine_b2048_p16_c8192 --steps 20 --warmup 3 > "$OUT/bench_ols_engine.json" 2>/dev/null || true
timeout -k 10 300 python tools/juce_rate.py > "$OUT/juce_rates.txt" 2>/dev/null || true
fi
echo "== done"
