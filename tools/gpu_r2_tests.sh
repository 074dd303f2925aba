#!/bin/bash
# full gpu suite (all failures) + smoke + default bench; short inner timeouts
OUT=$PWD/gpurun_out/${1:-r02t}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
tail -15 "$OUT/pytest_gpu.log"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], round(d["value"], 5), "frac", round(d["roofline"]["frac"], 3))
for s in d.get("secondary", []):
    print(" ", s["workload"], round(s["value"], 5), "ms", round(s["ms_per_step"], 3), "frac", round(s["roofline"]["frac"], 3))
PY
