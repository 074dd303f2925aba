#!/bin/bash
# round 2, first GPU session: parity suite (all failures, not -x), smoke, the default bench line with its secondary configs
OUT=$PWD/gpurun_out/r02a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
tail -40 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 6000 "$OUT/bench_default.json"
tail -5 "$OUT/bench_default.err"
