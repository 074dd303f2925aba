#!/bin/bash
# Quick state check on the GPU box: parity suite, smoke, the default bench line, and the same bench under the driver's
# multi-rank launcher with one rank (RCCL initialised, barrier + max-over-ranks path exercised).
TAG=${1:-verify}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1
RC=$?
tail -3 "$OUT/pytest_gpu.log"
[ $RC -ne 0 ] && exit $RC
echo "== smoke"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/smoke.log" || exit 1
echo "== bench"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail -5 "$OUT/bench_default.err"; exit 1; }
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], "value", round(d["value"], 5), "ms/step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3))
for s in d.get("secondary", []):
    print(" ", s["workload"], round(s["value"], 5), "ms", round(s["ms_per_step"], 3), "frac", round(s["roofline"]["frac"], 3))
PY
echo "== bench under torch.distributed.run, one rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > "$OUT/bench_torchrun1.json" 2> "$OUT/bench_torchrun1.err" || { tail -5 "$OUT/bench_torchrun1.err"; exit 1; }
tail -1 "$OUT/bench_torchrun1.json" | cut -c1-300
echo "== done"
