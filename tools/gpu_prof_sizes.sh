#!/bin/bash
# rocprofv3 kernel stats of fwd + bwd steps of the given sizes (tools/ab_env.py with a dummy switch): which sweep of a plan is the slow one?
# usage: bash tools/gpu_prof_sizes.sh r1048576,r524288 [tag]
W=${1:-r1048576}
TAG=${2:-prof_sizes}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/tools/ab_env.py" --env DUMMY=0 --rounds 2 --iters 10 --workloads "$W" > "$OUT/run.log" 2>&1
cd "$REPO"
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -r cat > "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "cfft" not in n:
        continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us avg  x{r["Calls"]:>5s}  {float(r["Percentage"]):5.1f} %  {n[:110]}')
PY
