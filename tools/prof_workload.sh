#!/bin/bash
# rocprofv3 kernel stats + HBM counters for one bench workload.  usage: tools/prof_workload.sh <workload> <tag> [extra bench args]
W=$1; TAG=$2; shift 2
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W "$@" > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_pmc_$C -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $W "$@" > $OUT/pmc_$C.log 2>&1
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cfft" in r["Name"]:
            print("%6s calls avg %10.1f us  %5s%%  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"][:120]))
PY
python tools/pmc_summary.py $OUT | grep -E "cfft|---"
