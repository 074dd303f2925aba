#!/bin/bash
# SQ counters and per-kernel times of the general two-pass sizes (profiles/r03_general_two_pass.txt)
mkdir -p gpurun_out/r3gen
SIZES=${1:-c20736,c46080,c245760,c2097152}
bash tools/prof_sq_cmd.sh r3gen python3 $PWD/tools/all_sizes_rate.py $SIZES > gpurun_out/r3gen/sq_counters.txt 2>&1 &&
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3gen/stats -- python3 $OLDPWD/tools/all_sizes_rate.py $SIZES > $OLDPWD/gpurun_out/r3gen/stats.log 2>&1) &&
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3gen/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cfft" in r["Name"]:
            print("%-120s calls %6s avg %10.1f us  %5s %%" % (r["Name"].split("(")[0][:120], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
