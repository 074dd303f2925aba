#!/bin/bash
# Which tile kernels the long-plan parity test runs: every static length should show up with its compile-time list
mkdir -p gpurun_out/r3tiles
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r3tiles/stats -- python3 -m pytest $REPO/tests/test_gpu_other_sizes.py -m gpu -q -k many_tiles -p no:cacheprovider > $REPO/gpurun_out/r3tiles/log.txt 2>&1
cd $REPO
tail -2 gpurun_out/r3tiles/log.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3tiles/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "generic_kernel" in r["Name"]:
            print("%-130s calls %5s" % (r["Name"].split("(")[0][:130], r["Calls"]))
PY
