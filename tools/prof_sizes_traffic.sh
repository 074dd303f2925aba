#!/bin/bash
# round 4: per-kernel time (rocprofv3 --kernel-trace --stats) and fabric traffic (--pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the
# forward + backward transform of any sizes of tools/all_sizes_rate.py.   usage: tools/prof_sizes_traffic.sh <tag> c131072,r131072 [options]
TAG=$1; SIZES=$2; shift 2
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/tools/all_sizes_rate.py $SIZES "$@" > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_pmc_$C -- python3 $REPO/tools/all_sizes_rate.py $SIZES "$@" > $OUT/pmc_$C.log 2>&1
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
print(open(out + "/stats.log").read().strip().splitlines()[-6:])
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cfft" in r["Name"]:
            print("%6s calls avg %10.1f us  %5s%%  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"].split("(")[0][:110]))
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + f"/prof_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cfft" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0][:110]][c].append(float(r["Counter_Value"]))
print("--- fabric traffic per launch (FETCH_SIZE is doubled: gfx950 reports half of a wide streaming read; both in KiB units x 1024)")
for k, d in acc.items():
    fe = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) * 2 * 1024 if d["FETCH_SIZE"] else 0  # counter unit KiB; x2 = the guide's gfx950 correction
    wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024 if d["WRITE_SIZE"] else 0
    print("   read %8.1f MiB  written %8.1f MiB   %s" % (fe / 2**20, wr / 2**20, k))
PY
