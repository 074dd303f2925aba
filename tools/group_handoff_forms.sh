#!/bin/bash
# The two hand-off forms of the one-launch kernels side by side (option group_handoff: 0 fast, 1 the form the CDNA4 guide lists):
# time (tools/all_sizes_rate.py c65536,r131072) and fabric traffic per kernel (separate --pmc FETCH_SIZE / WRITE_SIZE passes).
# usage: tools/group_handoff_forms.sh <tag>      -> gpurun_out/<tag>/group_handoff_forms.txt
TAG=${1:-handoff}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp
{
for form in 0 1 0 1; do
  echo "== group_handoff=$form  (time)"
  timeout -k 10 200 python tools/all_sizes_rate.py c65536,r131072 group_handoff=$form 2>/dev/null
done
for form in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== group_handoff=$form  (--pmc $ctr, KiB per dispatch, mean over the dispatches of each kernel)"
    cd /tmp
    rm -rf "$OUT/pmc_${form}_$ctr"
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_${form}_$ctr" -- python3 "$REPO/tools/all_sizes_rate.py" c65536,r131072 group_handoff=$form > /dev/null 2>&1
    cd "$REPO"
    python3 - "$OUT/pmc_${form}_$ctr" $ctr <<'PY'
import csv, glob, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == ctr and "group256" in row.get("Kernel_Name", ""):
            acc[row["Kernel_Name"].split("(")[0][:70]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"   {k:70s} n={len(v):4d}  mean {sum(v)/len(v):14.0f} KiB" + ("   (x2 on gfx950 for streaming reads: MI355X_MICROARCH.md)" if ctr == "FETCH_SIZE" else ""))
PY
  done
done
} > "$OUT/group_handoff_forms.txt" 2>&1
cat "$OUT/group_handoff_forms.txt"
