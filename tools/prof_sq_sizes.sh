#!/bin/bash
# SQ counters (LDS conflicts, wait / issue split) for tools/all_sizes_rate.py <sizes>.  usage: tools/prof_sq_sizes.sh c192,c480 <tag>
S=$1; TAG=$2
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 $REPO/tools/all_sizes_rate.py $S > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $REPO/tools/all_sizes_rate.py $S > $OUT/sq2.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cfft" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %16.0f  (x%d)" % (c, sum(v) / len(v), len(v)))
PY
