#!/bin/bash
# SQ counters (wait / issue split, LDS conflicts, instruction mix) for any command.  usage: tools/prof_sq_cmd.sh <tag> python3 <script> [args]
# Two rocprofv3 --pmc passes (8 SQ slots each), kernel trace only; prints per-kernel means and the derived fractions.
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- "$@" > $OUT/sq.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/sq.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/sq2 -- "$@" > $OUT/sq2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/sq2.log; exit 1; }
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cfft" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    print(f"{k}   ({n} dispatches)")
    for c in sorted(m):
        print("   %-26s %16.0f" % (c, m[c]))
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc > 0:
        waves = m.get("SQ_WAVES", 0)
        print("   -> of wave cycles: VALU active %.1f %%, issue stalls (WAIT_INST_ANY) %.1f %% (LDS-issue part %.1f %%), parked on waits (WAIT_ANY) %.1f %%"
              % (100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc))
        if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            print("   -> LDS array active cycles %.0f, of which bank conflicts %.1f %%" % (m["SQ_LDS_IDX_ACTIVE"], 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]))
        if waves > 0:
            print("   -> per wave: %.0f VALU, %.0f LDS, %.0f VMEM-read, %.0f VMEM-write, %.0f SALU instructions" % (m.get("SQ_INSTS_VALU", 0) / waves, m.get("SQ_INSTS_LDS", 0) / waves, m.get("SQ_INSTS_VMEM_RD", 0) / waves, m.get("SQ_INSTS_VMEM_WR", 0) / waves, m.get("SQ_INSTS_SALU", 0) / waves))
PY
