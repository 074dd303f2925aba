#!/bin/bash
set -o pipefail
REPO=$PWD; OUT=$REPO/gpurun_out/r3; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_fused_two_pass.py -x -q 2>&1 | tail -3 || exit 1
echo "== one slot per group"; timeout -k 10 120 python tools/group_time.py 8192 0,2 2>&1 | grep form || exit 1
echo "== two slots"; CHOWDSP_FFT_GROUP_SLOTS=2 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
echo "== one slot, 3 per CU"; CHOWDSP_FFT_GROUP_WG_PER_CU=3 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/group1_pmc_$C -- python3 $REPO/tools/group_time.py 8192 2 > $OUT/group1_pmc_$C.log 2>&1 || exit 1
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {}
    for path in glob.glob(out + f"/group1_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == c and "cfft" in r["Kernel_Name"]:
                vals.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        print(f"{c} {len(v):3d} x {sum(v)/len(v)/1024:9.1f} MiB/dispatch  {k}")
PY
export CHOWDSP_FFT_LIB=$REPO/chowdsp_fft_amd/variants/lib_gexp.so
for D in 0 1 6 24; do
  echo "== dbg $D"
  CHOWDSP_FFT_GROUP_DBG=$D timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
done
echo "== stamps (dbg 32)"
CHOWDSP_FFT_GROUP_DBG=32 CHOWDSP_FFT_FUSED_DEBUG=1 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep "stamps\|form" | tail -9
