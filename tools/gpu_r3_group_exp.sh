#!/bin/bash
# round 3: where does the static-group form's time go?  PMC traffic of the shipped kernel, then the experiment build's
# timing-only switches (results wrong by design) and phase stamps
set -o pipefail
REPO=$PWD; OUT=$REPO/gpurun_out/r3; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/group_pmc_$C -- python3 $REPO/tools/group_time.py 8192 2 > $OUT/group_pmc_$C.log 2>&1 || exit 1
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {}
    for path in glob.glob(out + f"/group_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == c and "cfft" in r["Kernel_Name"]:
                vals.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        print(f"{c} {len(v):3d} x {sum(v)/len(v)/1024:9.1f} MiB/dispatch  {k}")
PY
export CHOWDSP_FFT_LIB=$REPO/chowdsp_fft_amd/variants/lib_gexp.so
for D in 0 1 6 7 8 16 24 30 31; do
  echo "== dbg $D"
  CHOWDSP_FFT_GROUP_DBG=$D timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep form || exit 1
done
echo "== stamps (dbg 32)"
CHOWDSP_FFT_GROUP_DBG=32 CHOWDSP_FFT_FUSED_DEBUG=1 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep "stamps\|form" | tail -12
echo "== stamps, one workgroup per CU"
CHOWDSP_FFT_GROUP_WG_PER_CU=1 CHOWDSP_FFT_GROUP_DBG=32 CHOWDSP_FFT_FUSED_DEBUG=1 timeout -k 10 120 python tools/group_time.py 8192 2 2>&1 | grep "stamps\|form" | tail -12
