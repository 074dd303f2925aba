#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs written by tools/gpu_round.sh.

For every kernel: dispatches, mean counter value per dispatch (KiB -> bytes), and -- for the membench
calibration kernels whose true byte count is known -- the counter/true ratio, which is the correction
the FFT kernels' numbers need (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced
stream; other access widths must be calibrated).
usage: tools/pmc_summary.py gpurun_out/<tag>
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(dirname):
    rows = defaultdict(list)  # (kernel, counter) -> [values]
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r.get("Kernel_Name") or r.get("kernel_name") or "?"
                c = r.get("Counter_Name") or r.get("counter_name") or "?"
                v = r.get("Counter_Value") or r.get("counter_value") or "0"
                try:
                    rows[(k, c)].append(float(v))
                except ValueError:
                    pass
    return rows


def short(name):
    name = re.sub(r"HIP_vector_type<float, 2u>", "float2", name)
    name = re.sub(r"HIP_vector_type<float, 4u>", "float4", name)
    return name[:110]


def main():
    out = sys.argv[1]
    result = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for kind in ("cal", "pmc"):
            d = os.path.join(out, f"prof_{kind}_{counter}")
            if not os.path.isdir(d):
                continue
            rows = load(d)
            print(f"--- {counter} ({'membench calibration' if kind == 'cal' else 'bench.py'})")
            for (k, c), vals in sorted(rows.items()):
                if c != counter:
                    continue
                mean_bytes = sum(vals) / len(vals) * 1024.0
                print(f"{len(vals):4d} x {mean_bytes / 2**20:10.1f} MiB/dispatch  {short(k)}")
                result.setdefault(kind, {}).setdefault(short(k), {})[counter] = {"dispatches": len(vals), "bytes_per_dispatch": mean_bytes}
    with open(os.path.join(out, "pmc_summary.json"), "w") as f:
        json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
