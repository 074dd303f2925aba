#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes taken by tools/pmc_refresh.sh into profiles-ready pmc_traffic.json.

For every bench workload: bytes per library launch = (2 x sum FETCH_SIZE + sum WRITE_SIZE) over the workload's kernels
(bench.WORKLOAD_KERNELS) / library launches of the profiled command -- FETCH_SIZE reads half the bytes of a coalesced stream
on gfx950, WRITE_SIZE reads them exactly, both in KiB (MI355X_MICROARCH.md, HBM section; calibrated in
profiles/r01_rocprofv3_pmc_hbm_traffic.txt).  The file records a hash of the library SOURCES and of the shared object the
passes ran on; bench.py withholds a figure whose source hash is not the one it is timing.

usage: tools/pmc_refresh.py <dir with <workload>/prof_pmc_{FETCH,WRITE}_SIZE> <steps> <warmup> <out.json> [note]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chowdsp_fft_amd import _buildinfo  # noqa: E402


def counter_sum(dirname, counter, patterns):
    total, dispatches, names = 0.0, 0, set()
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if (r.get("Counter_Name") or "") != counter:
                    continue
                k = r.get("Kernel_Name") or ""
                if not any(p in k for p in patterns):
                    continue
                total += float(r.get("Counter_Value") or 0.0) * 1024.0
                dispatches += 1
                names.add(k.split("(")[0][:100])
    return total, dispatches, sorted(names)


def main():
    out_dir, steps, warmup, out_json = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    result = {
        "_note": "fabric bytes per bench launch = 2*FETCH_SIZE + WRITE_SIZE summed over the workload's kernels (rocprofv3 --pmc, "
                 "separate passes; FETCH_SIZE reads 1/2 on gfx950) / library launches; written by tools/pmc_refresh.sh",
        "_taken_at": note,
        "src_sha256": _buildinfo.source_sha256(),
        "lib_sha256": _buildinfo.lib_sha256(),
        "workloads": {},
    }
    for w in sorted(bench.WORKLOADS):
        d = os.path.join(out_dir, w)
        if not os.path.isdir(d):
            continue
        pats = bench.WORKLOAD_KERNELS.get(w, [])
        launches = (steps + warmup) * bench.launches_per_step(w)
        fetch, nf, names = counter_sum(os.path.join(d, "prof_pmc_FETCH_SIZE"), "FETCH_SIZE", pats)
        write, nw, _ = counter_sum(os.path.join(d, "prof_pmc_WRITE_SIZE"), "WRITE_SIZE", pats)
        if nf == 0 or nw == 0:
            print(f"{w}: no dispatches matched {pats}", file=sys.stderr)
            continue
        n, transform, frames, _ = bench.WORKLOADS[w]
        frame_floats = 2 * n if transform == "complex" else n
        alg = 4 * frame_floats * 2 * frames
        per_launch = (2.0 * fetch + write) / launches
        result["workloads"][w] = {"bytes_per_launch": per_launch, "read_bytes_per_launch": 2.0 * fetch / launches,
                                  "write_bytes_per_launch": write / launches, "algorithmic_bytes_per_launch": alg,
                                  "over_algorithmic": per_launch / alg, "kernels": names,
                                  "dispatches_per_launch": nf / launches, "library_launches_profiled": launches}
        print(f"{w:45s} {per_launch / 2**20:10.1f} MiB/launch = {per_launch / alg:.3f} x algorithmic  (read {2 * fetch / launches / 2**20:.1f} + write {write / launches / 2**20:.1f} MiB; {nf / launches:.1f} dispatches per launch)")
    with open(out_json, "w") as f:
        json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
