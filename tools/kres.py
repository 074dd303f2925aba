#!/usr/bin/env python3
"""Compile a .hip file for gfx950 and print one line per kernel: VGPRs, SGPRs, scratch, occupancy, LDS.

usage: tools/kres.py file.hip [extra hipcc flags...]
"""
import re, subprocess, sys
src = sys.argv[1]
# the per-unit CFFT_PACKED value of the build (csrc/Makefile PACKED_<unit>): the header's default (1) is NOT what most units are built with
import os
unit = os.path.splitext(os.path.basename(src))[0]
mk = os.path.join(os.path.dirname(os.path.abspath(src)), "Makefile")
if os.path.exists(mk) and not any(a.startswith("-DCFFT_PACKED") for a in sys.argv[2:]):
    m = re.search(r"^PACKED_" + re.escape(unit) + r"\s*:=\s*(\d+)", open(mk).read(), re.M)
    if m:
        sys.argv.append("-DCFFT_PACKED=" + m.group(1))
# the library's own flags (csrc/Makefile): without -fno-slp-vectorize / -ffp-contract=fast the register counts are not the build's
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffp-contract=fast", "-c", src, "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
p = subprocess.run(cmd, capture_output=True, text=True)
txt = p.stderr
if p.returncode != 0:
    print(txt); sys.exit(1)
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}; rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and " " + key in " " + line:
            cur[key.split(" ")[0]] = int(m.group(1))
for r in rows:
    n = re.sub(r"HIP_vector_type<float, 2u>", "float2", r["name"])
    n = re.sub(r"\(.*", "", n)
    print(f"vgpr={r.get('VGPRs'):4d} agpr={r.get('AGPRs', 0):3d} sgpr={r.get('TotalSGPRs'):3d} scratch={r.get('ScratchSize'):4d} occ={r.get('Occupancy')} {n}")
