#!/usr/bin/env python3
"""Static check of the VMEM schedule of every gfx950 kernel of one translation unit.

gfx9 has ONE counter (vmcnt) for loads and stores and the compiler treats it as in-order, so
  * a load whose value is needed right away ("global_load; s_waitcnt vmcnt(0)") costs a full round trip, and 32 of them in a row
    cost 32 (seen when loads sit in divergent branches, or when the scheduler trades load registers for waits), and
  * a wait for a load issued AFTER stores also waits for those stores to reach L2/memory ("store drain").
Both showed up as 1.5-2x on the two-pass tile kernels (profiles/r02_vmem_schedule.txt).  The scan walks the ISA linearly
(branches ignored) and reports, per kernel: loads, loads followed at once by vmcnt(0), and stores retired by a wait for a load.

    python tools/isa_vmem_scan.py kernels_multipass [CFFT_PACKED] [name-regex]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chowdsp_fft_amd", "csrc")


def scan(unit, packed=0, pattern=""):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffp-contract=fast",
               f"-DCFFT_PACKED={packed}", "-c", os.path.join(CSRC, unit + ".hip"), "-o", os.path.join(tmp, "u.o"), "-save-temps=obj",
               "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        asm = open(os.path.join(tmp, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    info, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            info[cur] = {}
        for key, tag in (("VGPRs", "vgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "occ")):
            m = re.search(" " + re.escape(key) + r": (\d+)", line)
            if m and cur:
                info[cur][tag] = int(m.group(1))
    out = []
    for m in re.finditer(r"^(_ZN4cfft\w+):\s*;.*?\n(.*?)s_endpgm", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pattern and not re.search(pattern, name):
            continue
        lines = body.split("\n")
        pending = []  # 'L' / 'S' in issue order
        loads = serial = drained = 0
        for i, l in enumerate(lines):
            t = l.strip()
            if t.startswith(("global_load", "buffer_load", "flat_load")) or (t.startswith("global_atomic") and "sc0" in t):
                loads += 1
                pending.append("L")
                if any("s_waitcnt vmcnt(0)" in x for x in lines[i + 1:i + 4]):
                    serial += 1
            elif t.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
                pending.append("S")
            else:
                w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
                if w:
                    keep = int(w.group(1))
                    retire = pending[:len(pending) - keep] if keep < len(pending) else []
                    if "L" in retire:  # a wait for a load: every store issued before that load's successors goes with it
                        last_load = max(j for j, k in enumerate(retire) if k == "L")
                        drained += retire[:last_load].count("S")
                    pending = pending[len(pending) - keep:] if keep < len(pending) else pending
        out.append(dict(name=name, loads=loads, serial=serial, store_drains=drained, **info.get(name, {})))
    return out


if __name__ == "__main__":
    unit = sys.argv[1] if len(sys.argv) > 1 else "kernels_multipass"
    packed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    pat = sys.argv[3] if len(sys.argv) > 3 else ""
    for k in scan(unit, packed, pat):
        print(f"{k['name'][8:78]:70s} loads={k['loads']:4d} serial={k['serial']:3d} store_drains={k['store_drains']:4d} "
              f"vgpr={k.get('vgpr')} scratch={k.get('scratch')} occ={k.get('occ')}")
