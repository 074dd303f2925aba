#!/usr/bin/env python3
"""Finds serialised load loops in a gfx950 .s file: loops (a backward branch to a label) whose body issues only a few
global loads and waits for vmcnt(0) -- one full memory round trip per iteration and work-item.  tools/isa_vmem_scan.py walks
the code linearly and misses these (the wait sits at the loop top, in another basic block than the load).

    python tools/isa_loop_scan.py file.s [name-regex] [max-loads-per-iteration, default 3]
"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    asm = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    max_loads = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for m in re.finditer(r"^(_ZN4cfft\w+):\s*;.*?\n(.*?)s_endpgm", asm, re.S | re.M):
        name, body = demangle(m.group(1)), m.group(2)
        if pat and not re.search(pat, name):
            continue
        lines = body.split("\n")
        labels = {}
        for i, l in enumerate(lines):
            lm = re.match(r"^(\.LBB\d+_\d+):", l)
            if lm:
                labels[lm.group(1)] = i
        found = []
        for i, l in enumerate(lines):
            bm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not bm or bm.group(1) not in labels or labels[bm.group(1)] >= i:
                continue
            seg = lines[labels[bm.group(1)]:i]
            if any(re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", s) and labels.get(re.search(r"(\.LBB\d+_\d+)", s).group(1), 1 << 30) < labels[bm.group(1)] for s in seg):
                continue
            loads = sum(1 for s in seg if s.strip().startswith(("global_load", "buffer_load", "flat_load")))
            stores = sum(1 for s in seg if s.strip().startswith(("global_store", "buffer_store")))
            wait0 = any("vmcnt(0)" in s for s in seg)
            if 1 <= loads <= max_loads and wait0 and len(seg) < 400:
                found.append((labels[bm.group(1)], len(seg), loads, stores))
        if found:
            print(name[:150])
            for f in found:
                print("    loop at +%d: %d instructions, %d loads, %d stores, waits vmcnt(0)" % f)


main()
