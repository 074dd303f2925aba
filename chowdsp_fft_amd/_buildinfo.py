"""Identity of the code a measurement was taken on: a hash of the library's sources and of the built shared object.

profiles/pmc_traffic.json (HBM / fabric bytes per launch from rocprofv3 PMC passes, which the bench run itself cannot take)
carries both; bench.py prints `roofline.traffic` only when they still describe the library it is timing, else `null` and
`"traffic_stale": true` (tools/pmc_refresh.sh regenerates every entry).
"""
from __future__ import annotations

import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_files():
    csrc = os.path.join(_HERE, "csrc")
    files = []
    for pat in ("*.hpp", "*.hip", "*.cpp", "Makefile"):
        files += glob.glob(os.path.join(csrc, pat))
    # the public headers are prerequisites of every object (Makefile HDRS)
    files += glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    return sorted(os.path.normpath(f) for f in files)


def source_sha256() -> str:
    """sha256 over the names and contents of csrc/*.{hpp,hip,cpp} + Makefile + include/*.h (what the shared object is built from)"""
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def lib_sha256(path: str | None = None) -> str | None:
    from . import _lib

    path = path or _lib.LIB_PATH
    if not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def lib_is_newer_than_sources(path: str | None = None) -> bool:
    from . import _lib

    path = path or _lib.LIB_PATH
    if not os.path.exists(path):
        return False
    t = os.path.getmtime(path)
    return all(os.path.getmtime(p) <= t for p in source_files())
