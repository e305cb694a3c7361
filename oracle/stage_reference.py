#!/usr/bin/env python3
"""Stage the reference's Python package for the GPU box.  TEST / MEASUREMENT INFRASTRUCTURE.

The reference is pure Python: there is nothing to compile.  What the GPU box lacks is the tree itself (/root/reference exists only
in the build container), so the measurements that need the reference's REAL daemons next to a GPU -- BASELINE configs[0] on the GPU
box's host cores, configs[4] with the gpu_compress operator inside gateway_daemon.py's DAG -- could not run there.  This recipe
copies /root/reference/skyplane into oracle/_ref/skyplane: oracle/_ref/ is git-ignored (no reference source ever enters this
repository's history) but travels with gpurun, exactly like a built .so.  __graft_entry__.build() runs it whenever /root/reference is
present; oracle/refshim.py falls back to the staged copy when /root/reference is not.
"""
import shutil
import sys
from pathlib import Path

SRC = Path("/root/reference/skyplane")
DST = Path(__file__).resolve().parent / "_ref" / "skyplane"


def stage() -> bool:
    if not (SRC / "chunk.py").is_file():
        return False
    if DST.exists():
        shutil.rmtree(DST)
    shutil.copytree(SRC, DST, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    (DST.parent / "STAGED_FROM").write_text(f"{SRC} (read-only reference tree; staged copy, never committed)\n")
    return True


if __name__ == "__main__":
    ok = stage()
    print("staged" if ok else "no reference tree here: nothing staged")
    sys.exit(0)
