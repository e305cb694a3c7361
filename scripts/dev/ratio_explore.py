#!/usr/bin/env python3
"""Ratio of the sequential model of the slice-parallel parse (tests/model/lz4s_model.c) under different spec switches, on the bench stream.
   scripts/dev/ratio_explore.py "-DLZ4S_INS_STEP=4" "-DLZ4S_VISITS_MODEL=8u" ...   (each argument = one variant's extra compiler flags; '' = the spec as shipped)"""
import ctypes as C, os, subprocess, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from skyplane_amd import synth

MIB = int(os.environ.get("MIB", "64"))
def run(flags):
    so = f"/tmp/mv/m_{abs(hash(flags))}.so"
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", f"-I{ROOT}/skyplane_amd/csrc", "-o", so, f"{ROOT}/tests/model/lz4s_model.c"] + flags.split(), check=True)
    lib = C.CDLL(so); lib.lz4s_model_block.restype = C.c_uint32; lib.lz4s_model_block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    data = synth.silesia_like(MIB << 20, config_id=2) if os.environ.get("STREAM", "silesia") == "silesia" else synth.mixed_chunks(MIB // 8, 8 << 20, config_id=4).reshape(-1)
    tot = 0
    for o in range(0, data.size, 65536):
        b = np.ascontiguousarray(data[o:o + 65536])
        tot += min(lib.lz4s_model_block(b.ctypes.data, b.size, None, None), b.size) + 4
    return flags, data.size / tot
if __name__ == "__main__":
    os.makedirs("/tmp/mv", exist_ok=True)
    variants = sys.argv[1:] or [""]
    with ProcessPoolExecutor(min(len(variants), 8)) as ex:
        res = list(ex.map(run, variants))
    base = res[0][1]
    for f, r in res:
        print(f"{r:8.4f}  {100 * (r / base - 1):+6.2f}%   {f or '(shipping spec)'}")
