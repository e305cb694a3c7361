#!/usr/bin/env python3
"""Per-class and bench-stream frame sizes of the sequential model (tests/model/lz4s_model.c) under spec switches, against the reference's frames
   (liblz4, python-lz4 defaults = block-linked): the two guards of tests/test_gpu_parity.py (every class within 10 %, the stream within 3 %) priced on the CPU.
   scripts/dev/ratio_classes.py "" "-DLZ4S_Q=1 -DLZ4S_RLOG=16 -DLZ4S_LOGB=14" ..."""
import ctypes as C, os, subprocess, sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from skyplane_amd import synth
from oracle import ref

def model(flags):
    so = f"/tmp/mv/m_{abs(hash(flags))}.so"
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", f"-I{ROOT}/skyplane_amd/csrc", "-o", so, os.environ.get("MODEL_SRC", f"{ROOT}/tests/model/lz4s_model.c")] + flags.split(), check=True)
    lib = C.CDLL(so); lib.lz4s_model_block.restype = C.c_uint32; lib.lz4s_model_block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return lib
def frame_bytes(lib, data):
    tot = 19
    for o in range(0, data.size, 65536):
        b = np.ascontiguousarray(data[o:o + 65536])
        tot += min(lib.lz4s_model_block(b.ctypes.data, b.size, None, None), b.size) + 4
    return tot
def run(flags):
    lib = model(flags)
    out = {}
    for name in synth.CLASSES:
        d = synth.gen_class(name, 4 << 20, synth.rng_for(9))
        out[name] = frame_bytes(lib, d) / len(ref.lz4f_compress(d.tobytes()))
    d = synth.silesia_like(32 << 20, config_id=2)
    ours = sum(frame_bytes(lib, d[i:i + synth.CHUNK_BYTES]) for i in range(0, d.size, synth.CHUNK_BYTES))
    theirs = sum(len(ref.lz4f_compress(d[i:i + synth.CHUNK_BYTES].tobytes())) for i in range(0, d.size, synth.CHUNK_BYTES))
    out["STREAM(3%)"] = ours / theirs
    out["ratio"] = d.size / ours
    return flags, out
if __name__ == "__main__":
    variants = sys.argv[1:] or [""]
    with ProcessPoolExecutor(min(len(variants), 8)) as ex:
        res = list(ex.map(run, variants))
    keys = list(res[0][1].keys())
    print(" ".join(f"{k:>11s}" for k in keys))
    for f, o in res:
        print(" ".join(f"{(o[k] - 1) * 100:+10.2f}%" if k != "ratio" else f"{o[k]:11.4f}" for k in keys), " ", f or "(shipping spec)")
