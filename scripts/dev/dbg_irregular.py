#!/usr/bin/env python3
"""dev: which of the flushed (irregular) frames does the GPU decode wrongly, and where?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref
from skyplane_amd import hip_ops, synth
torch.cuda.init()
ctx = hip_ops.SkyHipContext(0, 8 << 20, 8)
rng = synth.rng_for(77)
d = synth.gen_class("text", 300_000, rng).tobytes() + bytes(70_000) + synth.gen_class("records", 1_000_000, rng).tobytes()
cuts = [0, 70_000, 70_001, 200_000, 200_013, 330_000, 900_000, len(d)]
pieces = [d[a:b] for a, b in zip(cuts, cuts[1:])]
frames = [ref.lz4f_compress_stream(pieces, store_size=s, block_linked=l) for l in (True, False) for s in (True, False)] + [ref.lz4f_compress(d)]
for sel in ([0], [1], [2], [3], [4], list(range(5))):
    try:
        outs = ctx.decompress_batch([frames[i] for i in sel], [len(d)] * len(sel))
    except Exception as e:
        print(sel, "EXC", e, ctx.last_decode_status); continue
    for i, o in zip(sel, outs):
        a, b = np.frombuffer(o, np.uint8), np.frombuffer(d, np.uint8)
        bad = np.nonzero(a != b)[0] if a.size == b.size else None
        print(sel, i, len(o), None if bad is None else (int(bad.size), bad[:4].tolist()), flush=True)
