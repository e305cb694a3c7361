#!/usr/bin/env python3
"""One-off soak (not a test): many more structured-random cases than tests/test_fuzz_roundtrip.py through the GPU decoder as liblz4's block-LINKED frames
(sky_lz4_parse + sky_lz4_link), several launches of mixed sizes, outputs compared with the inputs.  SEEDS=a:b picks the generator seeds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import ref
from skyplane_amd import hip_ops, synth
from tests.test_fuzz_roundtrip import _cases
from tests.test_emu_decompress import link_stress_cases
a, b = (int(x) for x in os.environ.get("SEEDS", "100:140").split(":"))
ctx = hip_ops.SkyHipContext(0, 8 << 20, 64)
bad = n = 0
extra = link_stress_cases() + [synth.gen_class(c, 3 << 20, synth.rng_for(31)).tobytes() for c in synth.CLASSES]
for seed in range(a, b):
    chunks = _cases(seed, 48) + extra
    frames = [ref.lz4f_compress(d) for d in chunks]
    for rep in range(2):
        outs = ctx.decompress_batch(frames, [len(d) for d in chunks])
        ok = outs == chunks
        bad += not ok; n += len(chunks)
    print(seed, ok, flush=True)
print("decoded", n, "frames; launches with a wrong output:", bad)
sys.exit(1 if bad else 0)
