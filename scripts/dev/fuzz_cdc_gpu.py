#!/usr/bin/env python3
"""One-off soak (not a test): the two segment-digest kernels of the library (sky_segment_md5x -- rows staged through LDS, the default -- and sky_segment_md5,
SKYHIP_SEGMD5_STAGED=0) over many structured-random batches: cuts, fingerprints and first-seen indices of the two contexts must be identical, and a sample of the
fingerprints is recomputed with hashlib.  SEEDS=a:b picks the generator seeds."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from skyplane_amd import hip_ops, synth
from tests.test_fuzz_roundtrip import _cases
a, b = (int(x) for x in os.environ.get("SEEDS", "100:140").split(":"))
os.environ["SKYHIP_SEGMD5_STAGED"] = "1"
cx = hip_ops.SkyHipContext(0, 8 << 20, 64)
cx.process_batch([bytes(5000)], flags=hip_ops.F_CDC)            # (the variable is read at the first CDC call)
os.environ["SKYHIP_SEGMD5_STAGED"] = "0"
co = hip_ops.SkyHipContext(0, 8 << 20, 64)
co.process_batch([bytes(5000)], flags=hip_ops.F_CDC)
bad = nseg = 0
rng = np.random.default_rng(5)
for seed in range(a, b):
    chunks = _cases(seed, 40) + [synth.dedup_stream(int(rng.integers(1, 4)) << 20, config_id=seed).tobytes(), synth.gen_class(synth.CLASSES[seed % len(synth.CLASSES)], int(rng.integers(1, 3 << 20)), synth.rng_for(seed)).tobytes()]
    lens = [len(c) for c in chunks]
    out = []
    for ctx in (cx, co):
        ctx.dedup_reset()
        ctx.process_batch(chunks, flags=hip_ops.F_CDC | hip_ops.F_DEDUP)
        out.append(ctx.cdc_results(len(chunks), lens))
    (p1, c1, f1, s1, b1), (p2, c2, f2, s2, b2) = out
    ok = (p1 == p2).all() and (c1 == c2).all() and (f1 == f2).all() and ((s1 - b1) == (s2 - b2)).all()
    # a sample of segments against hashlib
    for k in rng.integers(0, len(chunks), 6):
        k = int(k)
        lo, hi = int(p1[k]), int(p1[k + 1])
        if hi == lo:
            continue
        j = int(rng.integers(lo, hi))
        st = 0 if j == lo else int(c1[j - 1])
        ok = ok and f1[j].tobytes() == hashlib.md5(chunks[k][st:int(c1[j])]).digest()
    bad += not ok; nseg += int(p1[-1])
    if seed % 20 == 0 or not ok:
        print(seed, bool(ok), flush=True)
print("segments digested by both kernels:", nseg, "; batches that differ:", bad)
sys.exit(1 if bad else 0)
