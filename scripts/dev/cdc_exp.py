#!/usr/bin/env python3
"""Timing of the CDC / fingerprint / dedup kernels ALONE (no compressor, no whole-chunk MD5 beside them) on the configs[2] stream -- not a test, not a
bench line.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.   CHUNKS=2048 REPS=3 FLAGS=12 python scripts/dev/cdc_exp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyplane_amd import hip_ops, synth
n = int(os.environ.get("CHUNKS", "2048")); cb = synth.CHUNK_BYTES; R = int(os.environ.get("REPS", "3"))
flags = int(os.environ.get("FLAGS", str(hip_ops.F_CDC | hip_ops.F_DEDUP)))
unit = synth.dedup_stream(min(256 << 20, n * cb), dup_fraction=0.5, config_id=3)
d_unit = torch.from_numpy(unit).cuda()
d_in = torch.empty(n * cb, dtype=torch.uint8, device="cuda")
per = unit.size
for t in range((n * cb + per - 1) // per):
    lo, hi = t * per, min((t + 1) * per, n * cb)
    tile = torch.roll(d_unit, -((t * 7919 * 4096 + t * 13) % per))[: hi - lo]
    d_in[lo:hi] = tile ^ (t & 0xFF) if (t & 0xFF) else tile
stride = (hip_ops.frame_bound(cb) + 255) & ~255
d_out = torch.empty(n * stride if flags & 1 else 16, dtype=torch.uint8, device="cuda")
in_off = np.arange(n, dtype=np.uint64) * cb; in_len = np.full(n, cb, np.uint64)
out_off = np.arange(n, dtype=np.uint64) * stride; out_cap = np.full(n, stride, np.uint64)
torch.cuda.synchronize()
ctx = hip_ops.SkyHipContext(0, cb, n)
ctx.dedup_reset(); ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, flags)
ctx.reset_timing(); t0 = time.perf_counter()
for _ in range(R):
    ctx.dedup_reset()
    ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, flags)
wall = (time.perf_counter() - t0) / R
t = ctx.timing()
prefix, cuts, fps, first, base = ctx.cdc_results(n, in_len)
dup = first != np.arange(base, base + len(first), dtype=np.uint64)
import hashlib
print(f"flags {flags}: {n} chunks  cdc {t.cdc_ms / R:8.2f} ms  lz4 {t.lz4_ms / R:8.2f}  md5 {t.md5_ms / R:8.2f}  wall {wall * 1e3:8.2f} ms  "
      f"({n * cb / wall / 2**30:7.1f} GiB/s)  segments {len(first)}  dup segments {int(dup.sum())}  cuts-digest {hashlib.md5(cuts.tobytes()).hexdigest()[:12]} "
      f"fps-digest {hashlib.md5(fps.tobytes()).hexdigest()[:12]} first-digest {hashlib.md5((first - base).tobytes()).hexdigest()[:12]}", flush=True)
