#!/usr/bin/env python3
"""Timing experiments on the slice-parallel compressor (not a test, not a bench line).
  LZ4 alone / MD5 alone / both, per launch; with SKYHIP_LIB_PATH=scripts/dev/libskyhip_prof.so also the s_memtime phase table."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyplane_amd import hip_ops, synth
n = int(os.environ.get("CHUNKS", "1024")); cb = synth.CHUNK_BYTES
kind = os.environ.get("STREAM", "silesia")
if kind in synth.CLASSES:      # STREAM=<class>: 64 MiB of one class of the Silesia-like stream (what does each kind of data cost per block?)
    unit = synth.gen_class(kind, 64 << 20, synth.rng_for(2, 77))
else:
    unit = synth.silesia_like(64 << 20, config_id=2) if kind == "silesia" else synth.mixed_chunks(8, cb, config_id=4).reshape(-1)
d_unit = torch.from_numpy(unit).cuda()
d_big = torch.zeros(n * cb + 8192, dtype=torch.uint8, device="cuda")      # slack on both sides: dev variants that read the block's neighbourhood from global memory
d_in = d_big[4096:4096 + n * cb]
for t in range(n * cb // unit.size):
    d_in[t * unit.size:(t + 1) * unit.size] = torch.roll(d_unit, -((t * 7919 * 4096 + t * 13) % unit.size) if kind != "mixed" else 0)
stride = (hip_ops.frame_bound(cb) + 255) & ~255
d_out = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
in_off = np.arange(n, dtype=np.uint64) * cb; in_len = np.full(n, cb, np.uint64)
out_off = np.arange(n, dtype=np.uint64) * stride; out_cap = np.full(n, stride, np.uint64)
torch.cuda.synchronize()      # the library runs on its own streams: the stream must be complete before it is handed over
ctx = hip_ops.SkyHipContext(0, cb, n)
names = ["load + table clear", "pre-pass", "barrier (pre-pass)", "probe-all", "parse", "scan 1 (waits for the slowest parse)", "records + pass A", "scans 2-4",
         "emit", "barrier + bulk copies", "flush + barrier", "block total", "blocks x waves"]
runs = (("lz4", hip_ops.F_LZ4), ("md5", hip_ops.F_MD5), ("lz4+md5", hip_ops.F_LZ4 | hip_ops.F_MD5))
if os.environ.get("ONLY"):
    runs = tuple(r for r in runs if r[0] == os.environ["ONLY"])
for label, flags in runs:
    ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, flags)
    pr = (ctypes.c_uint64 * 16)(); ctx._lib.skyhip_debug_prof(ctx._h, pr)      # discard the warm-up
    ctx.reset_timing()
    R = 3
    for _ in range(R):
        ol, _ = ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, flags)
    t = ctx.timing()
    print(f"{label:8s} lz4 {t.lz4_ms/R:8.2f} ms ({n*cb/(max(t.lz4_ms,1e-9)/R/1e3)/1e9:7.1f} GB/s of input)  md5 {t.md5_ms/R:8.2f} ms  gather {t.gather_ms/R:6.2f} ms  ratio {n*cb/max(int(ol.sum()),1):.4f}", flush=True)
    if os.environ.get("BY_WAVE") and label == "lz4":      # per wave of the workgroup: mean cycles per block in every phase (prof builds)
        allc = {}
        for w in range(16):
            os.environ["SKYHIP_PROF_WAVE"] = str(w)
            for _ in range(R):
                ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, flags)
            ctx._lib.skyhip_debug_prof(ctx._h, pr)
            allc[w] = [pr[i] / max(pr[12], 1) for i in range(12)]
        del os.environ["SKYHIP_PROF_WAVE"]
        print("  wave:  " + " ".join(f"{w:6d}" for w in range(16)))
        for i in range(11):
            print(f"  {names[i][:22]:22s} " + " ".join(f"{allc[w][i]:6.0f}" for w in range(16)))
        continue
    ctx._lib.skyhip_debug_prof(ctx._h, pr)
    if pr[12] and label == "lz4":
        tot = pr[11]
        print("  phase: wave-cycles per block (mean over the 16 waves) / share")
        for i in range(11):
            print(f"    {names[i]:40s} {pr[i]/pr[12]:10.0f}  {100.0*pr[i]/tot:5.1f}%")
        print(f"    {'block total':40s} {pr[11]/pr[12]:10.0f}   unattributed {100.0*(tot-sum(pr[:11]))/tot:.1f}%")
