#!/usr/bin/env python3
"""Timing experiment (not a test, not a bench line): LZ4 kernel time with parts switched off via SKYHIP_ABLATE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skyplane_amd import hip_ops, synth
n = int(os.environ.get("CHUNKS", "512")); cb = synth.CHUNK_BYTES
unit = synth.silesia_like(64 << 20, config_id=2)
d_unit = torch.from_numpy(unit).cuda()
d_in = torch.empty(n * cb, dtype=torch.uint8, device="cuda")
for t in range(n * cb // unit.size):
    d_in[t * unit.size:(t + 1) * unit.size] = torch.roll(d_unit, -((t * 7919 * 4096 + t * 13) % unit.size))
stride = (hip_ops.frame_bound(cb) + 255) & ~255
d_out = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
in_off = np.arange(n, dtype=np.uint64) * cb; in_len = np.full(n, cb, np.uint64)
out_off = np.arange(n, dtype=np.uint64) * stride; out_cap = np.full(n, stride, np.uint64)
ctx = hip_ops.SkyHipContext(0, cb, 512)
for abl in [int(x) for x in os.environ.get("ABLS", "0,1,2,4,5,7,15").split(",")]:
    os.environ["SKYHIP_ABLATE"] = str(abl)
    ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, hip_ops.F_LZ4)
    ctx.reset_timing()
    for _ in range(2):
        ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, hip_ops.F_LZ4)
    t = ctx.timing()
    import ctypes
    pr = (ctypes.c_uint64 * 16)()
    ctx._lib.skyhip_debug_prof(ctx._h, pr)
    if pr[9]:
        names = ["A1 input loads", "A2-3 hash/probe/period", "A4 cmp16", "A5 cmp16 #2 + finalize", "B1 parse+install", "B2 tokens", "B2 literals", "B2 carry copy", "block total", "blocks"]
        tot = pr[8]
        print("  phase cycles per block (sum over batches) / share of block time:")
        for i in range(8):
            print(f"    {names[i]:28s} {pr[i]/pr[9]:12.0f}  {100.0*pr[i]/tot:5.1f}%")
        print(f"    {'block total':28s} {pr[8]/pr[9]:12.0f}   ({pr[9]} blocks; unattributed {100.0*(tot-sum(pr[:8]))/tot:.1f}%)")
    print(f"ablate={abl:2d} lz4 {t.lz4_ms/2:8.2f} ms  -> {n*cb/ (t.lz4_ms/2e3)/1e9:7.1f} GB/s   gather {t.gather_ms/2:6.2f} ms", flush=True)
# CDC stage alone (no LZ4/MD5 co-running)
os.environ["SKYHIP_ABLATE"] = "0"
zero = np.zeros(n, np.uint64)
if not os.environ.get("DEC_ONLY"):
  ctx.process_device(d_in.data_ptr(), in_off, in_len, 0, zero, zero, hip_ops.F_CDC | hip_ops.F_DEDUP)
  ctx.dedup_reset(); ctx.reset_timing()
  for _ in range(2):
    ctx.dedup_reset()
    ctx.process_device(d_in.data_ptr(), in_off, in_len, 0, zero, zero, hip_ops.F_CDC | hip_ops.F_DEDUP)
  t = ctx.timing()
  print(f"cdc alone {t.cdc_ms/2:8.2f} ms -> {n*cb/(t.cdc_ms/2e3)/1e9:7.1f} GB/s", flush=True)
  ctx.reset_timing()
  for _ in range(2):
    ctx.process_device(d_in.data_ptr(), in_off, in_len, 0, zero, zero, hip_ops.F_MD5)
  t = ctx.timing()
  print(f"md5 alone {t.md5_ms/2:8.2f} ms ({n} chunks)", flush=True)
# decompression of the frames produced above (device resident)
ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, hip_ops.F_LZ4)
flen, _ = ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, hip_ops.F_LZ4)
d_back = torch.empty(n * cb, dtype=torch.uint8, device="cuda")
ctx.decompress_device(d_out.data_ptr(), out_off, flen, d_back.data_ptr(), in_off, in_len)
for lds in [int(x) for x in os.environ.get("DEC_LDS", "0").split(",")]:
    os.environ["SKYHIP_DEC_LDS"] = str(lds)          # honoured by -DSKY_ABL=1 builds only: occupancy experiment
    ctx.decompress_ms(reset=True)
    for _ in range(2):
        ctx.decompress_device(d_out.data_ptr(), out_off, flen, d_back.data_ptr(), in_off, in_len)
    ms = ctx.decompress_ms() / 2
    print(f"decompress (dyn LDS {lds:6d}) {ms:8.2f} ms -> {n*cb/(ms/1e3)/1e9:7.1f} GB/s of output; roundtrip ok = {bool(torch.equal(d_back, d_in))}", flush=True)
