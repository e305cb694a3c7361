#!/usr/bin/env python3
"""Measurement (not a test): GPU decode rate of the frames a STOCK reference sender produces -- lz4.frame.compress with python-lz4's
defaults, i.e. block-LINKED 64 KiB blocks (gateway_operator.py:359) -- next to the rate on this library's own block-independent frames.
16 distinct 8 MiB chunks are compressed on the CPU with liblz4 and replicated to N frames on the device."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref
from skyplane_amd import hip_ops, synth
n = int(os.environ.get("FRAMES", "1024")); cb = synth.CHUNK_BYTES
unit = synth.silesia_like(16 * cb, config_id=2)
chunks = [unit[i * cb:(i + 1) * cb] for i in range(16)]
ctx = hip_ops.SkyHipContext(0, cb, 64)
out = {}
kinds = ("linked (reference default)", "independent (liblz4)", "independent (this library)")
if os.environ.get("ONLY_LINKED"):
    kinds = kinds[:1]
if os.environ.get("ONLY_OURS"):
    kinds = kinds[2:]
sizes = tuple(int(x) for x in os.environ.get("SIZES", f"{n},32").split(","))
for kind in kinds:
    if kind.startswith("independent (this"):
        frames = [r.frame for r in ctx.process_batch([c.tobytes() for c in chunks], flags=hip_ops.F_LZ4)]
    else:
        frames = [ref.lz4f_compress(c, block_linked=kind.startswith("linked")) for c in chunks]
    stride = (max(len(f) for f in frames) + 255) & ~255
    h = np.zeros(16 * stride, np.uint8)
    for i, f in enumerate(frames):
        h[i * stride:i * stride + len(f)] = np.frombuffer(f, np.uint8)
    d16 = torch.from_numpy(h).cuda()
    d_in = d16.repeat((n + 15) // 16)[: n * stride].contiguous()
    d_out = torch.empty(n * cb, dtype=torch.uint8, device="cuda")
    in_off = np.arange(n, dtype=np.uint64) * stride
    in_len = np.array([len(frames[i % 16]) for i in range(n)], np.uint64)
    out_off = np.arange(n, dtype=np.uint64) * cb; out_cap = np.full(n, cb, np.uint64)
    torch.cuda.synchronize()
    for bsz in sizes:
        sl = slice(0, bsz)
        ctx.decompress_device(d_in.data_ptr(), in_off[sl], in_len[sl], d_out.data_ptr(), out_off[sl], out_cap[sl])
        ctx.decompress_ms(reset=True)
        R = 3
        for _ in range(R):
            ctx.decompress_device(d_in.data_ptr(), in_off[sl], in_len[sl], d_out.data_ptr(), out_off[sl], out_cap[sl])
        ms = ctx.decompress_ms() / R
        ok = bool(torch.equal(d_out[: 16 * cb].cpu(), torch.from_numpy(unit))) if bsz >= 16 else None
        out[f"{kind}, {bsz} frames"] = {"ms": round(ms, 3), "GB_per_s_of_output": round(bsz * cb / (ms / 1e3) / 1e9, 2), "decoded_ok": ok}
        print(f"{kind:30s} {bsz:5d} frames: {ms:9.2f} ms  {bsz * cb / (ms / 1e3) / 1e9:8.2f} GB/s of output  ok={ok}", flush=True)
print(json.dumps(out))
