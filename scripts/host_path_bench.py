#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (skyhip_process_batch / skyhip_decompress_batch_md5), the ones the gateway
operators call.

    python scripts/host_path_bench.py [--chunks 256] [--max-batch 64] [--reps 3] [--lanes 3]

Reports GiB/s of raw chunk bytes for
  * one call over all chunks: pageable numpy buffers and pinned buffers (skyhip_host_alloc), LZ4+MD5 and LZ4 only;
  * `--lanes` contexts in threads, each calling over its own share of the chunks at the same time -- what the operator's pipeline_depth does
    (gateway_operator.py here): whole-chunk MD5 is a serial chain of ~80 ms per chunk however many chunks a call holds, so one call's tail
    can only be hidden behind another call's upload;
  * decode (+ digest of the decoded bytes) of the frames just produced, one call and in lanes.
Every frame is checked against the input through the decode leg.  One JSON line."""
import argparse
import hashlib
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402  (first, so that libskyhip shares its HIP runtime)

from skyplane_amd import hip_ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=256)
ap.add_argument("--max-batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--skip-pageable", action="store_true")
a = ap.parse_args()

CB = synth.CHUNK_BYTES
unit = synth.silesia_like(32 * CB, config_id=2)
res = {"what": "skyhip_process_batch / skyhip_decompress_batch_md5, host buffers, PCIe included", "chunks": a.chunks, "chunk_bytes": CB,
       "max_batch": a.max_batch, "lanes": a.lanes}


def timed(fn, reps):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def in_lanes(fns):
    ths = [threading.Thread(target=f) for f in fns]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


ctxs = [hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=CB, max_batch=a.max_batch) for _ in range(max(1, a.lanes))]
try:
    c = ctxs[0]
    bound = c.frame_bound(CB)
    stride = (bound + 255) & ~255
    pin_in = c.pinned_buffer(a.chunks * CB)
    for i in range(a.chunks):
        pin_in[i * CB:(i + 1) * CB] = unit[(i % 32) * CB:((i % 32) + 1) * CB]
    pin_out = c.pinned_buffer(a.chunks * stride)
    pin_dec = c.pinned_buffer(a.chunks * CB)
    bufs = [("pinned", pin_in, pin_out)]
    if not a.skip_pageable:
        bufs.insert(0, ("pageable", np.array(pin_in), np.empty(a.chunks * stride, np.uint8)))
    GiB = a.chunks * CB / 2**30
    for name, bi, bo in bufs:
        vin = [bi[i * CB:(i + 1) * CB] for i in range(a.chunks)]
        vout = [bo[i * stride:i * stride + bound] for i in range(a.chunks)]
        for flags, tag in ((3, "lz4_md5"), (1, "lz4")):
            c.process_batch(vin[: a.max_batch], flags=flags, frames_into=vout[: a.max_batch])      # warm-up: allocations
            best = timed(lambda: c.process_batch(vin, flags=flags, frames_into=vout), a.reps)
            res[f"{name}_{tag}_gib_s"] = round(GiB / best, 2)
            res[f"{name}_{tag}_ms"] = round(best * 1e3, 1)

    # ---- lanes: contexts in threads over disjoint shares (pinned buffers) ----
    vin = [pin_in[i * CB:(i + 1) * CB] for i in range(a.chunks)]
    vout = [pin_out[i * stride:i * stride + bound] for i in range(a.chunks)]
    vdec = [pin_dec[i * CB:(i + 1) * CB] for i in range(a.chunks)]
    L = len(ctxs)
    share = [(k * a.chunks // L, (k + 1) * a.chunks // L) for k in range(L)]
    results = [None] * L

    def comp(k, flags):
        lo, hi = share[k]
        results[k] = ctxs[k].process_batch(vin[lo:hi], flags=flags, frames_into=vout[lo:hi])

    for flags, tag in ((3, "lz4_md5"), (1, "lz4")):
        in_lanes([lambda k=k: comp(k, flags) for k in range(L)])          # warm-up of every context
        best = timed(lambda: in_lanes([lambda k=k: comp(k, flags) for k in range(L)]), a.reps)
        res[f"lanes_{tag}_gib_s"] = round(GiB / best, 2)
        res[f"lanes_{tag}_ms"] = round(best * 1e3, 1)
    in_lanes([lambda k=k: comp(k, 3) for k in range(L)])
    r = [x for part in results for x in part]
    frames = [x.frame for x in r]
    res["ratio"] = round(a.chunks * CB / sum(len(f) for f in frames), 3)

    # ---- decode of those frames (+ digests of the decoded bytes) ----
    raw = [CB] * a.chunks
    for want_md5, tag in ((True, "decode_md5"), (False, "decode")):
        c.decompress_batch(frames[: a.max_batch], raw[: a.max_batch], want_md5=want_md5, into=vdec[: a.max_batch])
        best = timed(lambda: c.decompress_batch(frames, raw, want_md5=want_md5, into=vdec), a.reps)
        res[f"pinned_{tag}_gib_s"] = round(GiB / best, 2)
        res[f"pinned_{tag}_ms"] = round(best * 1e3, 1)

        def dec(k):
            lo, hi = share[k]
            results[k] = ctxs[k].decompress_batch(frames[lo:hi], raw[lo:hi], want_md5=want_md5, into=vdec[lo:hi])

        in_lanes([lambda k=k: dec(k) for k in range(L)])
        best = timed(lambda: in_lanes([lambda k=k: dec(k) for k in range(L)]), a.reps)
        res[f"lanes_{tag}_gib_s"] = round(GiB / best, 2)
        res[f"lanes_{tag}_ms"] = round(best * 1e3, 1)
    # round trip: decoded bytes == input, device digests (compress side) == hashlib of the input
    pin_dec[:] = 0
    outs, digs = c.decompress_batch(frames, raw, want_md5=True, into=vdec)
    ok = bool(np.array_equal(pin_dec, pin_in))
    step = max(1, a.chunks // 16)
    ok_md5 = all(r[i].md5 == hashlib.md5(vin[i]).digest() == digs[i] for i in range(0, a.chunks, step))
    res["round_trip_ok"] = ok
    res["digests_ok"] = bool(ok_md5)
finally:
    for x in ctxs:
        x.close()
print(json.dumps(res))
sys.exit(0 if res.get("round_trip_ok") and res.get("digests_ok") else 1)
