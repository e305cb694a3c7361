#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (skyhip_process_batch), the one the gateway operator calls.

    python scripts/host_path_bench.py [--chunks 256] [--max-batch 64] [--reps 3]

Reports GiB/s of input for (a) pageable numpy buffers, (b) pinned buffers from skyhip_host_alloc, each with LZ4+MD5 and
LZ4 only (whole-chunk MD5 is a ~0.1 s serial chain per sub-batch whatever its size).  One JSON line."""
import argparse
import json
import sys
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
a = ap.parse_args()

CB = synth.CHUNK_BYTES
unit = synth.silesia_like(32 * CB, config_id=2)
res = {"what": "skyhip_process_batch, host buffers, PCIe included", "chunks": a.chunks, "chunk_bytes": CB, "max_batch": a.max_batch}
with hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=CB, max_batch=a.max_batch) as c:
    bound = c.frame_bound(CB)
    stride = (bound + 255) & ~255
    pageable_in = np.empty(a.chunks * CB, np.uint8)
    for i in range(a.chunks):
        pageable_in[i * CB:(i + 1) * CB] = unit[(i % 32) * CB:((i % 32) + 1) * CB]
    pageable_out = np.empty(a.chunks * stride, np.uint8)
    pin_in = c.pinned_buffer(a.chunks * CB)
    pin_in[:] = pageable_in
    pin_out = c.pinned_buffer(a.chunks * stride)
    for name, bi, bo in (("pageable", pageable_in, pageable_out), ("pinned", pin_in, pin_out)):
        vin = [bi[i * CB:(i + 1) * CB] for i in range(a.chunks)]
        vout = [bo[i * stride:i * stride + bound] for i in range(a.chunks)]
        for flags, tag in ((3, "lz4_md5"), (1, "lz4")):
            c.process_batch(vin[: a.max_batch], flags=flags, frames_into=vout[: a.max_batch])      # warm-up: allocations
            best = 1e9
            for _ in range(a.reps):
                t0 = time.perf_counter()
                r = c.process_batch(vin, flags=flags, frames_into=vout)
                best = min(best, time.perf_counter() - t0)
            res[f"{name}_{tag}_gib_s"] = round(a.chunks * CB / best / 2**30, 2)
            res[f"{name}_{tag}_ms"] = round(best * 1e3, 1)
        res["ratio"] = round(a.chunks * CB / sum(len(x.frame) for x in r), 3)
print(json.dumps(res))
