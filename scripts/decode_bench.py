#!/usr/bin/env python3
"""Measurement of the destination side's hot loop: LZ4 frame decompression on the GPU, in the form of bench.py's line.

  python scripts/decode_bench.py [--frames 1024] [--steps 5] [--kind ours|liblz4|linked] [--no-cpu-baseline]

Replaces  lz4.frame.decompress(to_write)  of skyplane/gateway/operators/gateway_receiver.py:195-201.  16 distinct 8 MiB chunks of bench.py's
Silesia-like stream are compressed (by this library: block-independent frames; or by liblz4 with python-lz4's defaults: the block-LINKED frames a stock
reference sender puts on the wire), replicated to --frames frames resident in HBM, and decoded --steps times by skyhip_decompress_device; every distinct
frame's output is compared with its chunk.  One JSON line: GB/s of decoded output, the HBM roofline at the algorithmic traffic (frame bytes read + decoded
bytes written per launch; the launch time is the library's own hipEvent pair around scan + decode), and liblz4's LZ4F_decompress on the host cores beside it."""
import argparse
from pathlib import Path
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ref
from skyplane_amd import hip_ops, synth


def _cpu_worker(args):
    frames, raw_len, budget = args
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < budget:
        for f in frames:
            ref.lz4f_decompress(f, raw_len); n += 1
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--kind", choices=["ours", "liblz4", "linked"], default="ours")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    cb, n = synth.CHUNK_BYTES, a.frames
    unit = synth.silesia_like(16 * cb, config_id=2)
    chunks = [unit[i * cb:(i + 1) * cb] for i in range(16)]
    ctx = hip_ops.SkyHipContext(0, cb, 64)
    if a.kind == "ours":
        frames = [r.frame for r in ctx.process_batch([c.tobytes() for c in chunks], flags=hip_ops.F_LZ4)]
    else:
        frames = [ref.lz4f_compress(c, block_linked=(a.kind == "linked")) for c in chunks]
    stride = (max(len(f) for f in frames) + 255) & ~255
    h = np.zeros(16 * stride, np.uint8)
    for i, f in enumerate(frames):
        h[i * stride:i * stride + len(f)] = np.frombuffer(f, np.uint8)
    d_in = torch.from_numpy(h).cuda().repeat((n + 15) // 16)[: n * stride].contiguous()
    d_out = torch.empty(n * cb, dtype=torch.uint8, device="cuda")
    in_off = np.arange(n, dtype=np.uint64) * stride
    in_len = np.array([len(frames[i % 16]) for i in range(n)], np.uint64)
    out_off = np.arange(n, dtype=np.uint64) * cb
    out_cap = np.full(n, cb, np.uint64)
    torch.cuda.synchronize()
    for _ in range(a.warmup):
        ctx.decompress_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap)
    ctx.decompress_ms(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.decompress_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / a.steps
    kern = ctx.decompress_ms() / a.steps / 1e3
    ok = bool(torch.equal(d_out[: min(n, 16) * cb].cpu(), torch.from_numpy(unit[: min(n, 16) * cb])))
    raw, comp = n * cb, int(in_len.sum())
    line = {"metric": "GB/s of decoded output through LZ4 frame decompression", "value": round(raw / wall / 1e9, 2), "unit": "GB/s", "n_gpus": 1, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(wall * 1e3, 3), "higher_is_better": True, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{n} LZ4 frames of 8 MiB chunks of the Silesia-like stream resident in HBM, 16 distinct", "frames": a.kind, "frame_bytes": comp,
                       "lz4_ratio": round(raw / comp, 4)},
            "roofline": {"bound": "hbm", "kernel": "sky_lz4f_scan + " + ("sky_lz4_decode" if a.kind != "linked" else
                                                                    "sky_lz4_parse + " + ("sky_lz4_resolve + sky_lz4_chain" if n <= int(os.environ.get("SKYHIP_LINK_RESOLVE_MAX", "192")) else "sky_lz4_link")),
                         "achieved": round((raw + comp) / kern / 1e9, 2), "peak": 8000.0, "unit": "GB/s", "frac": round((raw + comp) / kern / 8e12, 5), "traffic": None,
                         "launch_ms": round(kern * 1e3, 3), "algorithmic_bytes_per_launch": raw + comp},
            "verified": {"outputs_equal_chunks": ok, "checked": min(n, 16)}}
    # HBM-side traffic of the decode kernels: a committed rocprofv3 --pmc measurement of THIS command (scripts/dev/pmc_decode.sh -> profiles/traffic_decode.json),
    # quoted only for the frame kind and batch size it was taken on
    tf = Path(__file__).resolve().parents[1] / "profiles" / "traffic_decode.json"
    if tf.exists():
        t = json.loads(tf.read_text()).get(f"{a.kind}:{n}")
        if t:
            line["roofline"]["traffic"] = int(raw * t["bytes_per_output_byte"])
            line["roofline"]["traffic_source"] = t["source"]
    if not a.no_cpu_baseline:
        import multiprocessing as mp
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        try:
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            if q[0] != "max":
                cores = max(1, min(cores, int(int(q[0]) / int(q[1]))))
        except OSError:
            pass
        fb = [bytes(f) for f in frames[:4]]
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(fb, cb, 6.0)] * cores)
        tot = sum(r[0] for r in res); el = max(r[1] for r in res)
        one = _cpu_worker((fb, cb, 3.0))
        line["cpu_baseline"] = {"value": round(tot * cb / el / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "reference",
                                "sample": f"{tot} x LZ4F_decompress (liblz4 {ref.liblz4_version()}, what lz4.frame.decompress calls) of 4 of the same frames in {el:.1f} s, one process per schedulable core",
                                "value_1core": round(one[0] * cb / one[1] / 1e9, 3)}
    print(json.dumps(line))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
