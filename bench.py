#!/usr/bin/env python3
"""bench.py -- compress+hash stage throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (LZ4 frame + MD5 per 8 MiB chunk) over a device-resident synthetic
stream: the Silesia-like stand-in of SURVEY.md 8d item 2 (a 256 MiB unit tiled with per-tile rotations).
Inputs are resident in HBM before the timed region; PCIe is excluded (see DESIGN.md for the host-inclusive rate).

N GPUs = N independent ranks, chunks sharded round-robin, no collective on the data path (weak scaling: every
rank processes its own --chunks chunks).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (must be imported before libskyhip so both share one HIP runtime)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(host_stream: np.ndarray, chunk_bytes: int, budget_s: float = 12.0):
    """The reference CPU path restated exactly: per chunk lz4.frame.compress(data) (system liblz4, python-lz4
    default preferences; gateway_operator.py:359) then hashlib.md5(data).digest() (s3_interface.py:181-192),
    on a bounded sample of the same stream, one thread per host core."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    from oracle import ref

    cores = os.cpu_count() or 1
    n_avail = host_stream.size // chunk_bytes
    bound = ref.lz4f_frame_bound(chunk_bytes)

    def work(tid, deadline, single):
        out = np.empty(bound, np.uint8)
        done = 0
        i = tid
        while time.perf_counter() < deadline:
            a = host_stream[(i % n_avail) * chunk_bytes:(i % n_avail + 1) * chunk_bytes]
            ref.lz4f_compress_into(a, out)
            hashlib.md5(a).digest()
            done += 1
            i += 1 if single else cores
        return done

    # single core
    t0 = time.perf_counter()
    n1 = work(0, t0 + budget_s / 3, True)
    t1 = time.perf_counter() - t0
    one = n1 * chunk_bytes / t1 / 2**30
    # all cores (ctypes and hashlib both release the GIL)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        futs = [ex.submit(work, t, t0 + 2 * budget_s / 3, False) for t in range(cores)]
        nall = sum(f.result() for f in futs)
    tall = time.perf_counter() - t0
    allc = nall * chunk_bytes / tall / 2**30
    return {"value": round(allc, 3), "unit": "GiB/s", "cores": cores, "kind": "reference",
            "sample": f"{nall} x 8 MiB chunks of the same stream in {tall:.1f}s, liblz4 {ref.liblz4_version()} LZ4F_compressFrame + hashlib.md5, {cores} threads",
            "value_1core": round(one, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=0, help="8 MiB chunks per GPU per step (0 = 8192 = 64 GiB if it fits)")
    ap.add_argument("--unit-mib", type=int, default=256)
    ap.add_argument("--max-batch", type=int, default=2048, help="chunks per LZ4 sub-batch (scratch = 8.06 MiB each; fewer, larger launches = fewer launch tails)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream", choices=["silesia", "mixed"], default="silesia",
                    help="silesia = configs[1] stand-in (default); mixed = configs[3] stream: every chunk one of {random, text, records+binary, sparse}")
    ap.add_argument("--cdc", action="store_true", help="configs[2]: add Gear CDC + segment fingerprints + dedup table on a 50 %%-duplicate stream")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from skyplane_amd import hip_ops, synth

    cb = synth.CHUNK_BYTES
    free_b, total_b = torch.cuda.mem_get_info(dev)
    bound = hip_ops.frame_bound(cb)
    stride = (bound + 255) & ~255
    scratch = args.max_batch * 128 * 66048
    n_chunks = args.chunks or 8192
    while n_chunks > 64 and n_chunks * (cb + stride) + scratch + (6 << 30) > free_b:
        n_chunks //= 2
    if world > 1:   # every rank must process the same number of chunks (weak scaling, value = world x chunks x bytes)
        nc = torch.tensor([n_chunks], dtype=torch.int64, device=dev)
        dist.all_reduce(nc, op=dist.ReduceOp.MIN)
        n_chunks = int(nc.item())
    unit_bytes = min(args.unit_mib << 20, n_chunks * cb)
    unit_bytes -= unit_bytes % cb

    # ---- synthetic stream: unit generated on the host (deterministic), tiled + rotated on the device ----
    t0 = time.perf_counter()
    if args.cdc:
        unit = synth.dedup_stream(unit_bytes, dup_fraction=0.5, config_id=3)
    elif args.stream == "mixed":
        unit = synth.mixed_chunks(unit_bytes // cb, cb, config_id=4).reshape(-1)
    else:
        unit = synth.silesia_like(unit_bytes, config_id=2)
    d_unit = torch.from_numpy(unit).to(dev)
    d_in = torch.empty(n_chunks * cb, dtype=torch.uint8, device=dev)
    per = unit_bytes // cb
    n_tiles = (n_chunks + per - 1) // per
    rots = []
    for t in range(n_tiles):
        # rank-dependent rotation so ranks do not hold byte-identical shards
        rot = ((t + 131 * rank) * 7919 * 4096 + t * 13) % unit_bytes
        rots.append(rot)
        lo = t * unit_bytes
        hi = min(lo + unit_bytes, n_chunks * cb)
        tile = torch.roll(d_unit, -rot)[: hi - lo]
        if args.cdc and t:
            tile = tile ^ (t & 0xFF)      # keep the unit's internal duplicate structure, make tiles mutually distinct
        d_in[lo:hi] = tile
    d_out = torch.empty(n_chunks * stride, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    gen_s = time.perf_counter() - t0

    in_off = np.arange(n_chunks, dtype=np.uint64) * cb
    in_len = np.full(n_chunks, cb, np.uint64)
    out_off = np.arange(n_chunks, dtype=np.uint64) * stride
    out_cap = np.full(n_chunks, stride, np.uint64)

    ctx = hip_ops.SkyHipContext(device_id=local_rank, max_chunk_bytes=cb, max_batch=args.max_batch)
    flags = hip_ops.F_LZ4 | hip_ops.F_MD5 | ((hip_ops.F_CDC | hip_ops.F_DEDUP) if args.cdc else 0)

    def step():
        if args.cdc:
            ctx.dedup_reset()      # every step sees the stream for the first time
        return ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap, flags)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_len, md5 = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tm = ctx.timing()

    if rank == 0:
        total_bytes = world * n_chunks * cb * args.steps
        value = total_bytes / elapsed / 2**30
        comp_bytes = int(out_len.sum())
        # dominant kernel: sky_lz4_compress.  algorithmic bytes per launch = raw bytes read + frame bytes produced
        # (N + C per chunk; the 16-byte digest and cut offsets are negligible), divided by the HIP-event
        # duration of that kernel on the library's LZ4 stream.
        lz4_s = tm.lz4_ms / 1e3
        achieved = (tm.lz4_in_bytes + tm.lz4_out_bytes) / lz4_s / 1e9 if lz4_s > 0 else 0.0
        res = {
            "metric": "GiB/s through compress+hash stage (input bytes)", "value": round(value, 3), "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[1]: 1 MI355X, 8 MiB chunks, LZ4 frame + MD5 HIP kernels, Silesia-like synthetic stream "
                                   f"({n_chunks * cb / 2**30:.0f} GiB/GPU = {n_chunks} chunks, {unit_bytes >> 20} MiB unit tiled+rotated; no Silesia corpus offline)",
                       "chunk_bytes": cb, "chunks_per_gpu": n_chunks, "lz4_ratio": round(n_chunks * cb / comp_bytes, 4),
                       "sharding": "round-robin, no collective" if world > 1 else "single GPU", "max_batch": args.max_batch},
            "roofline": {"bound": "hbm", "kernel": "sky_lz4_compress", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
                         "avg_launch_ms": round(tm.lz4_ms / max(tm.lz4_launches, 1), 4), "launches": int(tm.lz4_launches),
                         "algorithmic_bytes_per_launch": int((tm.lz4_in_bytes + tm.lz4_out_bytes) / max(tm.lz4_launches, 1))},
            "kernels_ms_per_step": {"lz4": round(tm.lz4_ms / args.steps, 3), "layout": round(tm.layout_ms / args.steps, 3),
                                    "gather": round(tm.gather_ms / args.steps, 3), "md5": round(tm.md5_ms / args.steps, 3)},
            "setup_s": round(gen_s, 1),
        }
        # HBM-side traffic of the dominant kernel: PMC FETCH_SIZE / WRITE_SIZE cannot be collected inside this run
        # (separate rocprofv3 --pmc passes, scripts/pmc.sh); the committed per-input-byte factors are applied here.
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            t = json.loads(tf.read_text())
            per_launch_in = tm.lz4_in_bytes / max(tm.lz4_launches, 1)
            res["roofline"]["traffic"] = int(per_launch_in * (t["fetch_bytes_per_input_byte"] + t["write_bytes_per_input_byte"]))
            res["roofline"]["traffic_source"] = t["source"]
        if args.stream == "mixed" and not args.cdc:
            res["config"]["workload"] = (res["config"]["workload"].replace("configs[1]", "configs[3] stream on one GPU")
                                         .replace("Silesia-like synthetic stream", "mixed-compressibility stream (per chunk one of random / text / records+binary / sparse)"))
        if args.cdc:
            res["config"]["workload"] = (res["config"]["workload"].replace("configs[1]", "configs[2] (+ Gear CDC, segment MD5 fingerprints, dedup table)")
                                         .replace("Silesia-like synthetic stream", "synthetic stream with ~50 % of its 8-64 KiB spans copied from earlier spans at unaligned offsets")
                                         .replace("tiled+rotated; no Silesia corpus offline", "tiled, rotated and XOR-ed per tile so that tiles are mutually distinct"))
            res["kernels_ms_per_step"]["cdc"] = round(tm.cdc_ms / args.steps, 3)
            prefix, cuts, fps, first, base = ctx.cdc_results(n_chunks, in_len)
            import numpy as _np
            seg_end = cuts.astype(_np.int64)
            seg_start = _np.concatenate([[0], seg_end[:-1]])
            seg_start[prefix[:-1][prefix[:-1] < prefix[1:]].astype(_np.int64)] = 0      # first segment of every chunk starts at 0
            seg_len = seg_end - seg_start
            dup = first != _np.arange(base, base + len(first), dtype=_np.uint64)
            res["config"]["segments"] = int(len(first)); res["config"]["avg_segment_bytes"] = round(float(seg_len.mean()), 1)
            res["config"]["duplicate_bytes_fraction"] = round(float(seg_len[dup].sum() / seg_len.sum()), 4)
        # spot check (outside the timed region): first and last chunk against the oracle
        import hashlib

        from oracle import ref

        for i in (0, n_chunks - 1):
            t = i * cb // unit_bytes
            raw = np.roll(unit, -rots[t])[(i * cb) % unit_bytes:(i * cb) % unit_bytes + cb]
            if args.cdc and t:
                raw = raw ^ np.uint8(t & 0xFF)
            assert md5[i].tobytes() == hashlib.md5(raw).digest(), f"bench spot check: md5 of chunk {i}"
            f = d_out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].cpu().numpy()
            assert ref.lz4f_decompress(f, cb) == raw.tobytes(), f"bench spot check: frame of chunk {i}"
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(unit, cb)
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
