#!/usr/bin/env python3
"""bench.py -- compress+hash stage throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (LZ4 frame + MD5 per 8 MiB chunk) over a device-resident synthetic stream.
  * 1 GPU   : BASELINE configs[1] -- 8192 x 8 MiB = 64 GiB of the Silesia-like stand-in.
  * N GPUs  : BASELINE configs[3] -- the mixed-compressibility stream (per chunk one of random / text / records+binary /
              sparse), 16384 chunks = 128 GiB per GPU, processed as two resident halves.  N independent ranks, chunk i of
              the node's queue on rank i % N, no collective on the data path (weak scaling).
Inputs are resident in HBM before the timed region; PCIe is excluded (DESIGN.md has the host-inclusive rate).
Rank 0 prints ONE JSON line.  After the timed region every digest of the step is checked against hashlib and every
frame still resident against liblz4 (the reference's decoder), on a pool of CPU processes forked before HIP is
initialised; the same pool times the reference's CPU path (cpu_baseline: one PROCESS per schedulable core).
The default 1-GPU run (no workload flags) also carries a `secondary` object: short verified runs of configs[2]
(`--cdc`) and of the configs[3] stream on one GPU (`--stream mixed --chunks 16384`), each its own process BEFORE this one
touches the GPU, each with the same roofline / verified fields (`--no-secondary` skips them).
Progress goes to stderr as `[bench +seconds] ...` lines (a heartbeat: the JSON line is the only thing on stdout).

--context emu (tests only): the shipping kernel source under the CPU SIMT emulator, tensors on the host, gloo instead of
RCCL -- that is how tests/test_host_operator.py runs this file's multi-rank logic with world_size 2 on a box without GPUs.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

from skyplane_amd import shard, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

# ---------------------------------------------------------------------------------------------------------------------
# CPU side: a pool of worker processes forked BEFORE the HIP runtime exists in this process.  They inherit the synthetic
# unit (copy-on-write) and do two jobs: the whole-stream verification and the reference CPU baseline.
# ---------------------------------------------------------------------------------------------------------------------
_G = {}


def schedulable_cores():
    n = len(os.sched_getaffinity(0))
    quota = None
    try:    # cgroup v2 CPU quota, if any
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        pass
    return n, quota


def chunk_view(i, rots, xor_tiles):
    """Chunk i of the resident stream as (piece, piece) of the host unit: the stream is the unit tiled with a rotation per
    tile (and, for the dedup stream, XOR-ed with the tile index so that tiles are mutually distinct)."""
    unit, cb = _G["unit"], _G["cb"]
    U = unit.size
    t = (i * cb) // U
    o = ((i * cb) % U + rots[t]) % U
    a = unit[o:o + cb]
    b = unit[:cb - a.size] if a.size < cb else unit[:0]
    if xor_tiles and (t & 0xFF):
        a = a ^ np.uint8(t & 0xFF)
        b = b ^ np.uint8(t & 0xFF)
    return a, b


def _w_md5(args):
    lo, hi, rots, xor_tiles = args
    out = []
    for i in range(lo, hi):
        a, b = chunk_view(i, rots, xor_tiles)
        h = hashlib.md5(a)
        if b.size:
            h.update(b)
        out.append(h.digest())
    return lo, out


def _w_frames(args):
    """Frames of one verification round (packed in the shared arena `which`): each must decode, with liblz4 -- the library behind the
    reference's lz4.frame.decompress (gateway_receiver.py:195-201) --, to exactly the chunk it was made from."""
    which, items, rots, xor_tiles = args
    from oracle import ref

    arena, cb = _G["arena"][which], _G["cb"]
    for i, off, ln in items:
        a, b = chunk_view(i, rots, xor_tiles)
        dec = ref.lz4f_decompress(arena[off:off + ln], cb)
        if len(dec) != cb or dec[:a.size] != a.tobytes() or dec[a.size:] != b.tobytes():
            return i
    return -1


def _w_baseline(args):
    """The reference CPU path restated exactly: per chunk lz4.frame.compress(data) (system liblz4 through oracle.ref, python-lz4's
    default preferences; gateway_operator.py:359) then hashlib.md5(data).digest() (s3_interface.py:181-192)."""
    wid, nworkers, seconds, rots = args
    from oracle import ref

    cb = _G["cb"]
    n_avail = _G["unit"].size // cb
    out = np.empty(ref.lz4f_frame_bound(cb), np.uint8)
    buf = np.empty(cb, np.uint8)
    done, i = 0, wid
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        a, b = chunk_view(i % n_avail, rots, False)
        buf[:a.size] = a
        buf[a.size:] = b
        ref.lz4f_compress_into(buf, out)
        hashlib.md5(buf).digest()
        done += 1
        i += nworkers
    return done, time.perf_counter() - t0


def cpu_baseline(pool, cores, quota, rots, budget_s):
    from oracle import ref

    cb = _G["cb"]
    (n1, t1), = pool.map(_w_baseline, [(0, 1, budget_s / 3, rots)])
    one = n1 * cb / t1 / 2**30
    t0 = time.perf_counter()
    res = pool.map(_w_baseline, [(w, cores, 2 * budget_s / 3, rots) for w in range(cores)], chunksize=1)
    wall = time.perf_counter() - t0
    nall = sum(r[0] for r in res)
    allc = nall * cb / max(r[1] for r in res) / 2**30
    return {"value": round(allc, 3), "unit": "GiB/s", "cores": cores, "kind": "reference",
            "sample": f"{nall} x {cb >> 20} MiB chunks of the same stream in {wall:.1f}s: liblz4 {ref.liblz4_version()} LZ4F_compressFrame (python-lz4 default "
                      f"preferences) + hashlib.md5, one process per schedulable core ({cores} = min(affinity mask {len(os.sched_getaffinity(0))}, cgroup cpu quota {quota if quota else 'none'}))",
            "value_1core": round(one, 3), "scaling_efficiency": round(allc / (cores * one), 3) if one > 0 else None}


# ---------------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------------
class EmuContext:
    """--context emu: the shipping kernels' source under the CPU emulator (tests/emu).  Same call shape as SkyHipContext."""

    def __init__(self):
        from tests.emu import emulib

        self.emulib = emulib
        emulib.lib()

    def process_device(self, t_in, in_off, in_len, t_out, out_off, out_cap, flags):
        chunks = [t_in[int(o):int(o) + int(l)].tobytes() for o, l in zip(in_off, in_len)]
        frames, md5s, _ = self.emulib.process(chunks, flags=flags & 3)
        out_len = np.zeros(len(chunks), np.uint64)
        for i, f in enumerate(frames):
            t_out[int(out_off[i]):int(out_off[i]) + len(f)] = np.frombuffer(f, np.uint8)
            out_len[i] = len(f)
        return out_len, np.frombuffer(b"".join(md5s), np.uint8).reshape(-1, 16).copy()

    def timing(self):
        from skyplane_amd._lib import Timing

        return Timing()

    def reset_timing(self):
        pass

    def dedup_reset(self):
        pass

    def close(self):
        pass


_T0 = time.perf_counter()


def log(msg):
    """heartbeat on stderr (stdout carries the one JSON line): a driver that waits on a long run sees where it is"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def run_secondary(extra, steps, warmup, script=None):
    """One of the other single-GPU measurements as its own process (own HIP context, own CPU pool): returns its JSON line as a dict, or the reason
    it has none.  script=None: this very file (the same verification, fewer steps); otherwise one of scripts/*.py that print one JSON line."""
    import subprocess

    if script is None:
        cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-secondary"] + extra
        shown = "python bench.py " + " ".join(cmd[2:])
    else:
        cmd = [sys.executable, str(ROOT / "scripts" / script)] + extra
        shown = f"python scripts/{script} " + " ".join(extra)
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=None, timeout=600 if script is None else 150,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
        line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": f"exit code {p.returncode}", "command": shown}
        r = json.loads(line[-1])
    except subprocess.TimeoutExpired:
        return {"error": "timeout", "command": shown}
    if script is None or "metric" in r:
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "kernels_ms_per_step", "verified", "cpu_baseline", "per_gpu")
        out = {k: r[k] for k in keep if k in r}
    else:
        out = r
    out["command"] = shown
    out["run_s"] = round(time.perf_counter() - t0, 1)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: re-execute this very command line as N ranks of one node under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 -- the container's hostname may not resolve)."""
    import socket

    with socket.socket() as sk:          # a free port chosen by the kernel
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=0, help="chunks per GPU per step (0 = the configuration's own: 8192 at 1 GPU, 16384 at N GPUs)")
    ap.add_argument("--unit-mib", type=int, default=256)
    ap.add_argument("--max-batch", type=int, default=1024, help="chunks per LZ4 launch on the block-queue path (device-resident batches below 2 chunks per CU and "
                                                                "SKYHIP_FRAMES_MIN=0 runs; block scratch = 8.06 MiB per chunk); the default run writes frames in place and never uses it")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short configs[2] / configs[3]-on-one-GPU runs appended to the default 1-GPU line")
    ap.add_argument("--secondary-steps", type=int, default=16)
    ap.add_argument("--secondary-budget-s", type=float, default=100.0, help="no further secondary run is STARTED once the ones before it took this long")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream", choices=["auto", "silesia", "mixed"], default="auto",
                    help="auto = silesia (configs[1]) on one GPU, mixed (configs[3]) on several")
    ap.add_argument("--cdc", action="store_true", help="configs[2]: add Gear CDC + segment fingerprints + dedup table on a 50 %%-duplicate stream")
    ap.add_argument("--verify", choices=["full", "sample", "none"], default="full", help="post-run check of digests (all / 64 chunks) and of sampled frames")
    ap.add_argument("--context", choices=["hip", "emu"], default="hip", help="emu = CPU emulator + gloo (tests of this file's rank logic only)")
    ap.add_argument("--chunk-bytes", type=int, default=synth.CHUNK_BYTES, help="tests only; the metric is defined on 8 MiB chunks")
    ap.add_argument("--md5-lanes", type=int, default=0, help="experiment: digest chains as their own device calls on this many extra contexts (0 = inside the "
                    "compressor's call, the default): see the comment at md5_lanes below")
    ap.add_argument("--depth", type=int, default=0, help="steps in flight (0 = 2 when a second set of frame slots fits beside the stream, else 1): "
                    "step k + 1's compressor runs while step k's digest chains finish, the way the gateway operator's lanes overlap their batches")
    ap.add_argument("--halves", type=int, default=0, help="tests only: resident halves per step (0 = 2 when the stream has more than 8192 chunks)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)      # bare `python bench.py --gpus N`: become N ranks (does not return)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    emu = args.context == "emu"
    cb = args.chunk_bytes
    stream = args.stream if args.stream != "auto" else ("silesia" if world == 1 else "mixed")
    n_target = args.chunks or (8192 if (stream == "silesia" or args.cdc or world == 1) else 16384)

    # ---- the other single-GPU configurations, in front of whoever runs the default command: each in its own process, BEFORE this process touches the
    # GPU.  (Round 5's first version ran them after the headline, with this process's HIP runtime still holding its hardware queues: the configs[2] run,
    # which asks for 16 queues of its own, then took 326 ms per step instead of 151-156 -- GPU call r5y -- : queues of two processes beyond what the
    # hardware holds are time-sliced, and kernels that are meant to run side by side take turns.)
    secondary_res = None
    if rank == 0 and world == 1 and not emu and not args.cdc and args.stream == "auto" and args.chunks == 0 and cb == synth.CHUNK_BYTES and not args.no_secondary:
        secondary_res = {}
        for key, extra in (("configs[2]", ["--cdc"]), ("configs[3] stream on one GPU", ["--stream", "mixed", "--chunks", "16384"])):
            log(f"secondary run {key}: bench.py {' '.join(extra)} --steps {args.secondary_steps}")
            secondary_res[key] = run_secondary(extra, args.secondary_steps, 1)
        # the rest of the path: the destination's decoder on this library's frames at a large batch and on the reference sender's default
        # (block-linked) frames at an operator's batch, and the host-buffer entry points the operators call (PCIe included) -- each verified by its script
        for key, script, extra in (("f1 decode: own frames x 1024", "decode_bench.py", ["--frames", "1024", "--kind", "ours", "--no-cpu-baseline"]),
                                   ("f1 decode: reference-default linked frames x 32", "decode_bench.py", ["--frames", "32", "--kind", "linked"]),
                                   ("host buffers (PCIe included) x 256", "host_path_bench.py", ["--chunks", "256", "--skip-pageable", "--reps", "2"])):
            if time.perf_counter() - _T0 > args.secondary_budget_s:
                secondary_res[key] = {"skipped": f"the secondary runs before it used up their {args.secondary_budget_s:.0f} s"}
                continue
            log(f"secondary run {key}: scripts/{script} {' '.join(extra)}")
            secondary_res[key] = run_secondary(extra, 0, 0, script=script)

    # ---- host unit (deterministic), then the CPU pool, then -- and only then -- the HIP runtime ----
    t0 = time.perf_counter()
    unit_bytes = min(args.unit_mib << 20, n_target * cb)
    unit_bytes -= unit_bytes % cb
    if args.cdc:
        unit = synth.dedup_stream(unit_bytes, dup_fraction=0.5, config_id=3)
    elif stream == "mixed":
        unit = synth.mixed_chunks(unit_bytes // cb, cb, config_id=4).reshape(-1)
    else:
        unit = synth.silesia_like(unit_bytes, config_id=2)
    _G["unit"], _G["cb"] = unit, cb
    cores, quota = schedulable_cores()
    if quota:       # a container with a CPU quota cannot run more than that many processes at once, whatever its affinity mask says
        cores = max(1, min(cores, int(quota + 0.5)))
    want_cpu = rank == 0 and not args.no_cpu_baseline
    pool_n = cores if want_cpu else max(1, cores // world)
    # two shared arenas (anonymous shared mappings, made before the fork) carry the frames of a verification round to the workers
    bound_h = 15 + cb + 4 * ((cb + 65535) // 65536) + 4
    round_n = max(1, min(128, (1 << 30) // bound_h))
    if args.verify != "none":
        import mmap

        _G["arena"] = [np.frombuffer(mmap.mmap(-1, round_n * bound_h), np.uint8) for _ in range(2)]
    pool = mp.get_context("fork").Pool(pool_n) if (args.verify != "none" or want_cpu) else None

    log(f"host unit of {unit_bytes >> 20} MiB generated, CPU pool of {pool_n} forked")
    # (configs[2] keeps four kernel streams of two contexts busy at once.  The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues --
    # default 4 -- round-robin, and kernels of streams that share a queue run one after the other; 8 / 16 / 24 / 32 queues were all measured for `--cdc`
    # (GPU calls r5g-r6f) and every one of them showed a second mode at 250-300 ms per step next to its 140-160: the default count is what is stable.)
    import torch  # noqa: E402  (imported before libskyhip so both share one HIP runtime; no device touched before the fork above)

    if emu:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    log("torch imported" + (f", process group of {world} up" if world > 1 else ""))
    from skyplane_amd import hip_ops

    bound = 15 + cb + 4 * ((cb + 65535) // 65536) + 4 if emu else hip_ops.frame_bound(cb)
    stride = (bound + 255) & ~255
    halves = args.halves or (2 if (n_target > 8192 and not args.cdc) else 1)          # frame slots are reused by the second half
    n_chunks = n_target
    depth = args.depth or (1 if (emu or halves > 1) else 2)
    if not emu:
        free_b, _total_b = torch.cuda.mem_get_info(dev)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        frames_min = int(os.environ.get("SKYHIP_FRAMES_MIN", 2 * cus))      # (the library's rule: skyhip.hip)

        def need(n, d, shared=True):
            in_place = frames_min > 0 and n >= frames_min
            scratch = 0 if in_place else int(2.5 * args.max_batch * 128 * 66048)      # block scratch, double-buffered, only on the block-queue path
            return n * cb + d * (n // (halves if shared else 1)) * stride + d * scratch + (6 << 30)      # (6 GiB: the library's metadata, RCCL's buffers, torch's cache)

        if depth > 1 and need(n_chunks, depth) > free_b:
            depth = 1
        while n_chunks > 64 and need(n_chunks, depth) > free_b:
            n_chunks //= 2
        # every chunk of a step gets its own frame slot when that fits (include/skyhip.h: frame regions of one call must not overlap); only when HBM is too
        # small for that do the halves of a step share slots (chunk i + n/2 over chunk i's: the queue hands chunks out in index order, a slot's first
        # frame is complete thousands of chunks before its second one is started -- and every resident frame is decoded and compared after the run)
        share_slots = halves > 1 and need(n_chunks, depth, shared=False) > free_b
        in_place = frames_min > 0 and n_chunks >= frames_min
    else:
        in_place = False
        share_slots = halves > 1
    if world > 1:   # every rank processes the same number of chunks (weak scaling: value = world x chunks x bytes / time)
        nc = torch.tensor([n_chunks], dtype=torch.int64, device=dev)
        dist.all_reduce(nc, op=dist.ReduceOp.MIN)
        n_chunks = int(nc.item())
    if world > 1:
        sh = torch.tensor([int(share_slots)], dtype=torch.int64, device=dev)
        dist.all_reduce(sh, op=dist.ReduceOp.MAX)
        share_slots = bool(sh.item())
    n_half = n_chunks // halves
    n_slots = n_half if share_slots else n_chunks

    # ---- the resident stream: the unit tiled and rotated on the device.  This rank holds the chunks rank, rank + world, ... of
    # the node's queue (chunk_index % n_gpus); with a synthetic queue that is a rank-dependent rotation per tile. ----
    d_unit = torch.from_numpy(unit).to(dev)
    d_in = torch.empty(n_chunks * cb, dtype=torch.uint8, device=dev)
    per = unit_bytes // cb
    n_tiles = (n_chunks + per - 1) // per
    rots = []
    for t in range(n_tiles):
        rot = (((t + 131 * rank) * 7919 * 4096 + t * 13) % unit_bytes) if unit_bytes > cb else 0
        rot -= rot % cb if stream == "mixed" and not args.cdc else 0     # mixed: the class is a property of the whole chunk -- rotate by whole chunks
        rots.append(rot)
        lo, hi = t * unit_bytes, min((t + 1) * unit_bytes, n_chunks * cb)
        tile = torch.roll(d_unit, -rot)[: hi - lo]
        if args.cdc and (t & 0xFF):
            tile = tile ^ (t & 0xFF)      # keep the unit's internal duplicate structure, make tiles mutually distinct
        d_in[lo:hi] = tile
    del d_unit
    d_outs = [torch.empty(n_slots * stride, dtype=torch.uint8, device=dev) for _ in range(depth)]
    d_out = d_outs[0]
    if not emu:
        torch.cuda.synchronize(dev)
    gen_s = time.perf_counter() - t0

    in_off = np.arange(n_chunks, dtype=np.uint64) * cb
    in_len = np.full(n_chunks, cb, np.uint64)
    out_off = np.arange(n_slots, dtype=np.uint64) * stride
    out_cap = np.full(n_slots, stride, np.uint64)
    out_off_all, out_cap_all = (np.tile(out_off, halves), np.tile(out_cap, halves)) if share_slots else (out_off, out_cap)

    if emu:
        ctxs = [EmuContext()]
        p_in, p_outs = d_in.numpy(), [d_out.numpy()]
    else:
        ctxs = [hip_ops.SkyHipContext(device_id=local_rank, max_chunk_bytes=cb, max_batch=args.max_batch) for _ in range(depth)]
        p_in, p_outs = d_in.data_ptr(), [t.data_ptr() for t in d_outs]
    ctx = ctxs[0]
    # Digest lanes (--md5-lanes M, an experiment switch; default 0 = the digests ride in the compressor's call): a step's digest chains as their OWN device
    # call (flags = MD5 only, own context and streams).  A chain needs no frame slots, so more steps' chains could be in flight than calls that compress.
    # Measured (GPU calls r5g-r5j): no gain -- 386 against 551 GiB/s on configs[1] at the default 4 hardware queues (more streams, more queue sharing), equal
    # at 16; configs[2] 308-397 against 385-423.  The chip's time per step is the SUM of what compressor, candidates and segment digests need alone (they
    # compete for the same LDS and issue slots), not the longest of them: deeper pipelining of the one latency-bound kernel buys nothing.
    md5_lanes = args.md5_lanes if args.md5_lanes > 0 else 0
    if emu:
        md5_lanes = 0
    flags = hip_ops.F_LZ4 | (hip_ops.F_MD5 if md5_lanes == 0 else 0) | ((hip_ops.F_CDC | hip_ops.F_DEDUP) if args.cdc else 0)
    lasts = [{} for _ in range(depth)]
    md5_ctxs = [hip_ops.SkyHipContext(device_id=local_rank, max_chunk_bytes=cb, max_batch=args.max_batch) for _ in range(md5_lanes)]
    md5_lasts = [{} for _ in range(md5_lanes)]

    def step(lane=0):
        if args.cdc:
            ctxs[lane].dedup_reset()      # every step sees the stream for the first time
        # ONE call for the whole resident stream: one MD5 launch over every chunk (a chain per chunk, all chains at once) beside the compressor
        # (out_off_all: a slot per chunk, or -- share_slots, when HBM is short -- the second half's frames over the first half's)
        lasts[lane]["out_len"], md5_ = ctxs[lane].process_device(p_in, in_off, in_len, p_outs[lane], out_off_all, out_cap_all, flags)
        if md5_lanes == 0:
            lasts[lane]["md5"] = md5_

    def md5_step(m):
        _ol, md5_lasts[m]["md5"] = md5_ctxs[m].process_device(p_in, in_off, in_len, 0, in_off, in_off, hip_ops.F_MD5)

    def run_steps(k_steps):
        """k_steps steps, `depth` of them in flight: lane j (its own context, streams and frame slots) runs steps j, j + depth, ...  A step's digest
        chains (one per chunk, ~82 ms when alone, longer beside the compressor) end after its compressor launch does; with a second step in flight the
        next compressor launch runs meanwhile instead of waiting for them."""
        if depth == 1 and md5_lanes == 0:
            for _ in range(k_steps):
                step(0)
            return
        import threading

        errs = []

        def lane_loop(j):
            try:
                for _ in range(j, k_steps, depth):
                    step(j)
            except BaseException as e:      # noqa: BLE001
                errs.append(e)

        def md5_loop(m):
            try:
                for _ in range(m, k_steps, md5_lanes):
                    md5_step(m)
            except BaseException as e:      # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=lane_loop, args=(j,)) for j in range(depth)] + [threading.Thread(target=md5_loop, args=(m,)) for m in range(md5_lanes)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    def sync():
        if not emu:
            torch.cuda.synchronize(dev)

    log(f"stream resident ({n_chunks} chunks of {cb} B per rank, {halves} half(s), depth {depth}, setup {gen_s:.1f}s); warmup")
    run_steps(args.warmup)
    if max(depth, md5_lanes) > 1 and args.warmup < max(depth, md5_lanes):
        run_steps(max(depth, md5_lanes) - args.warmup)      # every lane's context has run once before the clock starts
    for c_ in ctxs + md5_ctxs:
        c_.reset_timing()
    log(f"timed region: {args.steps} steps")
    _local, elapsed = shard.timed_region(lambda: run_steps(args.steps), 1, dist=dist, sync=sync)      # EXACTLY args.steps steps between barrier + synchronize, MAX over ranks
    tms = [c_.timing() for c_ in ctxs + md5_ctxs]

    class _Tm:
        pass

    tm = _Tm()
    for f in ("lz4_ms", "layout_ms", "gather_ms", "md5_ms", "cdc_ms", "lz4_launches", "lz4_in_bytes", "lz4_out_bytes", "md5_launches", "md5_in_bytes"):
        setattr(tm, f, sum(getattr(t, f) for t in tms))
    # configs[3] asks for per-GPU AND aggregate rates: every rank's own clock and kernel times travel to rank 0 (one all_gather of nine doubles, outside
    # the timed region); a slow GPU, or a rank whose streams share a hardware queue, shows up as its own row
    mine = [float(rank), _local, tm.lz4_ms, tm.md5_ms, tm.cdc_ms, float(tm.lz4_in_bytes), float(tm.lz4_out_bytes), float(max(tm.lz4_launches, 1)), float(n_chunks)]
    if world > 1:
        me = torch.tensor(mine, dtype=torch.float64, device=dev)
        rows = [torch.empty_like(me) for _ in range(world)]
        dist.all_gather(rows, me)
        per_rank = [r_.tolist() for r_ in rows]
    else:
        per_rank = [mine]
    last_lane = (args.steps - 1) % depth if args.steps >= depth else 0
    out_len = lasts[last_lane]["out_len"]
    md5 = lasts[last_lane]["md5"] if md5_lanes == 0 else md5_lasts[(args.steps - 1) % md5_lanes if args.steps >= md5_lanes else 0]["md5"]
    d_out = d_outs[last_lane]
    for j in range(depth):      # every lane hashed and compressed the same stream: same digests, same frame lengths
        if lasts[j]:
            assert (md5_lanes or np.array_equal(lasts[j]["md5"], md5)) and np.array_equal(lasts[j]["out_len"], out_len), f"lane {j} disagrees with lane {last_lane}"
    for m in range(md5_lanes):
        if md5_lasts[m]:
            assert np.array_equal(md5_lasts[m]["md5"], md5), f"digest lane {m} disagrees"
    # the second roofline (min(HBM, chain)): MD5 is one serial chain per chunk, so a step can never be shorter than one chain however
    # many lanes idle.  Measured, outside the timed region: the digest kernel alone over the same resident chunks.
    md5_alone_ms = None
    if not emu and rank == 0:
        ctx.reset_timing()
        ctx.process_device(p_in, in_off, in_len, p_outs[0], out_off_all, out_cap_all, hip_ops.F_MD5)
        md5_alone_ms = ctx.timing().md5_ms

    # ---- verification, outside the timed region: every digest of the last step, a sample of the frames of its last half ----
    verified = {"digests": 0, "frames": 0}
    if args.verify != "none":
        # what the check costs on this host (hashlib ~0.55 GiB/s and liblz4 decode ~2 GiB/s per process): said BEFORE it starts, so that a long
        # silence on stdout is never a mystery
        nv = n_chunks if args.verify == "full" else 64
        est = nv * cb / 2**30 / (0.5 * pool_n) + ((n_half if share_slots else n_chunks) if args.verify == "full" else 24) * cb / 2**30 / (1.5 * pool_n)
        log(f"timed region done ({elapsed / args.steps * 1e3:.1f} ms per step); verifying {nv} digests + frames on {pool_n} processes, about {est:.0f} s")
        idx = list(range(n_chunks)) if args.verify == "full" else sorted(set(np.linspace(0, n_chunks - 1, 64).astype(int).tolist()))
        spans = [(idx[k], idx[k] + 1) for k in range(len(idx))] if args.verify != "full" else [(lo, min(lo + 16, n_chunks)) for lo in range(0, n_chunks, 16)]
        for lo, digs in pool.imap_unordered(_w_md5, [(lo, hi, rots, args.cdc) for lo, hi in spans], chunksize=1):
            for k, d in enumerate(digs):
                assert md5[lo + k].tobytes() == d, f"rank {rank}: md5 of chunk {lo + k} differs from hashlib"
        verified["digests"] = len(idx)
        # every frame still resident (all of them; with two halves the second half, whose frames replaced the first's in the slots): D2H into
        # a shared arena round by round (double buffered), decoded and compared by the pool while the next round is copied
        n_res = n_half if share_slots else n_chunks      # frames still resident: every one, or (shared slots) the last half's
        first_res = n_chunks - n_res
        js = list(range(n_res)) if args.verify == "full" else sorted(set(np.linspace(0, n_res - 1, 24).astype(int).tolist()))
        arenas = [torch.from_numpy(a) for a in _G["arena"]]
        if not emu:
            for a in arenas:      # page-lock the arenas for asynchronous D2H (best effort: pageable copies work too)
                try:
                    torch.cuda.cudart().cudaHostRegister(a.data_ptr(), a.numel(), 0)
                except Exception:
                    pass
        pending = None

        def reap(p):
            for bad in p.get():
                assert bad < 0, f"rank {rank}: frame of chunk {bad} does not decode to the chunk (liblz4)"

        for r0 in range(0, len(js), round_n):
            which = (r0 // round_n) & 1
            items, o = [], 0
            for j in js[r0:r0 + round_n]:
                i = first_res + j
                ln = int(out_len[i])
                arenas[which][o:o + ln].copy_(d_out[int(out_off_all[i]):int(out_off_all[i]) + ln], non_blocking=True)
                items.append((i, o, ln)); o += ln
            sync()
            if pending is not None:
                reap(pending)
            per = max(1, (len(items) + pool_n - 1) // pool_n)
            pending = pool.map_async(_w_frames, [(which, items[k:k + per], rots, args.cdc) for k in range(0, len(items), per)], chunksize=1)
            verified["frames"] += len(items)
        if pending is not None:
            reap(pending)
    dd = hashlib.md5(md5.tobytes()).hexdigest()      # digest of digests of this rank's stream
    ok = 1
    if world > 1:
        okt = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)       # a failed assert on any rank kills the job before this line
        ok = int(okt.item())

    if rank == 0:
        total_bytes = world * n_chunks * cb * args.steps
        value = total_bytes / elapsed / 2**30
        comp_bytes = int(out_len.sum())
        cfg_id = "configs[2] (+ Gear CDC, segment MD5 fingerprints, dedup table)" if args.cdc else ("configs[1]" if stream == "silesia" else ("configs[3]" if world > 1 else "configs[3] stream on one GPU"))
        sdesc = ("synthetic stream with ~50 % of its 8-64 KiB spans copied from earlier spans at unaligned offsets" if args.cdc else
                 "Silesia-like synthetic stream (no Silesia corpus offline)" if stream == "silesia" else
                 "mixed-compressibility stream (per chunk one of random / text / records+binary / sparse)")
        # dominant kernel: the LZ4 compressor.  algorithmic bytes per launch = raw bytes read + frame bytes produced (N + C per
        # chunk; the 16-byte digest and cut offsets are negligible), divided by the HIP-event duration of that kernel on the library's stream.
        lz4_s = tm.lz4_ms / 1e3
        achieved = (tm.lz4_in_bytes + tm.lz4_out_bytes) / lz4_s / 1e9 if lz4_s > 0 else 0.0
        kname = "sky_lz4s_frames" if in_place else "sky_lz4s_compress"      # large device-resident batches: frames written in place (skyhip.hip)
        res = {
            "metric": "GiB/s through compress+hash stage (input bytes)", "value": round(value, 3), "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{cfg_id}: {world} MI355X, {cb >> 20} MiB chunks, LZ4 frame + MD5 HIP kernels, {sdesc}; "
                                   f"{n_chunks * cb / 2**30:.0f} GiB/GPU = {n_chunks} chunks per step in {halves} resident half(s){' sharing frame slots' if share_slots and halves > 1 else ''}, {unit_bytes >> 20} MiB unit tiled + rotated",
                       "chunk_bytes": cb, "chunks_per_gpu": n_chunks, "lz4_ratio": round(n_chunks * cb / comp_bytes, 4),
                       "sharding": "chunk i of the node's queue on rank i % N, no collective on the data path" if world > 1 else "single GPU", "max_batch": args.max_batch,
                       "steps_in_flight": depth, "digest_lanes": md5_lanes, "lz4_path": "frames written in place, one launch per step" if in_place else "block queue + frame gather, sub-batches of max_batch"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
                         "avg_launch_ms": round(tm.lz4_ms / max(tm.lz4_launches, 1), 4), "launches": int(tm.lz4_launches),
                         "algorithmic_bytes_per_launch": int((tm.lz4_in_bytes + tm.lz4_out_bytes) / max(tm.lz4_launches, 1))},
            "kernels_ms_per_step": {"lz4": round(tm.lz4_ms / args.steps, 3), "layout": round(tm.layout_ms / args.steps, 3),
                                    "gather": round(tm.gather_ms / args.steps, 3), "md5": round(tm.md5_ms / args.steps, 3)},
            "verified": {"digests_vs_hashlib": verified["digests"], "frames_vs_liblz4": verified["frames"], "digest_of_digests": dd, "all_ranks_ok": bool(ok)},
            "setup_s": round(gen_s, 1),
        }
        per_gpu = []
        for r_, loc, lz, md, cd, ib, ob, nl, nc_ in per_rank:
            a_ = (ib + ob) / (lz / 1e3) / 1e9 if lz > 0 else 0.0
            per_gpu.append({"rank": int(r_), "GiBps": round(nc_ * cb * args.steps / loc / 2**30, 4), "elapsed_s": round(loc, 4), "lz4_ms_per_step": round(lz / args.steps, 3),
                            "md5_ms_per_step": round(md / args.steps, 3), **({"cdc_ms_per_step": round(cd / args.steps, 3)} if args.cdc else {}),
                            "avg_launch_ms": round(lz / nl, 4), "frac": round(a_ / HBM_PEAK_GBPS, 5)})
        res["per_gpu"] = per_gpu
        res["aggregate"] = {"GiBps": res["value"], "sum_of_per_gpu_GiBps": round(sum(g["GiBps"] for g in per_gpu), 3),
                            "slowest_rank": int(min(per_gpu, key=lambda g: g["GiBps"])["rank"]), "timing": "value = all ranks' bytes / MAX over ranks of the barrier-to-barrier time"}
        if world > 1 or args.stream == "mixed":
            # The N = 1 point of THIS workload (a bare `--gpus 1` runs configs[1], another stream: dividing the N-GPU value by N x that number mixes two
            # workloads).  A scaling sweep's N = 1 run is `python bench.py --gpus 1 --stream mixed --chunks 16384`; the last such run committed with the
            # tree is quoted here so that the N-GPU line carries its own like-for-like reference.
            nf = ROOT / "profiles" / "n1_reference.json"
            if nf.exists() and not emu:
                n1 = json.loads(nf.read_text()).get("cdc" if args.cdc else stream)
                if n1:
                    res["n1_reference"] = {**n1, "quoted": True}
                    res["scaling_efficiency_vs_n1_reference"] = round(value / (world * n1["GiBps"]), 4)
        if md5_alone_ms:
            hbm_in = HBM_PEAK_GBPS * 1e9 / (1.0 + comp_bytes / (n_chunks * cb)) / 2**30      # input GiB/s at which N + C bytes saturate the HBM peak
            chain = n_chunks * cb / (md5_alone_ms / 1e3) / 2**30
            res["roofline"]["chain"] = {
                "kernel": "sky_md5_chunks", "md5_alone_ms": round(md5_alone_ms, 3), "per_chain_MBps": round(cb / (md5_alone_ms / 1e3) / 1e6, 2),
                "resident_chains": int(n_chunks * depth), "bound_GiBps": round(chain * depth, 1), "hbm_bound_GiBps": round(hbm_in, 1),
                "min_bound": "chain" if chain * depth < hbm_in else "hbm", "frac_of_min_bound": round(value / world / min(chain * depth, hbm_in), 4),
                "note": "whole-chunk MD5 is a serial dependency chain per chunk (RFC 1321): a chunk's digest cannot arrive sooner than one chain however many "
                        "lanes idle, and the stage's throughput cannot exceed resident chains x measured per-chain rate (rank 0's GPU); resident = the chunks of "
                        "the steps in flight"}
        # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (scripts/pmc.sh); a committed measurement is
        # quoted only for the kernel and stream it was taken on
        # which unit of the CU the dominant kernel keeps busy (it is not HBM): the committed SQ-counter summary of the shipping kernel, quoted
        bf = ROOT / "profiles" / "binding.json"
        if bf.exists() and not emu:
            bnd = json.loads(bf.read_text()).get(kname)
            if bnd:
                res["roofline"]["binding"] = {**bnd, "quoted": True}
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            t = json.loads(tf.read_text()).get(f"{kname}:{'cdc' if args.cdc else stream}")
            if t:
                per_launch_in = tm.lz4_in_bytes / max(tm.lz4_launches, 1)
                res["roofline"]["traffic"] = int(per_launch_in * (t["fetch_bytes_per_input_byte"] + t["write_bytes_per_input_byte"]))
                # (a quoted ratio x this run's bytes, NOT a counter read in this run: PMC passes serialise kernels and cannot ride in a timed bench)
                res["roofline"]["traffic_source"] = f"QUOTED from profiles/traffic.json ({t.get('measured', 'round 5, GPU calls r5p-r5r')}), scaled by this run's input bytes per launch: " + t["source"]
                # The STAGE reads the stream more than once: whole-chunk MD5 is a second pass over the same bytes -- it cannot ride the
                # compressor's LDS copy: a chain consumes its chunk at ~100 MB/s, a workgroup eats a chunk at ~2.3 GB/s --, FETCH 1.0 B per input byte with
                # no re-reads (profiles/r3_pmc_traffic_kernels.txt); with --cdc the candidates kernel reads it again with a 64-byte warm-up per 512 (x 1.125) and
                # the segment digests once more.  SURVEY 8d's algorithmic figure (N + C) counts one read.
                tm5 = json.loads(tf.read_text()).get("sky_md5_chunks:any")
                _md5_part = (f"sky_md5_chunks ({tm5['measured']})", tm5["fetch_bytes_per_input_byte"]) if tm5 else ("sky_md5_chunks (one read, measured r3)", 1.0)
                extra = _md5_part[1] + ((1.125 + 1.0) if args.cdc else 0.0)
                res["roofline"]["stage_traffic"] = {
                    "bytes_per_input_byte": round(t["fetch_bytes_per_input_byte"] + t["write_bytes_per_input_byte"] + extra, 4),
                    "algorithmic_bytes_per_input_byte": round(1.0 + tm.lz4_out_bytes / max(tm.lz4_in_bytes, 1), 4),
                    "parts": {"compressor (measured)": round(t["fetch_bytes_per_input_byte"] + t["write_bytes_per_input_byte"], 4), _md5_part[0]: _md5_part[1],
                              **({"sky_gear_candidates (512 + 64 bytes per lane)": 1.125, "sky_segment_md5x (one read of every segment byte)": 1.0} if args.cdc else {})},
                    "note": "the digest(s) and the CDC kernels are separate passes over the resident stream by construction: a serial chain per chunk / per segment "
                            "cannot share the compressor's one-block-at-a-time LDS copy"}
        # The STEP's own fraction, next to the dominant kernel's: algorithmic bytes of a step (N + C) over the step's wall time.  With --cdc the step is
        # compressor + candidates + segment digests + whole-chunk digests sharing the chip (their times overlap and SUM in chip time, profiles/r5_cdc.txt):
        # the compressor's launch-level `frac` alone overstates what the stage reaches.
        step_s = elapsed / args.steps
        step_alg = (n_chunks * cb + comp_bytes) * (world if world > 1 else 1)
        res["roofline"]["step"] = {"algorithmic_bytes": int(step_alg), "ms": round(step_s * 1e3, 3), "achieved": round(step_alg / step_s / 1e9, 2),
                                   "frac": round(step_alg / step_s / 1e9 / (HBM_PEAK_GBPS * world), 5),
                                   "kernels_ms_per_step": {kname: round(tm.lz4_ms / args.steps, 3), "sky_md5_chunks": round(tm.md5_ms / args.steps, 3),
                                                           **({"sky_frame_layout": round(tm.layout_ms / args.steps, 3), "sky_frame_gather": round(tm.gather_ms / args.steps, 3)} if not in_place else {}),
                                                           **({"sky_gear_candidates + sky_gear_select + sky_segment_md5x + sky_dedup_*": round(tm.cdc_ms / args.steps, 3)} if args.cdc else {})},
                                   "note": "kernel times are HIP-event spans on their own streams and overlap each other (and, with two steps in flight, the neighbouring step): they do not add up to ms"}
        if args.cdc:
            res["kernels_ms_per_step"]["cdc"] = round(tm.cdc_ms / args.steps, 3)
            prefix, cuts, fps, first, base = ctxs[last_lane].cdc_results(n_chunks, in_len)      # (the context that ran the last step)
            seg_end = cuts.astype(np.int64)
            seg_start = np.concatenate([[0], seg_end[:-1]])
            seg_start[prefix[:-1][prefix[:-1] < prefix[1:]].astype(np.int64)] = 0      # first segment of every chunk starts at 0
            seg_len = seg_end - seg_start
            dup = first != np.arange(base, base + len(first), dtype=np.uint64)
            res["config"]["segments"] = int(len(first)); res["config"]["avg_segment_bytes"] = round(float(seg_len.mean()), 1)
            res["config"]["duplicate_bytes_fraction"] = round(float(seg_len[dup].sum() / seg_len.sum()), 4)
            # the fraction the frozen CPU specification (oracle/skyoracle.c) finds on the same generator, next to the GPU's.  The
            # specification is run on the first 64 MiB of tile 0 (the stream is that unit tiled; duplicates live inside a tile).
            from oracle import ref

            n_s = min(n_chunks, max(1, (64 << 20) // cb))
            chunks_s = [np.concatenate(chunk_view(i, rots, True)) for i in range(n_s)]
            fps, lens = [], []
            for c in chunks_s:
                st = 0
                for e in ref.gear_cdc(c):
                    fps.append(hashlib.md5(c[st:int(e)]).digest()); lens.append(int(e) - st); st = int(e)
            first = ref.dedup_first(np.frombuffer(b"".join(fps), np.uint8).reshape(-1, 16))
            lens = np.array(lens)
            res["config"]["duplicate_bytes_fraction_cpu_spec"] = round(float(lens[first != np.arange(len(first))].sum() / lens.sum()), 4)
            res["config"]["cpu_spec_sample"] = f"{n_s} chunks ({n_s * cb >> 20} MiB) of tile 0: oracle gear_cdc + md5 + first-seen"
            g0 = int(prefix[n_s])
            res["config"]["duplicate_bytes_fraction_same_sample_gpu"] = round(float(seg_len[:g0][dup[:g0]].sum() / seg_len[:g0].sum()), 4)
        if want_cpu:
            log(f"cpu_baseline on {cores} processes")
            res["cpu_baseline"] = cpu_baseline(pool, cores, quota, rots, 12.0 if world == 1 else 6.0)
    if pool is not None:
        pool.close(); pool.join()
    for c_ in ctxs + md5_ctxs:
        c_.close()
    if rank == 0:
        if secondary_res is not None:
            res["secondary"] = secondary_res
        log("done")
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
