#!/usr/bin/env python3
"""Steady-state variant of scripts/e2e_loopback.py: same topology

    chunk files -> gpu_compress operator -> sender threads -> K loopback TCP connections -> deferred receiver
                -> gpu_decompress operator (decode + on-device MD5 check) -> chunk files

but the clock starts only after one warm-up chunk per connection has travelled the whole path, i.e. after every forked
worker has created its device context, grown its pinned arenas and every TCP connection is up -- what a transfer of
minutes amortises anyway.  (e2e_loopback.py times a cold start and is the one the GPU test suite runs.)

--context emu runs both operators on the shipping kernel source under the CPU emulator (tests/emu): that is how this
script is tested without a GPU (tests/test_host_operator.py); use small --chunk-kib with it.
"""
import argparse
import dataclasses
import hashlib
import json
import os
import queue as pyqueue
import socket
import sys
import tempfile
import threading
import time
import uuid
from multiprocessing import Event, Process, Queue
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
from e2e_loopback import receiver_main  # noqa: E402  (the deferred receiver process)
from skyplane_amd import synth  # noqa: E402
from skyplane_amd.chunk import Chunk, ChunkRequest  # noqa: E402
from skyplane_amd.gateway.chunk_store import ChunkStore  # noqa: E402
from skyplane_amd.gateway.gateway_queue import GatewayQueue  # noqa: E402
from skyplane_amd.gateway.operators import hip_sender  # noqa: E402
from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress, GatewayHipDecompress  # noqa: E402


def emu_context_factory(device_id, max_chunk_bytes, max_batch):
    """Device stand-in for machines without a GPU: the shipping kernels' source run by the CPU SIMT emulator."""
    from skyplane_amd.hip_ops import ChunkResult
    from tests.emu import emulib

    class EmuContext:
        def process_batch(self, chunks, flags=3):
            frames, md5s, _ = emulib.process([bytes(c) for c in chunks], flags=flags)
            return [ChunkResult(frame=f, md5=m if flags & 2 else None) for f, m in zip(frames, md5s)]

        def decompress_batch(self, frames, raw_lens, want_md5=False, into=None):
            rc, outs, status = emulib.decompress([bytes(f) for f in frames], [int(r) for r in raw_lens])
            if rc != 0:
                raise ValueError(f"frame rejected: {status}")
            return (outs, emulib.process(outs, flags=2)[1]) if want_md5 else outs

        def close(self):
            pass

    class EmuDedupContext(EmuContext):
        """+ Gear CDC, fingerprints and the dedup table (--dedup-wire): the call shape of skyhip_cdc_results / skyhip_dedup_reset."""

        def __init__(self):
            from oracle import ref

            self.cdc, self.gear, self.last = emulib.EmuCdc(16), ref.gear_table(), None

        def process_batch(self, chunks, flags=3, frames_into=None):
            raw = [bytes(c) for c in chunks]
            frames, md5s, _ = emulib.process(raw, flags=flags & 3)
            if flags & 4:
                prefix, seg_end, fps, first, base, _ = self.cdc.run(raw, self.gear, dedup=bool(flags & 8))
                self.last = (prefix.astype("uint64"), seg_end, fps, first, base)
            return [ChunkResult(frame=f if flags & 1 else None, md5=m if flags & 2 else None) for f, m in zip(frames, md5s)]

        def cdc_results(self, n, in_len):
            return self.last

        def dedup_reset(self):
            base = self.cdc.seg_base
            self.cdc = emulib.EmuCdc(16)
            self.cdc.seg_base = base

    return EmuDedupContext() if os.environ.get("E2E_DEDUP_WIRE") else EmuContext()


def null_context_factory(device_id, max_chunk_bytes, max_batch):
    """Plumbing-only stand-in (--context null): "frames" are the raw bytes behind a 16-byte tag, digests are zeros, one memcpy each way.
    What is left is the cost of everything around the device: files, queues, sockets, Python.  Not a correctness run."""
    import numpy as np
    from skyplane_amd.hip_ops import ChunkResult

    class NullContext:
        def pinned_buffer(self, n):
            return np.empty(n, np.uint8)

        def release_pinned(self, a):
            pass

        def frame_bound(self, n):
            return n + 16

        def process_batch(self, chunks, flags=3, frames_into=None):
            out = []
            for i, c in enumerate(chunks):
                v = frames_into[i][: len(c) + 16]
                v[:16] = 0x5A
                v[16:] = c
                out.append(ChunkResult(frame=v, md5=bytes(16) if flags & 2 else None))
            return out

        def decompress_batch(self, frames, raw_lens, want_md5=False, into=None):
            outs = []
            for f, r, o in zip(frames, raw_lens, into):
                o[:r] = f[16:16 + r]
                outs.append(o[:r])
            return (outs, [bytes(16)] * len(outs)) if want_md5 else outs

        def close(self):
            pass

    return NullContext()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=256)
    ap.add_argument("--chunk-kib", type=int, default=8192)
    ap.add_argument("--connections", type=int, default=8)
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--no-prealloc", action="store_true", help="let the operators grow their pinned arenas inside the timed region (round-1 behaviour)")
    ap.add_argument("--context", choices=["hip", "emu", "null"], default="hip")
    ap.add_argument("--handoff", choices=["arena", "files"], default="arena", help="how payloads travel operator -> sender and receiver -> operator: slots of a "
                                                                                   "shared page-locked arena (gateway/shm_arena.py) or one tmpfs file per chunk")
    ap.add_argument("--dedup-wire", action="store_true", help="dedup on the wire (gateway/dedup_wire.py): a 50 %%-duplicate stream, recipes instead of frames, "
                                                            "one destination worker process (its lanes share the segment store)")
    ap.add_argument("--dedup-store", choices=["memory", "files"], default="memory", help="--dedup-wire: where the destination keeps literal segments -- in the "
                    "one worker process (memory) or in files of the chunk directory shared by --workers processes")
    ap.add_argument("--dst-consume", action="store_true", help="stand in for write_object_store + the daemon's clean-up: every decoded chunk is checked (size; MD5 of "
                    "one in eight -- the destination operator has compared every chunk's device-side digest with the sender's already) and DELETED as it "
                    "arrives, which is what frees the destination's page-locked slot files (shm_arena.LinkSlots) for the next chunks")
    ap.add_argument("--out-slots", type=int, default=-1, help="slot files per destination lane (-1 = the operator's default, 0 = plain writes)")
    ap.add_argument("--dedup-epoch-mb", type=int, default=2048, help="--dedup-wire: the source starts a new table epoch after this many MB (the planner patch's default, INTEGRATION 10)")
    ap.add_argument("--dst-fill-wait-ms", type=float, default=-1, help="destination lanes collect for up to this long before a device call (-1 = the operator's default)")
    ap.add_argument("--dst-depth", type=int, default=0, help="pipeline lanes per destination worker (0 = the operator's default)")
    ap.add_argument("--src-depth", type=int, default=0, help="pipeline lanes per source worker (0 = the operator's default)")
    ap.add_argument("--src-reader", choices=["none", "files", "slots"], default="none", help="stand in for read_object_store INSIDE the timed region (none = the chunk "
                    "files exist before the clock starts, as in rounds 2-5): four reader threads write every chunk's bytes -- 'files': open(path, \"wb\") as "
                    "download_object does today; 'slots': INTEGRATION 6e -- the name is first linked to one of gpu_compress's page-locked source slots "
                    "(shm_arena.claim_slot) and the bytes are written into the existing file -- and the sender's side deletes <id>.chunk when it is sent "
                    "(the daemon's clean-up, gateway_daemon_api.py:125-127), which frees the slot")
    ap.add_argument("--in-slots", type=int, default=256, help="--src-reader slots: source slot files per source worker")
    ap.add_argument("--dedup-verify", choices=["segments", "chunk"], default="segments", help="--dedup-wire: how the destination checks a rebuilt chunk -- its newly "
                    "arrived literal segments against their fingerprints (one parallel device call per batch, the operator's default) or its own whole-chunk MD5 chain")
    a = ap.parse_args()
    if a.dedup_wire:
        os.environ["E2E_DEDUP_WIRE"] = "1"
    size = a.chunk_kib << 10
    factory = {"emu": emu_context_factory, "null": null_context_factory}.get(a.context)
    if a.context == "emu":
        from tests.emu import emulib
        emulib.lib()                                    # build before anything forks
    K = a.connections
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        src, dst = ChunkStore(Path(tmp) / "src"), Path(tmp) / "dst"
        q_in, q_out = GatewayQueue(), GatewayQueue()
        src.add_partition("0", q_in)
        dst_store = ChunkStore(dst)
        dq_in, dq_out = GatewayQueue(), GatewayQueue()
        dst_store.add_partition("0", dq_in)
        base = synth.mixed_chunks(4, size, config_id=4)
        # --dedup-wire: a 50 %-duplicate stream (half of its 8-64 KiB spans are copies of earlier spans at unaligned offsets).  Generated for at most 128 chunks
        # and tiled like bench.py --cdc tiles its unit (tile t = the base XOR t: the duplicate structure inside a tile stays, tiles are mutually distinct):
        # the generator is a Python loop over spans, and 8.6 GiB of it was minutes of set-up around a 5 s measurement
        DD_TILE = 128
        dd_stream = synth.dedup_stream(min(a.connections + a.chunks, DD_TILE) * size, dup_fraction=0.5, config_id=3) if a.dedup_wire else None
        digests = {}

        # Without --dedup-wire the stream is four chunk contents in turn: they are written and hashed ONCE and every chunk file is a hard link (the operators only
        # read the source's files, and the daemon API's unlink of `<id>.chunk` drops a name, not the data) -- set-up used to write and hash every one of the
        # thousand-odd files (60-80 s around a 2 s measurement, NOTES.md), which is what kept the runs too short for their pipeline fill and drain not to show.
        proto = {}

        reader_jobs = {}                                  # --src-reader: chunk id -> the bytes its reader thread will write

        def make(i, late=False):
            cid = uuid.uuid4().hex
            if late and dd_stream is None:
                k = i % 4
                if k not in proto:
                    data = base[k].tobytes()
                    proto[k] = (data, hashlib.md5(data).digest())
                reader_jobs[cid] = proto[k][0]
                digests[cid] = proto[k][1]
                return ChunkRequest(chunk=Chunk(src_key=f"/s/{i}", dest_key=str(i), chunk_id=cid, chunk_length_bytes=size, partition_id="0"))
            if dd_stream is None:
                k = i % 4
                if k not in proto:
                    data = base[k].tobytes()
                    pp = Path(tmp) / f"proto{k}.bin"
                    pp.write_bytes(data)
                    proto[k] = (pp, hashlib.md5(data).digest())
                os.link(proto[k][0], src.get_chunk_file_path(cid))
                digests[cid] = proto[k][1]
            else:
                k, t = i % DD_TILE, i // DD_TILE
                data = (dd_stream[k * size:(k + 1) * size] ^ np.uint8(t & 0xFF)).tobytes() if t else dd_stream[k * size:(k + 1) * size].tobytes()
                src.get_chunk_file_path(cid).write_bytes(data)
                digests[cid] = hashlib.md5(data).digest()
            return ChunkRequest(chunk=Chunk(src_key=f"/s/{i}", dest_key=str(i), chunk_id=cid, chunk_length_bytes=size, partition_id="0"))

        late = a.src_reader != "none"
        assert not (late and a.dedup_wire), "--src-reader is for the plain-frame stream"
        warm = [make(i, late) for i in range(K)]              # one per connection
        main_reqs = [make(K + i, late) for i in range(a.chunks)]
        shares = [[warm[k]] + main_reqs[k::K] for k in range(K)]
        port_q, done_q = Queue(), Queue()
        rx = Process(target=receiver_main, args=(dst, port_q, done_q, K, (max(2, a.dst_depth or 3) * a.max_batch * max(a.workers, 1) + K) if a.handoff == "arena" else 0, size))
        rx.start()
        port = port_q.get(timeout=120)
        err_ev, err_q = Event(), Queue()
        kw = {"context_factory": factory} if factory else {}
        kw["prealloc"] = not a.no_prealloc
        op = GatewayHipCompress("gpu_compress_0", "local:e2e", q_in, q_out, err_ev, err_q, src, n_processes=a.workers, max_batch=a.max_batch,
                                max_chunk_bytes=size, device_ids=[0], dedup_wire=a.dedup_wire, handoff=a.handoff, pipeline_depth=a.src_depth or None,
                                dedup_epoch_bytes=a.dedup_epoch_mb << 20, in_slots=a.in_slots if a.src_reader == "slots" else 0,
                                in_slot_chunk_bytes=size if a.src_reader == "slots" else 0, **kw)
        dop = GatewayHipDecompress("gpu_decompress_0", "local:e2e-dst", dq_in, dq_out, err_ev, err_q, dst_store, n_processes=1 if (a.dedup_wire and a.dedup_store == "memory") else a.workers,
                                   max_batch=a.max_batch, max_chunk_bytes=size, device_ids=[0], dedup_store=a.dedup_store, pipeline_depth=a.dst_depth or None,
                                   dedup_wire=a.dedup_wire,      # (what the planner patch passes: INTEGRATION 10 -- three lanes by default then)
                                   dedup_verify=a.dedup_verify,
                                   out_slots=None if a.out_slots < 0 else a.out_slots, fill_wait_s=None if a.dst_fill_wait_ms < 0 else a.dst_fill_wait_ms / 1e3, **kw)
        total = K + a.chunks
        ready, ready_cv = {}, threading.Condition()
        go = threading.Event()
        stop_drain = threading.Event()
        status_records = []
        wire = [0] * K
        trace = {"compressed": [], "sent": [], "decoded": []}

        def collect():
            got = 0
            while got < total and not err_ev.is_set():
                try:
                    cr = q_out.q.get(timeout=0.2)
                except pyqueue.Empty:
                    continue
                with ready_cv:
                    ready[cr.chunk.chunk_id] = cr
                    ready_cv.notify_all()
                got += 1
                trace["compressed"].append(time.perf_counter())

        def drain_status():
            while not stop_drain.is_set() or not src.chunk_status_queue.empty() or not dst_store.chunk_status_queue.empty():
                for sq in (src.chunk_status_queue, dst_store.chunk_status_queue):
                    try:
                        status_records.append(sq.get(timeout=0.05))
                    except pyqueue.Empty:
                        pass

        def send(k):
            with socket.create_connection(("127.0.0.1", port)) as sock:
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                for idx, cr in enumerate(shares[k]):
                    if idx == 1:
                        go.wait(600)                    # the warm-up chunk is through: wait for the clock
                    with ready_cv:
                        ready_cv.wait_for(lambda: cr.chunk.chunk_id in ready or err_ev.is_set(), timeout=600)
                    if err_ev.is_set():
                        return
                    dig = hip_sender.chunk_digest(src, cr.chunk.chunk_id)
                    dst_store.add_chunk_request(ChunkRequest(chunk=dataclasses.replace(cr.chunk, md5_hash=dig.hex() if dig else None)))
                    n_sent = hip_sender.send_chunk(sock, src, cr, n_chunks_left_on_socket=len(shares[k]) - idx - 1, release=True)
                    if late:                            # the daemon's unlink of a completed chunk: what hands a source slot back
                        try:
                            os.unlink(src.get_chunk_file_path(cr.chunk.chunk_id))
                        except FileNotFoundError:
                            pass
                    if idx:
                        wire[k] += n_sent
                    trace["sent"].append(time.perf_counter())
                hip_sender.drain_releases(sock)         # slots go back when the peer has acknowledged their bytes, not when sendfile returns

        consumed = {"n": 0, "hashed": 0, "bad": []}
        from concurrent.futures import ThreadPoolExecutor
        consumers = ThreadPoolExecutor(4) if a.dst_consume else None      # (stat / read / hashlib / unlink all release the GIL)

        def consume(cr, do_hash):                      # write_object_store's view of the chunk, then the daemon's unlink
            f = dst / f"{cr.chunk.chunk_id}.chunk"
            try:
                ok = f.stat().st_size == cr.chunk.chunk_length_bytes and (not do_hash or hashlib.md5(f.read_bytes()).digest() == digests[cr.chunk.chunk_id])
                if not ok:
                    consumed["bad"].append(f.name)
                f.unlink()
            except BaseException as e:      # noqa: BLE001
                consumed["bad"].append(f"{f.name}: {e!r}")

        def wait_decoded(n):
            got = 0
            while got < n and not err_ev.is_set():
                try:
                    cr = dq_out.q.get(timeout=0.2)
                    got += 1
                    trace["decoded"].append(time.perf_counter())
                except pyqueue.Empty:
                    continue
                if consumers is not None:
                    do_hash = a.context != "null" and consumed["n"] % 8 == 0
                    consumed["hashed"] += int(do_hash)
                    consumed["n"] += 1
                    consumers.submit(consume, cr, do_hash)
            return got

        slot_claims = {"slots": 0, "files": 0}

        def read_and_queue(cr):                        # GatewayObjStoreReadOperator.process + download_object for one chunk, then on to gpu_compress
            from skyplane_amd.gateway import shm_arena
            path = src.get_chunk_file_path(cr.chunk.chunk_id)
            data = reader_jobs[cr.chunk.chunk_id]
            claimed = a.src_reader == "slots" and shm_arena.claim_slot(path, len(data))
            slot_claims["slots" if claimed else "files"] += 1
            with open(path, "r+b" if claimed else "wb") as f:
                f.write(data)
            src.add_chunk_request(cr)

        readers = ThreadPoolExecutor(4) if late else None

        op.start_workers()
        dop.start_workers()
        if a.src_reader == "slots":                    # the source's slots are made when its lanes start: the first download finds them
            t_w = time.time()
            while time.time() - t_w < 120 and not list(Path(src.get_chunk_file_path("x")).parent.glob("_inslot_*.bin")):
                time.sleep(0.05)
        drainer = threading.Thread(target=drain_status)
        drainer.start()
        threads = [threading.Thread(target=collect)] + [threading.Thread(target=send, args=(k,)) for k in range(K)]
        for t in threads:
            t.start()
        t_cold = time.perf_counter()
        for cr in warm:
            if late:
                readers.submit(read_and_queue, cr)
            else:
                src.add_chunk_request(cr)
        assert wait_decoded(K) == K or err_ev.is_set()
        warm_s = time.perf_counter() - t_cold
        t0 = time.perf_counter()
        go.set()
        for cr in main_reqs:
            if late:
                readers.submit(read_and_queue, cr)
            else:
                src.add_chunk_request(cr)
        n_dec = wait_decoded(a.chunks)
        if consumers is not None:
            consumers.shutdown(wait=True)              # the clock stops when the last chunk has been checked and deleted
            assert not consumed["bad"], consumed["bad"][:4]
        elapsed = time.perf_counter() - t0
        for t in threads:
            t.join(60)
        if err_ev.is_set():               # an operator failed: say why before anything else times out (r5g: 4096 chunks = 32 GiB of destination files did not fit the box's tmpfs)
            print("operator error:", err_q.get(timeout=5) if not err_q.empty() else "(no message)", file=sys.stderr, flush=True)
            os._exit(3)
        n_rx = 0
        for _ in range(K):
            n_rx += len(done_q.get(timeout=120))
        stop_drain.set()
        op.stop_workers()
        dop.stop_workers()
        drainer.join()
        rx.join(60)
        assert not err_ev.is_set(), err_q.get() if not err_q.empty() else "operator error"
        assert n_dec == a.chunks and n_rx == total
        if a.context != "null" and not a.dst_consume:           # every chunk that arrived, hashed on all the cores the container may use
            from concurrent.futures import ThreadPoolExecutor

            def check(cr):
                got = (dst / f"{cr.chunk.chunk_id}.chunk").read_bytes()
                return hashlib.md5(got).digest() == digests[cr.chunk.chunk_id] == hip_sender.chunk_digest(src, cr.chunk.chunk_id)

            with ThreadPoolExecutor(max(2, min(16, os.cpu_count() or 2))) as ex:      # (hashlib releases the GIL)
                assert all(ex.map(check, warm + main_reqs))
        raw = a.chunks * size
        # the rate between the first and the last quarter of the chunks leaving the destination operator: what a transfer of minutes sees, without this
        # run's pipeline fill (the first device call of each side: ~0.1 s each, one MD5 chain) and drain
        dts = sorted(t - t0 for t in trace["decoded"] if t >= t0)
        steady = None
        if len(dts) >= 64:
            q1, q3 = dts[len(dts) // 4], dts[(3 * len(dts)) // 4]
            if q3 > q1:
                steady = round(((3 * len(dts)) // 4 - len(dts) // 4) * size * 8 / (q3 - q1) / 1e9, 3)
        if os.environ.get("E2E_TRACE"):
            for name, ts in trace.items():
                ts = sorted(t - t0 for t in ts if t >= t0)
                if ts:
                    print(f"trace {name:10s} n={len(ts)} first={ts[0]:.3f}s median={ts[len(ts) // 2]:.3f}s last={ts[-1]:.3f}s", file=sys.stderr)
        print(json.dumps({"e2e": "loopback, steady state", "context": a.context, "chunks": a.chunks, "chunk_bytes": size, "connections": K, "workers": a.workers,
                          "max_batch": a.max_batch, "src_depth": a.src_depth or None, "dst_depth": a.dst_depth or None, "prealloc": not a.no_prealloc, "handoff": a.handoff, "dedup_wire": a.dedup_wire, "dedup_store": a.dedup_store if a.dedup_wire else None, "dedup_verify": a.dedup_verify if a.dedup_wire else None, "src_reader": a.src_reader, "src_reader_writes": slot_claims if late else None, "effective_gbit_s": round(raw * 8 / elapsed / 1e9, 3), "middle_half_gbit_s": steady, "seconds": round(elapsed, 3),
                          "warmup_seconds": round(warm_s, 3), "raw_GiB": round(raw / 2**30, 3), "wire_ratio": round(raw / max(sum(wire), 1), 3),
                          "status_records": len(status_records), "verified": a.context != "null",
                          "dst_consume": ({"chunks_deleted_on_arrival": consumed["n"], "hashed_on_the_cpu": consumed["hashed"]} if a.dst_consume else None),
                          "out_slots": None if a.out_slots < 0 else a.out_slots}))


if __name__ == "__main__":
    main()
