#!/usr/bin/env python3
"""Config-5 shape at reduced scale, with this repo's stand-alone mirrors of the gateway pieces (the reference daemon
itself cannot be imported offline):

    chunk files -> gpu_compress operator (forked worker, GPU) -> sender threads (sidecar frames, 53-byte headers)
                -> loopback TCP, K connections -> receiver process (deferred decode: payloads left as sidecars)
                -> gpu_decompress operator (forked worker, GPU: batched decode + MD5 check on the device) -> chunk files

Reports effective Gbit/s = raw bytes x 8 / wall time and verifies every destination file against its source.
Plumbing (Python, tmpfs files, pickled queues, per-batch MD5 latency) bounds this number, not the kernels.
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
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from skyplane_amd import synth  # noqa: E402
from skyplane_amd.chunk import Chunk, ChunkRequest  # noqa: E402
from skyplane_amd.gateway.chunk_store import ChunkStore  # noqa: E402
from skyplane_amd.gateway.gateway_queue import GatewayQueue  # noqa: E402
from skyplane_amd.gateway.operators import hip_receiver, hip_sender  # noqa: E402
from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress, GatewayHipDecompress  # noqa: E402


def receiver_main(dst_dir, port_q, done_q, n_conn, arena_slots=0, max_chunk_bytes=0):
    """Destination gateway's receiver stand-in: one process, one thread per connection, no GPU -- the decode is the
    gpu_decompress operator's job (hip_receiver.recv_chunks(decompress=None))."""
    store = ChunkStore(dst_dir)
    arena = hip_receiver.make_arena(store, "e2e", max_chunk_bytes, arena_slots) if arena_slots else None     # shared-memory hand-off to gpu_decompress
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", 0))
    srv.listen(n_conn)
    port_q.put(srv.getsockname()[1])

    def serve(conn):
        with conn:
            done_q.put(hip_receiver.recv_chunks(conn, store, None, arena=arena))

    threads = []
    for _ in range(n_conn):
        conn, _ = srv.accept()
        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        t = threading.Thread(target=serve, args=(conn,))
        t.start()
        threads.append(t)
    for t in threads:
        t.join()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=64)
    ap.add_argument("--chunk-mib", type=int, default=8)
    ap.add_argument("--connections", type=int, default=4)
    ap.add_argument("--max-batch", type=int, default=32)
    ap.add_argument("--workers", type=int, default=1, help="forked operator workers per side (each with its own HIP context on GPU 0)")
    args = ap.parse_args()
    size = args.chunk_mib << 20
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        src, dst = ChunkStore(Path(tmp) / "src"), Path(tmp) / "dst"
        q_in, q_out = GatewayQueue(), GatewayQueue()
        src.add_partition("0", q_in)
        dst_store = ChunkStore(dst)
        dq_in, dq_out = GatewayQueue(), GatewayQueue()
        dst_store.add_partition("0", dq_in)
        base = synth.mixed_chunks(4, size, config_id=4)
        reqs, digests = [], {}
        for i in range(args.chunks):
            cid = uuid.uuid4().hex
            data = base[i % 4].tobytes()
            src.get_chunk_file_path(cid).write_bytes(data)
            digests[cid] = hashlib.md5(data).digest()
            reqs.append(ChunkRequest(chunk=Chunk(src_key=f"/s/{i}", dest_key=str(i), chunk_id=cid, chunk_length_bytes=size, partition_id="0")))
        port_q, done_q = Queue(), Queue()
        rx = Process(target=receiver_main, args=(dst, port_q, done_q, args.connections))
        rx.start()
        port = port_q.get(timeout=120)
        err_ev, err_q = Event(), Queue()
        op = GatewayHipCompress("gpu_compress_0", "local:e2e", q_in, q_out, err_ev, err_q, src, n_processes=args.workers, max_batch=args.max_batch, max_chunk_bytes=size, device_ids=[0])
        dop = GatewayHipDecompress("gpu_decompress_0", "local:e2e-dst", dq_in, dq_out, err_ev, err_q, dst_store, n_processes=args.workers, max_batch=args.max_batch,
                                   max_chunk_bytes=size, device_ids=[0])
        # static split of the chunk set over the connections, like the reference's per-connection chunk lists
        shares = [reqs[k::args.connections] for k in range(args.connections)]
        ready = {}            # chunk_id -> ChunkRequest once the operator has produced its sidecar
        ready_cv = threading.Condition()

        def collect():
            got = 0
            while got < len(reqs) and not err_ev.is_set():
                try:
                    cr = q_out.q.get(timeout=0.2)
                except pyqueue.Empty:
                    continue
                with ready_cv:
                    ready[cr.chunk.chunk_id] = cr
                    ready_cv.notify_all()
                got += 1

        wire = [0] * args.connections
        status_records = []
        stop_drain = threading.Event()

        def drain_status():
            # the daemon's main loop does this (gateway_daemon.py:329-330); an undrained multiprocessing queue would
            # keep the operator's worker process from exiting
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
                    with ready_cv:
                        ready_cv.wait_for(lambda: cr.chunk.chunk_id in ready or err_ev.is_set(), timeout=300)
                    # what the reference's pre-registration POST does (gateway_operator.py:279-316), with the digest the
                    # source operator computed riding along as a JSON-safe hex string
                    dig = hip_sender.chunk_digest(src, cr.chunk.chunk_id)
                    dst_store.add_chunk_request(ChunkRequest(chunk=dataclasses.replace(cr.chunk, md5_hash=dig.hex() if dig else None)))
                    n_sent = hip_sender.send_chunk(sock, src, cr, n_chunks_left_on_socket=len(shares[k]) - idx - 1)
                    wire[k] += n_sent

        t0 = time.perf_counter()
        for cr in reqs:
            src.add_chunk_request(cr)
        op.start_workers()
        dop.start_workers()
        drainer = threading.Thread(target=drain_status)
        drainer.start()
        threads = [threading.Thread(target=collect)] + [threading.Thread(target=send, args=(k,)) for k in range(args.connections) if shares[k]]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        n_rx = 0
        for _ in range(sum(1 for s in shares if s)):
            n_rx += len(done_q.get(timeout=300))
        n_dec = 0
        while n_dec < len(reqs) and not err_ev.is_set():                     # the destination operator has written every chunk file
            try:
                dq_out.q.get(timeout=0.2)
                n_dec += 1
            except pyqueue.Empty:
                pass
        elapsed = time.perf_counter() - t0
        stop_drain.set()
        op.stop_workers()
        dop.stop_workers()
        drainer.join()
        rx.join(60)
        assert not err_ev.is_set(), err_q.get() if not err_q.empty() else "operator error"
        assert n_rx == len(reqs)
        for cr in reqs:
            got = (dst / f"{cr.chunk.chunk_id}.chunk").read_bytes()
            assert hashlib.md5(got).digest() == digests[cr.chunk.chunk_id] == hip_sender.chunk_digest(src, cr.chunk.chunk_id)
        raw = len(reqs) * size
        print(json.dumps({"e2e": "loopback", "chunks": len(reqs), "chunk_mib": args.chunk_mib, "connections": args.connections, "workers": args.workers,
                          "effective_gbit_s": round(raw * 8 / elapsed / 1e9, 2), "raw_GiB": round(raw / 2**30, 2), "wire_ratio": round(raw / sum(wire), 3),
                          "seconds": round(elapsed, 2), "status_records": len(status_records), "verified": True}))


if __name__ == "__main__":
    main()
