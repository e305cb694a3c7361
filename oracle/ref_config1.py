#!/usr/bin/env python3
"""BASELINE configs[0] ("plumbing"): the REAL reference gateway data path timed on this container's CPU.

    chunk files -> reference GatewaySender (worker_loop + process: read, lz4.frame.compress, header, sendall)
                -> loopback TCP -> reference GatewayReceiver (recv, lz4.frame.decompress, write, size check)

Only the sender's HTTP control calls are answered by a stub (there is no gateway_daemon_api here); lz4.frame is the
system liblz4 behind python-lz4's defaults (oracle/refshim.py).  TEST/MEASUREMENT INFRASTRUCTURE: it needs
/root/reference, so it runs in the build container only; the result is committed as profiles/r1_reference_config1.json
and quoted in DESIGN.md.  Reports (i) the rate of the unmodified worker_loop (0.1 s sleep per chunk,
gateway_operator.py:102) and (ii) the codec+socket rate of process() alone, per worker.

usage: python oracle/ref_config1.py [--chunks 48] [--workers 1,4] [--data silesia|random] [--out profiles/...json]
"""
import argparse
import json
import os
import shutil
import socket
import sys
import tempfile
import threading
import time
import uuid
from multiprocessing import Event, Queue
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import refshim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=48)
ap.add_argument("--workers", default="1,4")
ap.add_argument("--data", default="silesia", choices=["silesia", "random"])
ap.add_argument("--out", default="")
args = ap.parse_args()

base = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path(tempfile.gettempdir())
scratch = Path(tempfile.mkdtemp(prefix="sky_cfg1_", dir=base))
refshim.install(scratch / "shim")

import numpy as np  # noqa: E402

from skyplane_amd import synth  # noqa: E402

import skyplane.chunk as ref_chunk  # noqa: E402
from skyplane.gateway.chunk_store import ChunkStore  # noqa: E402
from skyplane.gateway.gateway_queue import GatewayQueue  # noqa: E402
from skyplane.gateway.operators.gateway_operator import GatewaySender  # noqa: E402
from skyplane.gateway.operators.gateway_receiver import GatewayReceiver  # noqa: E402

CB = synth.CHUNK_BYTES


class _Reply:
    status = 200

    def __init__(self, body):
        self.data = json.dumps(body).encode()


class ControlPlaneStub:
    """POST /servers -> start a receiver server (gateway_daemon_api.py does the same); POST /chunk_requests -> n_added;
    GET /incomplete_chunk_requests -> none left; DELETE /servers/<port> -> ok."""

    def __init__(self, receiver):
        self.receiver = receiver

    def request(self, method, url, body=None, headers=None):
        if method == "POST" and url.endswith("/api/v1/servers"):
            return _Reply({"server_port": self.receiver.start_server()})
        if method == "POST" and url.endswith("/api/v1/chunk_requests"):
            return _Reply({"status": "ok", "n_added": len(json.loads(body))})
        if method == "GET" and url.endswith("/incomplete_chunk_requests"):
            return _Reply({"chunk_requests": {}})
        if method == "DELETE":
            return _Reply({"status": "ok"})
        raise AssertionError((method, url))


def make_data(n):
    if args.data == "random":
        rng = np.random.Generator(np.random.PCG64(0x5EED0001))
        return [rng.integers(0, 256, CB, dtype=np.uint8).tobytes() for _ in range(n)]
    unit = synth.silesia_like(min(n, 16) * CB, config_id=2)
    return [unit[(i % 16) * CB:((i % 16) + 1) * CB].tobytes() for i in range(n)]


def drain(q, stop):
    while not stop.is_set():
        try:
            q.get(timeout=0.05)
        except Exception:  # noqa: BLE001
            pass


def run(n_workers, datas):
    src_dir, dst_dir = scratch / f"src{n_workers}", scratch / f"dst{n_workers}"
    src, dst = ChunkStore(str(src_dir)), ChunkStore(str(dst_dir))
    error_event, error_queue = Event(), Queue()
    receiver = GatewayReceiver("recv", "local:dst", dst, error_event, error_queue, use_tls=False, use_compression=True)
    in_q, out_q = GatewayQueue(), GatewayQueue()
    in_q.register_handle("send")
    sender = GatewaySender("send", "local:src", in_q, out_q, error_event, error_queue, src, ip_addr="127.0.0.1", use_tls=False,
                           use_compression=True, n_processes=n_workers)
    sender.http_pool = ControlPlaneStub(receiver)
    stop = threading.Event()
    for q in (src.chunk_status_queue, dst.chunk_status_queue, receiver.socket_profiler_event_queue):
        threading.Thread(target=drain, args=(q, stop), daemon=True).start()
    reqs = []
    for i, d in enumerate(datas):
        cid = uuid.uuid4().hex
        src.get_chunk_file_path(cid).write_bytes(d)
        reqs.append(ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk(src_key=f"k{i}", dest_key=f"k{i}", chunk_id=cid, chunk_length_bytes=len(d), partition_id="0")))
    sender.start_workers()
    t0 = time.time()
    for r in reqs:
        in_q.put(r)
    import queue as _queue
    done = 0
    while done < len(reqs):
        try:
            out_q.get_nowait()
            done += 1
        except _queue.Empty:
            if error_event.is_set():
                raise RuntimeError(error_queue.get())
            time.sleep(0.002)
    # the sender marks a chunk complete after sendall(); wait until the receiver has written the last file too
    deadline = time.time() + 60
    while time.time() < deadline:
        if all(dst.get_chunk_file_path(r.chunk.chunk_id).exists() and dst.get_chunk_file_path(r.chunk.chunk_id).stat().st_size == r.chunk.chunk_length_bytes for r in reqs):
            break
        time.sleep(0.01)
    wall = time.time() - t0
    ok = all(dst.get_chunk_file_path(r.chunk.chunk_id).read_bytes() == d for r, d in zip(reqs, datas))
    for i in range(n_workers):
        sender.exit_flags[i].set()
    for p in sender.processes:
        p.join(20)
        if p.is_alive():
            p.terminate()
    stop.set()
    shutil.rmtree(src_dir, ignore_errors=True)
    shutil.rmtree(dst_dir, ignore_errors=True)
    raw = sum(len(d) for d in datas)
    return {"workers": n_workers, "chunks": len(datas), "wall_s": round(wall, 3), "gbit_s": round(raw * 8 / wall / 1e9, 3), "gib_s": round(raw / wall / 2**30, 4), "verified": ok}


def process_only(datas):
    """process() without the loop's sleep: codec + socket cost per chunk on one worker."""
    src, dst = ChunkStore(str(scratch / "srcp")), ChunkStore(str(scratch / "dstp"))
    error_event, error_queue = Event(), Queue()
    receiver = GatewayReceiver("recv", "local:dst", dst, error_event, error_queue, use_tls=False, use_compression=True)
    sender = GatewaySender("send", "local:src", GatewayQueue(), GatewayQueue(), error_event, error_queue, src, ip_addr="127.0.0.1", use_tls=False,
                           use_compression=True, n_processes=1)
    sender.worker_id = 0
    sender.http_pool = ControlPlaneStub(receiver)
    stop = threading.Event()
    threading.Thread(target=drain, args=(receiver.socket_profiler_event_queue, stop), daemon=True).start()
    reqs = []
    for i, d in enumerate(datas):
        cid = uuid.uuid4().hex
        src.get_chunk_file_path(cid).write_bytes(d)
        reqs.append(ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk(src_key=f"k{i}", dest_key=f"k{i}", chunk_id=cid, chunk_length_bytes=len(d), partition_id="0")))
    t0 = time.time()
    for r in reqs:
        assert sender.process(r, "127.0.0.1")
    last = dst.get_chunk_file_path(reqs[-1].chunk.chunk_id)
    while not (last.exists() and last.stat().st_size == len(datas[-1])):
        time.sleep(0.005)
    wall = time.time() - t0
    for s in sender.destination_sockets.values():
        s.close()
    for p in receiver.server_processes:
        p.terminate()
    stop.set()
    raw = sum(len(d) for d in datas)
    return {"chunks": len(datas), "wall_s": round(wall, 3), "gbit_s": round(raw * 8 / wall / 1e9, 3), "gib_s": round(raw / wall / 2**30, 4)}


if __name__ == "__main__":
    import contextlib
    import io
    import signal

    try:
        os.setpgrp()  # own process group: the reference's receiver servers spin forever on a closed socket
                      # (WireProtocolHeader.from_socket has no EOF check) and are reaped as a group at the end
    except PermissionError:
        pass

    datas = make_data(args.chunks)
    res = {"what": "reference gateway sender->receiver over loopback, CPU only (BASELINE configs[0] shape)", "data": args.data, "chunk_bytes": CB,
           "host_cores": os.cpu_count(), "worker_loop": [], "note": "lz4.frame = system liblz4 via oracle/refshim.py; HTTP control plane stubbed; TLS/e2ee off"}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):        # the reference prints per chunk
        res["process_only_1_worker"] = process_only(datas[: min(len(datas), 24)])
        for w in [int(x) for x in args.workers.split(",")]:
            res["worker_loop"].append(run(w, datas))
    shutil.rmtree(scratch, ignore_errors=True)
    line = json.dumps(res)
    print(line)
    if args.out:
        Path(args.out).write_text(line + "\n")
    sys.stdout.flush()
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    os.killpg(os.getpgrp(), signal.SIGTERM)
