#!/usr/bin/env python3
"""BASELINE configs[0] with the reference's OWN daemons: two real `GatewayDaemon`s (source, destination) on localhost,

    POST /api/v1/chunk_requests -> read_object_store(local:) -> send(compress) == TCP ==> receive -> write_object_store(local:)

Flask control plane, forked operator workers and receiver servers all from /root/reference (import shim:
oracle/refshim.py).  What the harness adds: the gateway program / info JSON files, a TLS front for the destination's
API (the sender speaks https://<dst>:8080, production puts stunnel there; here: stdlib `ssl` + an `openssl`-made
self-signed certificate forwarding to the API's plain 8081), a different API port for the source daemon so that both
fit on one host, and the client role (dispatch chunk requests, poll for completion, verify the files).

With --gpu-op the source program is read_object_store -> gpu_compress -> send and the daemon/sender sources get
INTEGRATION.md sections 5 and 6 applied in memory; the operator's device context is the shipping kernel source under the
CPU emulator (there is no GPU in the build container), so keep the chunks small.

TEST / MEASUREMENT INFRASTRUCTURE, build container only.  One JSON line on stdout.
usage: python oracle/ref_daemon.py [--chunks 32] [--chunk-kib 8192] [--connections 4] [--gpu-op] [--out FILE]
"""
import argparse
import hashlib
import json
import os
import shutil
import signal
import socket
import ssl
import subprocess
import sys
import tempfile
import threading
import time
import types
import urllib.request
import uuid
from multiprocessing import Process
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import refshim  # noqa: E402

DST_API, DST_TLS, SRC_API = 8081, 8080, 8083


def _apply(src: str, edits, what: str) -> str:
    for old, new in edits:
        assert src.count(old) == 1, f"{what}: anchor not found exactly once: {old!r}"
        src = src.replace(old, new)
    return src


SENDER_EDITS = [
    ('            with open(chunk_file_path, "rb") as f:\n                data = f.read()\n',
     '            lz4f_path = chunk_file_path.with_name(chunk_file_path.name + ".lz4f")\n'
     '            precompressed = lz4f_path.exists()                      # produced by gpu_compress\n'
     '            with open(lz4f_path if precompressed else chunk_file_path, "rb") as f:\n                data = f.read()\n'
     '            if precompressed:                                       # a frame, or a pointer into gpu_compress\'s shared arena\n'
     '                from skyplane_amd.gateway.shm_arena import take_payload\n'
     '                data = take_payload(lz4f_path, data)\n'),
    ('            assert len(data) == chunk.chunk_length_bytes, f"chunk {chunk_id} has size',
     '            assert precompressed or len(data) == chunk.chunk_length_bytes, f"chunk {chunk_id} has size'),
    ('            raw_wire_length = wire_length\n', '            raw_wire_length = chunk.chunk_length_bytes if precompressed else wire_length\n'),
    ('            if self.use_compression:\n                data = lz4.frame.compress(data)\n',
     '            if precompressed:\n                compressed_length = wire_length\n'
     '            elif self.use_compression:\n                data = lz4.frame.compress(data)\n'),
]

IDLE_EDITS = [    # INTEGRATION.md section 11 (a harness accommodation, not part of the drop-in): GatewayOperator.worker_loop busy-spins on an empty queue
    # (gateway_operator.py:84-88); with 128 connections that is ~400 spinning processes, and the GPU box's container may use 16 CPUs.  One line.
    ('                except queue.Empty:\n                    continue\n',
     '                except queue.Empty:\n                    time.sleep(IDLE_SLEEP_S)                # was: (nothing) -- a busy spin\n                    continue\n'),
]

DAEMON_EDITS = [   # one more branch in create_gateway_operators
    ('                elif op["op_type"] == "write_local":\n',
     '                elif op["op_type"] == "gpu_compress":\n'
     '                    operators[handle] = GatewayHipCompress(\n'
     '                        handle=handle, region=self.region, input_queue=input_queue, output_queue=output_queue,\n'
     '                        error_event=self.error_event, error_queue=self.error_queue, chunk_store=self.chunk_store,\n'
     '                        n_processes=op["num_workers"], max_batch=op["max_batch"], max_chunk_bytes=op["max_chunk_mb"] * 1024 * 1024,\n'
     '                        compute_md5=op["compute_md5"], cdc=op["cdc"], dedup=op["dedup"], context_factory=GPU_CONTEXT_FACTORY,\n'
     '                        handoff=op.get("handoff", "arena"))\n'
     '                    total_p += op["num_workers"]\n'
     '                elif op["op_type"] == "write_local":\n'),
]


def _emu_context_factory(device_id, max_chunk_bytes, max_batch):
    from skyplane_amd.hip_ops import ChunkResult
    from tests.emu import emulib

    class EmuContext:
        def process_batch(self, chunks, flags=3):
            frames, md5s, _ = emulib.process([bytes(c) for c in chunks], flags=flags)
            return [ChunkResult(frame=f, md5=m if flags & 2 else None) for f, m in zip(frames, md5s)]

        def close(self):
            pass

    return EmuContext()


def daemon_main(role, region, chunk_dir, program, info, api_port, work, gpu_op, log, context="emu", idle_sleep=0.0):
    """Child process: one reference GatewayDaemon."""
    fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(fd, 1)
    os.dup2(fd, 2)
    # Fork safety of the log plumbing.  The daemon forks its receiver servers and operator workers from a process that has other threads printing
    # (the API thread forks on POST /servers while the main loop prints chunk states): a child that inherits sys.stdout's buffer lock, or rich's
    # console lock, in the locked state blocks for ever at its first print()/logger call -- a receiver server then accepts its connection and never
    # reads it, and that connection's chunks are "sent" and never arrive (seen here: 15 of 16 chunks, about one run in eight).  Unbuffered text
    # streams have no lock to inherit; rich's lock is replaced in every child.
    import io

    sys.stdout = io.TextIOWrapper(io.FileIO(1, "w", closefd=False), write_through=True, errors="replace")
    sys.stderr = io.TextIOWrapper(io.FileIO(2, "w", closefd=False), write_through=True, errors="replace")

    def _fresh_locks():
        try:
            import rich

            rich.get_console()._lock = threading.RLock()
        except Exception:  # noqa: BLE001
            pass

    os.register_at_fork(after_in_child=_fresh_locks)
    refshim.install(Path(work) / f"shim_{role}")
    (Path(work) / f"{role}_program.json").write_text(json.dumps(program))
    (Path(work) / f"{role}_info.json").write_text(json.dumps(info))
    os.environ["GATEWAY_PROGRAM_FILE"] = str(Path(work) / f"{role}_program.json")
    os.environ["GATEWAY_INFO_FILE"] = str(Path(work) / f"{role}_info.json")
    import skyplane.gateway.gateway_daemon_api as api_mod

    api_mod.GatewayDaemonAPI.__init__.__defaults__ = ("127.0.0.1", api_port)     # host, port (two daemons on one machine)
    edits = (SENDER_EDITS if gpu_op else []) + (IDLE_EDITS if idle_sleep > 0 else [])
    if edits:        # the reference's operator module, patched in memory (sender cooperation for gpu_compress; the idle back-off of --idle-sleep)
        op_path = refshim.REFERENCE / "skyplane" / "gateway" / "operators" / "gateway_operator.py"
        op_mod = types.ModuleType("skyplane.gateway.operators.gateway_operator")
        op_mod.__file__ = str(op_path)
        op_mod.IDLE_SLEEP_S = idle_sleep
        exec(compile(_apply(op_path.read_text(), edits, "operator patches"), str(op_path), "exec"), op_mod.__dict__)
        sys.modules["skyplane.gateway.operators.gateway_operator"] = op_mod
    if gpu_op:
        import skyplane.chunk as ref_chunk
        import skyplane.gateway.chunk_store as ref_chunk_store
        import skyplane.gateway.gateway_queue as ref_queue

        sys.modules["skyplane_amd.chunk"] = ref_chunk
        sys.modules["skyplane_amd.gateway.chunk_store"] = ref_chunk_store
        sys.modules["skyplane_amd.gateway.gateway_queue"] = ref_queue
        from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress

        def _no_cpu_compress(data, **kw):          # every chunk has a frame from gpu_compress: the CPU codec must stay idle
            raise AssertionError("the sender compressed on the CPU although gpu_compress had left a frame")

        op_mod.lz4 = types.SimpleNamespace(frame=types.SimpleNamespace(compress=_no_cpu_compress))
        d_path = refshim.REFERENCE / "skyplane" / "gateway" / "gateway_daemon.py"
        d_mod = types.ModuleType("skyplane.gateway.gateway_daemon")
        d_mod.__file__ = str(d_path)
        d_mod.GatewayHipCompress = GatewayHipCompress
        d_mod.GPU_CONTEXT_FACTORY = _emu_context_factory if context == "emu" else None      # None: SkyHipContext on a real GPU, created after the fork
        exec(compile(_apply(d_path.read_text(), DAEMON_EDITS, "daemon patch").replace('if __name__ == "__main__":', "if False:"), str(d_path), "exec"), d_mod.__dict__)
        GatewayDaemon = d_mod.GatewayDaemon
    else:
        from skyplane.gateway.gateway_daemon import GatewayDaemon
    os.makedirs(chunk_dir, exist_ok=True)
    daemon = GatewayDaemon(region=region, chunk_dir=chunk_dir, use_tls=False, use_e2ee=False, use_compression=True)
    daemon.run()


def tls_front(listen_port, target_port, cert, key, stop):
    """https://127.0.0.1:listen_port -> http://127.0.0.1:target_port (what stunnel does in the gateway image)."""
    ctx = ssl.SSLContext(ssl.PROTOCOL_TLS_SERVER)
    ctx.load_cert_chain(cert, key)
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", listen_port))
    srv.listen(256)
    srv.settimeout(0.2)

    def pump(a, b):
        try:
            while True:
                d = a.recv(65536)
                if not d:
                    break
                b.sendall(d)
        except OSError:
            pass
        finally:
            for s in (a, b):
                try:
                    s.shutdown(socket.SHUT_RDWR)
                except OSError:
                    pass

    def serve(raw):
        try:
            tls = ctx.wrap_socket(raw, server_side=True)
            up = socket.create_connection(("127.0.0.1", target_port))
        except (OSError, ssl.SSLError):
            raw.close()
            return
        threading.Thread(target=pump, args=(tls, up), daemon=True).start()
        threading.Thread(target=pump, args=(up, tls), daemon=True).start()

    while not stop.is_set():
        try:
            c, _ = srv.accept()
        except socket.timeout:
            continue
        threading.Thread(target=serve, args=(c,), daemon=True).start()
    srv.close()


def http_json(method, url, body=None, timeout=30):
    data = json.dumps(body).encode() if body is not None else None
    req = urllib.request.Request(url, data=data, method=method, headers={"Content-Type": "application/json"} if data else {})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return json.loads(r.read().decode())


def wait_api(port, deadline):
    while time.time() < deadline:
        try:
            http_json("GET", f"http://127.0.0.1:{port}/api/v1/status", timeout=2)
            return
        except Exception:  # noqa: BLE001
            time.sleep(0.1)
    raise RuntimeError(f"API on {port} did not come up")


def _reap_group():
    """SIGKILL every other member of this process group: the daemons' own SIGTERM handlers wait minutes for their
    senders (worker_exit polls the destination) and the receiver servers spin on closed sockets."""
    me, pg = os.getpid(), os.getpgrp()
    for _ in range(3):
        for d in os.listdir("/proc"):
            if not d.isdigit() or int(d) == me:
                continue
            try:
                if os.getpgid(int(d)) == pg:
                    os.kill(int(d), signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        time.sleep(0.1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=32)
    ap.add_argument("--chunk-kib", type=int, default=8192)
    ap.add_argument("--connections", type=int, default=4)
    ap.add_argument("--gpu-op", action="store_true")
    ap.add_argument("--context", choices=["emu", "hip"], default="emu", help="device behind gpu_compress: the CPU emulator (build container) or a real MI355X")
    ap.add_argument("--workers", type=int, default=1, help="gpu_compress worker processes")
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--stream", choices=["silesia", "random"], default="silesia", help="random = BASELINE configs[0]'s PRNG bytes")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--handoff", choices=["arena", "files"], default="arena", help="gpu_compress -> sender: slots of the shared arena or one payload file per chunk")
    ap.add_argument("--idle-sleep", type=float, default=0.0, help="seconds an idle reference operator worker sleeps instead of spinning (IDLE_EDITS; 0 = the reference as it is)")
    ap.add_argument("--io-workers", type=int, default=0, help="workers of read_object_store / write_object_store (0 = --connections, the reference's planner default); "
                                                             "--connections is always the sender's socket count")
    ap.add_argument("--out", default="")
    ap.add_argument("--keep-logs", action="store_true")
    ap.add_argument("--log-dir", default="", help="copy the daemons' full logs (API access lines removed) here when the run fails")
    a = ap.parse_args()
    for port in (DST_API, DST_TLS, SRC_API):    # a daemon left over from an earlier run would silently take our requests
        for attempt in range(30):                # (the workers of a run that ended seconds ago may still be on their way out: GPU call r6h)
            try:
                with socket.socket() as probe:
                    probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    probe.bind(("127.0.0.1", port))
                break
            except OSError:
                if attempt == 29:
                    raise
                time.sleep(1.0)
    try:
        os.setpgrp()                            # everything forked below is reaped as a group at the end
    except PermissionError:
        pass                                    # already a session (and group) leader
    base = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path(tempfile.gettempdir())
    work = Path(tempfile.mkdtemp(prefix="sky_daemon_", dir=base))
    sys.path.insert(0, str(ROOT))
    from skyplane_amd import synth

    size = a.chunk_kib << 10
    src_dir, dst_dir = work / "src_bucket", work / "dst_bucket"
    src_dir.mkdir()
    dst_dir.mkdir()
    unit = synth.silesia_like(min(a.chunks, 16) * size, config_id=2) if a.stream == "silesia" else synth.gen_random(synth.rng_for(1), min(a.chunks, 16) * size)
    chunks, datas = [], {}
    for i in range(a.chunks):
        cid = uuid.uuid4().hex
        d = unit[(i % 16) * size:((i % 16) + 1) * size].tobytes()
        (src_dir / f"obj{i}").write_bytes(d)
        datas[cid] = (str(dst_dir / f"obj{i}"), hashlib.md5(d).digest())
        chunks.append({"src_key": str(src_dir / f"obj{i}"), "dest_key": str(dst_dir / f"obj{i}"), "chunk_id": cid, "chunk_length_bytes": size,
                       "partition_id": "0", "mime_type": None, "md5_hash": None, "multi_part": False, "file_offset_bytes": 0, "part_number": None,
                       "upload_id": None})
    info = {"dst": {"public_ip_address": "127.0.0.1", "private_ip_address": "127.0.0.1"}}
    send = {"op_type": "send", "handle": "send", "target_gateway_id": "dst", "region": "local:dst", "num_connections": a.connections, "compress": True,
            "encrypt": False, "private_ip": False, "children": []}
    if a.gpu_op:
        mid = {"op_type": "gpu_compress", "handle": "gpu", "num_workers": a.workers, "max_batch": a.max_batch, "max_chunk_mb": 64, "compute_md5": True, "cdc": False,
               "dedup": False, "handoff": a.handoff, "children": [send]}
    else:
        mid = send
    src_program = [{"partitions": ["0"], "value": [{"op_type": "read_object_store", "handle": "read", "bucket_name": str(src_dir), "bucket_region": "local:src",
                                                    "num_connections": a.io_workers or a.connections, "children": [mid]}]}]
    dst_program = [{"partitions": ["0"], "value": [{"op_type": "receive", "handle": "recv", "decompress": True, "decrypt": False, "max_pending_chunks": 1000,
                                                    "children": [{"op_type": "write_object_store", "handle": "write", "bucket_name": str(dst_dir),
                                                                  "bucket_region": "local:dst", "num_connections": a.io_workers or a.connections, "key_prefix": "",
                                                                  "children": []}]}]}]
    cert, key = work / "cert.pem", work / "key.pem"
    subprocess.run(["openssl", "req", "-x509", "-newkey", "rsa:2048", "-nodes", "-keyout", str(key), "-out", str(cert), "-days", "1", "-subj", "/CN=skyplane"],
                   check=True, capture_output=True)
    stop = threading.Event()
    threading.Thread(target=tls_front, args=(DST_TLS, DST_API, str(cert), str(key), stop), daemon=True).start()
    if a.gpu_op and a.context == "emu":
        from tests.emu import emulib
        emulib.lib()                            # build once, before the daemons fork their workers
    procs = [Process(target=daemon_main, args=("dst", "local:dst", str(work / "dst_chunks"), dst_program, info, DST_API, str(work), False, str(work / "dst.log"), "emu", a.idle_sleep)),
             Process(target=daemon_main, args=("src", "local:src", str(work / "src_chunks"), src_program, info, SRC_API, str(work), a.gpu_op, str(work / "src.log"), a.context, a.idle_sleep))]
    res = {"what": "two reference GatewayDaemons on localhost: read_object_store(local) -> " + ("gpu_compress -> " if a.gpu_op else "") +
                   "send(compress) -> receive -> write_object_store(local)", "chunks": a.chunks, "chunk_bytes": size, "connections": a.connections,
           "host_cores": os.cpu_count(), "schedulable_cores": len(os.sched_getaffinity(0)), "gpu_op": bool(a.gpu_op), "context": a.context if a.gpu_op else None,
           "stream": a.stream, "gpu_workers": a.workers if a.gpu_op else None, "io_workers": a.io_workers or a.connections,
           "reference_idle_sleep_s": a.idle_sleep}
    ok = False
    try:
        for p in procs:
            p.start()
        wait_api(DST_API, time.time() + 90)
        wait_api(SRC_API, time.time() + 90)
        t0 = time.time()
        r = http_json("POST", f"http://127.0.0.1:{SRC_API}/api/v1/chunk_requests", chunks)
        assert r["n_added"] == len(chunks), r
        deadline = time.time() + a.timeout
        done = 0
        while time.time() < deadline:
            for port in (SRC_API, DST_API):
                errs = http_json("GET", f"http://127.0.0.1:{port}/api/v1/errors")["errors"]
                if errs:
                    raise RuntimeError(f"daemon on {port} reported: {errs[0][-2000:]}")
            log = http_json("GET", f"http://127.0.0.1:{DST_API}/api/v1/chunk_status_log")["chunk_status_log"]
            done = len({e["chunk_id"] for e in log if e["state"] == "complete" and e["handle"].startswith("write_object_store")})
            if done == len(chunks):
                break
            time.sleep(0.05)
        wall = time.time() - t0
        res["completed_chunks"] = done
        if done < len(chunks):
            # out of time: report what the DAG managed (a rate over the completed chunks) instead of nothing
            res.update({"wall_s": round(wall, 3), "gbit_s": round(done * size * 8 / wall / 1e9, 3), "gib_s": round(done * size / wall / 2**30, 4),
                        "verified": False, "timed_out": True})
            try:      # where did the missing chunks stop?  last state per operator, from both daemons' status logs
                fin = {e["chunk_id"] for e in log if e["state"] == "complete" and e["handle"].startswith("write_object_store")}
                missing = [c["chunk_id"] for c in chunks if c["chunk_id"] not in fin][:4]
                trail = {}
                for port in (SRC_API, DST_API):
                    for e in http_json("GET", f"http://127.0.0.1:{port}/api/v1/chunk_status_log")["chunk_status_log"]:
                        if e["chunk_id"] in missing:
                            trail.setdefault(e["chunk_id"], []).append(f"{port}:{e.get('handle')}:{e['state']}")
                res["stuck"] = trail
            except Exception as ex:  # noqa: BLE001
                res["stuck"] = repr(ex)
            raise TimeoutError(f"{done} of {len(chunks)} chunks completed in {wall:.0f} s")
        for cid, (path, dig) in datas.items():
            assert hashlib.md5(Path(path).read_bytes()).digest() == dig, path
        if a.gpu_op:
            prof = http_json("GET", f"http://127.0.0.1:{SRC_API}/api/v1/chunk_status_log")["chunk_status_log"]
            # (the API logs non-terminal operators' in_progress records only, gateway_daemon_api.py:147-153)
            gp = {e["chunk_id"] for e in prof if (e.get("handle") or "").startswith("gpu_compress") and e["state"] == "in_progress"}
            # (a lower bound: the API drops the records of a chunk that reach it after the chunk's terminal `complete`, gateway_daemon_api.py:107-109,
            # and with many worker processes feeding one status queue an operator's in_progress can arrive after the sender's complete.  That every
            # chunk went through the GPU operator is what the booby-trapped CPU compressor proves: a chunk without a frame would have raised there.)
            assert len(gp) > 0, "gpu_compress logged nothing"
            assert "compressed on the CPU" not in (work / "src.log").read_text()
            res["gpu_compress_chunks_logged"] = len(gp)
            res["cpu_compress_calls_in_sender"] = 0
        raw = len(chunks) * size
        res.update({"wall_s": round(wall, 3), "gbit_s": round(raw * 8 / wall / 1e9, 3), "gib_s": round(raw / wall / 2**30, 4), "verified": True})
        ok = True
    finally:
        stop.set()
        try:
            if not ok or a.keep_logs:
                for n in ("src.log", "dst.log"):
                    if (work / n).exists():
                        lines = [l for l in (work / n).read_text(errors="replace").splitlines() if "werkzeug" not in l and "/api/v1/" not in l]
                        if a.log_dir and not ok:
                            Path(a.log_dir).mkdir(parents=True, exist_ok=True)
                            (Path(a.log_dir) / n).write_text("\n".join(lines) + "\n")
                        sys.stderr.write(f"---- {n} (tail, API access lines removed) ----\n" + "\n".join(lines[-60:])[-6000:] + "\n")
            line = json.dumps(res)
            print(line, flush=True)
            if a.out and (ok or res.get("timed_out")):
                Path(a.out).write_text(line + "\n")
        finally:
            shutil.rmtree(work, ignore_errors=True)
            _reap_group()
    os._exit(0 if ok else 1)       # not sys.exit: multiprocessing's atexit hook would try to join the (killed) daemons


if __name__ == "__main__":
    main()
