"""The REAL reference gateway code on one end of the wire, this repo's sender/receiver on the other.

Run by tests/test_reference_interop.py in a child interpreter (the import shim of oracle/refshim.py edits sys.path,
HOME and SKYPLANE_CONFIG).  Needs /root/reference, so it runs in the build container only.  TEST INFRASTRUCTURE ONLY.

  to_reference    this repo's hip_sender.send_chunks ships frames made by the shipping compressor source (CPU emulator)
                  -> reference GatewayReceiver (skyplane/gateway/operators/gateway_receiver.py:22-237, its own forked
                  server process, its own ChunkStore, lz4.frame.decompress over liblz4) -> files compared with the raw bytes.
  from_reference  reference GatewaySender.process (gateway_operator.py:268-412: read file, lz4.frame.compress, header,
                  sendall; only its HTTP control calls are answered by a stub) -> this repo's hip_receiver.recv_chunks
                  decoding with the shipping decompressor source (CPU emulator) -> files compared with the raw bytes.
"""
import json
import socket
import sys
import tempfile
import threading
import uuid
from multiprocessing import Event, Queue
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import refshim  # noqa: E402

scratch = Path(tempfile.mkdtemp(prefix="sky_interop_"))
refshim.install(scratch / "shim")

import numpy as np  # noqa: E402

from skyplane_amd import synth  # noqa: E402
from skyplane_amd import chunk as my_chunk  # noqa: E402
from skyplane_amd.gateway import chunk_store as my_store  # noqa: E402
from skyplane_amd.gateway.operators import hip_receiver, hip_sender  # noqa: E402
from tests.emu import emulib  # noqa: E402

import skyplane.chunk as ref_chunk  # noqa: E402  (the reference's, via the shim)
from skyplane.gateway.chunk_store import ChunkStore as RefChunkStore  # noqa: E402
from skyplane.gateway.gateway_queue import GatewayQueue as RefGatewayQueue  # noqa: E402
from skyplane.gateway.operators.gateway_operator import GatewaySender as RefGatewaySender  # noqa: E402
from skyplane.gateway.operators.gateway_receiver import GatewayReceiver as RefGatewayReceiver  # noqa: E402


def payloads():
    out = {}
    for name, data in synth.small_cases().items():
        if len(data) <= 300_000:
            out[name] = data
    for i, cls in enumerate(synth.CLASSES):
        out[f"class_{cls}"] = synth.gen_class(cls, 150_000 + 1000 * i, synth.rng_for(77, i)).tobytes()
    out["zeros_2blocks"] = bytes(2 * 65536)
    return out


def to_reference():
    data = payloads()
    ids = {name: uuid.uuid4().hex for name in data}
    src = my_store.ChunkStore(scratch / "src_chunks")
    reqs = []
    raw_passthrough = {"class_random"}           # no sidecar: the sender ships raw bytes with is_compressed=False
    names = list(data)
    frames, _, _ = emulib.process([data[n] for n in names], flags=1)
    for n, f in zip(names, frames):
        c = my_chunk.Chunk(src_key=n, dest_key=n, chunk_id=ids[n], chunk_length_bytes=len(data[n]), partition_id="0")
        src.get_chunk_file_path(ids[n]).write_bytes(data[n])
        if n not in raw_passthrough:
            src.get_compressed_file_path(ids[n]).write_bytes(f)
        reqs.append(my_chunk.ChunkRequest(chunk=c))

    error_event, error_queue = Event(), Queue()
    dst = RefChunkStore(str(scratch / "dst_chunks"))
    receiver = RefGatewayReceiver("recv", "local:dst", dst, error_event, error_queue, use_tls=False, use_compression=True)
    port = receiver.start_server()
    sock = socket.create_connection(("127.0.0.1", port))
    sent = hip_sender.send_chunks(sock, src, reqs)       # one stream, n_chunks_left_on_socket counts down to 0
    # the reference receiver returns from recv_chunks at n_left == 0; wait for the last file, then close
    import time
    deadline = time.time() + 60
    last = dst.get_chunk_file_path(ids[names[-1]])
    while time.time() < deadline and not (last.exists() and last.stat().st_size == len(data[names[-1]])):
        time.sleep(0.05)
    sock.close()
    # (WireProtocolHeader.from_socket spins on a closed connection and stop_server() waits 30 s for it: end the
    # forked server process directly instead)
    for p in receiver.server_processes:
        p.terminate()
        p.join(10)
    for n in names:
        got = dst.get_chunk_file_path(ids[n]).read_bytes()
        assert got == data[n], f"reference receiver wrote {len(got)} bytes for {n}, expected {len(data[n])}"
    print(f"OK to_reference chunks={len(names)} wire_bytes={sent} raw_bytes={sum(map(len, data.values()))}")


class _Reply:
    status = 200

    def __init__(self, body):
        self.data = json.dumps(body).encode()


class _ControlPlaneStub:
    """answers GatewaySender's pre-registration POST (gateway_operator.py:279-316); nothing else is called"""

    def request(self, method, url, body=None, headers=None):
        assert method == "POST" and url.endswith("/api/v1/chunk_requests"), (method, url)
        return _Reply({"status": "ok", "n_added": len(json.loads(body))})


def from_reference():
    data = payloads()
    names = list(data)
    ids = {name: uuid.uuid4().hex for name in data}
    src = RefChunkStore(str(scratch / "ref_src_chunks"))
    reqs = []
    for n in names:
        src.get_chunk_file_path(ids[n]).write_bytes(data[n])
        reqs.append(ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk(src_key=n, dest_key=n, chunk_id=ids[n], chunk_length_bytes=len(data[n]), partition_id="0")))

    dst = my_store.ChunkStore(scratch / "my_dst_chunks")
    lsock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    lsock.bind(("127.0.0.1", 0))
    lsock.listen()
    port = lsock.getsockname()[1]
    received, errors = [], []

    def decode(frame: bytes, raw_len: int) -> bytes:
        rc, outs, status = emulib.decompress([frame], [raw_len])
        if rc != 0:
            raise ValueError(f"frame rejected: {status}")
        return outs[0]

    def serve():
        conn, _ = lsock.accept()
        try:
            while len(received) < len(names):        # the reference sender sends one chunk per process() call, n_left = 0 each time
                received.extend(hip_receiver.recv_chunks(conn, dst, decode))
        except Exception as e:                        # noqa: BLE001
            errors.append(e)
        finally:
            conn.close()

    t = threading.Thread(target=serve, daemon=True)
    t.start()

    error_event, error_queue = Event(), Queue()
    sender = RefGatewaySender("send", "local:src", RefGatewayQueue(), RefGatewayQueue(), error_event, error_queue, src, ip_addr="127.0.0.1",
                              use_tls=False, use_compression=True, n_processes=1)
    sender.worker_id = 0
    sender.http_pool = _ControlPlaneStub()
    sock = socket.create_connection(("127.0.0.1", port))
    sender.destination_ports["127.0.0.1"] = port
    sender.destination_sockets["127.0.0.1"] = sock
    for r in reqs:
        assert sender.process(r, "127.0.0.1") is True
    t.join(120)
    sock.close()
    assert not errors, errors
    assert received == [ids[n] for n in names]
    for n in names:
        assert dst.get_chunk_file_path(ids[n]).read_bytes() == data[n], n
    print(f"OK from_reference chunks={len(names)}")


if __name__ == "__main__":
    {"to_reference": to_reference, "from_reference": from_reference}[sys.argv[1]]()
