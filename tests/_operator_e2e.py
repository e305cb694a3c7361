"""Helper process for tests/test_gpu_operator.py: runs the gpu_compress operator end to end.

It is a separate PROCESS on purpose: the operator forks its workers and each worker initialises HIP after the fork
(gateway_operator.py:66-70 semantics); that only works when the forking parent has never touched HIP, which is true of
the gateway daemon but not of a pytest process that already ran other GPU tests."""
import hashlib
import queue as pyqueue
import sys
import tempfile
import time
import uuid
from multiprocessing import Event, Queue
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import ref  # noqa: E402
from skyplane_amd import synth  # noqa: E402
from skyplane_amd.chunk import Chunk, ChunkRequest, WireProtocolHeader  # noqa: E402
from skyplane_amd.gateway.chunk_store import ChunkStore  # noqa: E402
from skyplane_amd.gateway.gateway_queue import GatewayQueue  # noqa: E402
from skyplane_amd.gateway.operators import hip_sender  # noqa: E402
from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress  # noqa: E402


def run(tmp_path):
    n, size = 12, 8 << 20
    store = ChunkStore(tmp_path / "chunks")
    q_in, q_out = GatewayQueue(), GatewayQueue()
    store.add_partition("0", q_in)
    stream = synth.mixed_chunks(4, size, config_id=4)
    reqs = []
    for i in range(n):
        cid = uuid.uuid4().hex
        data = stream[i % 4].tobytes() if i < 8 else synth.silesia_like(size - i * 1000, config_id=20 + i).tobytes()
        store.get_chunk_file_path(cid).write_bytes(data)
        reqs.append((ChunkRequest(chunk=Chunk(src_key=f"/s/{i}", dest_key=str(i), chunk_id=cid, chunk_length_bytes=len(data), partition_id="0")), data))
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "local:test", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=4, max_chunk_bytes=size,
                            device_ids=[0], cdc=True)
    for cr, _ in reqs:
        assert store.add_chunk_request(cr)[1]
    op.start_workers()
    done, t0 = [], time.time()
    while len(done) < n and time.time() - t0 < 180 and not err_ev.is_set():
        try:
            done.append(q_out.q.get(timeout=0.5))
        except pyqueue.Empty:
            pass
    op.stop_workers()
    assert not err_ev.is_set(), err_q.get() if not err_q.empty() else "error event set"
    assert len(done) == n
    total_raw = total_wire = 0
    for cr, data in reqs:
        hdr, payload = hip_sender.wire_payload(store, cr, 0)
        h = WireProtocolHeader.from_bytes(hdr.to_bytes())
        assert h.is_compressed and h.data_len == len(payload) and h.raw_data_len == len(data)
        assert ref.lz4f_decompress(payload, h.raw_data_len) == data
        assert hip_sender.chunk_digest(store, cr.chunk.chunk_id) == hashlib.md5(data).digest()
        total_raw += len(data); total_wire += len(payload)
    assert total_wire < total_raw
    recs = []
    while True:
        try:
            recs.append(store.chunk_status_queue.get(timeout=0.2))
        except pyqueue.Empty:
            break
    comp = [r for r in recs if r["state"] == "complete"]
    assert len(comp) == n and all(r["cdc_segments"] > 0 and len(r["md5_hex"]) == 32 for r in comp)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        run(Path(d))
    print("operator e2e ok")
