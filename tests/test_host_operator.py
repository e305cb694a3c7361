"""Host logic of the gpu_compress operator on a GPU-less box: queue draining, status records, sidecars, error path,
sender cooperation, program-node registration, C-ABI surface, multi-rank sharding (gloo, world_size 2).

The HIP context is replaced by a TEST DOUBLE that produces frames with the oracle -- that is legal here because this
file tests plumbing, not arithmetic (the arithmetic is tested through the real C ABI in test_gpu_*.py), and it shows
the product has no CPU fallback: without the double the worker fails loudly (test_operator_fails_loudly_without_gpu)."""
import ctypes
import hashlib
import os
import queue as pyqueue
import re
import time
import uuid
from multiprocessing import Event, Queue
from pathlib import Path

import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth
from skyplane_amd.chunk import Chunk, ChunkRequest, ChunkState, WireProtocolHeader
from skyplane_amd.gateway import gateway_program
from skyplane_amd.gateway.chunk_store import ChunkStore
from skyplane_amd.gateway.gateway_queue import GatewayANDQueue, GatewayQueue
from skyplane_amd.gateway.operators import hip_sender
from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress, GatewayOperator

ROOT = Path(__file__).resolve().parents[1]


class _OracleContext:
    """Test double with SkyHipContext's process_batch signature."""

    def __init__(self, device_id, max_chunk_bytes, max_batch):
        self.device_id = device_id
        Path(os.environ["SKYTEST_DEVLOG"]).open("a").write(f"{os.getpid()} {device_id}\n")

    def process_batch(self, chunks, flags=3):
        from skyplane_amd.hip_ops import ChunkResult

        return [ChunkResult(frame=ref.lz4f_compress_port(c), md5=ref.md5(c) if flags & 2 else None) for c in chunks]

    def close(self):
        pass


def _factory(dev, mc, mb):
    return _OracleContext(dev, mc, mb)


class _ArenaContext(_OracleContext):
    """Same double plus SkyHipContext's staging interface (pinned_buffer / release_pinned / frame_bound /
    frames_into), over ordinary memory: exercises the operator's zero-copy read/write path on the CPU."""

    def pinned_buffer(self, nbytes):
        Path(os.environ["SKYTEST_DEVLOG"]).open("a").write(f"arena {nbytes}\n")
        return np.full(nbytes, 0xCD, np.uint8)

    def release_pinned(self, buf):
        Path(os.environ["SKYTEST_DEVLOG"]).open("a").write(f"release {buf.size}\n")

    @staticmethod
    def frame_bound(n):
        return 15 + n + 4 * ((n + 65535) // 65536) + 4

    def process_batch(self, chunks, flags=3, frames_into=None):
        from skyplane_amd.hip_ops import ChunkResult

        assert frames_into is not None and all(isinstance(c, np.ndarray) for c in chunks)
        res = []
        for c, out in zip(chunks, frames_into):
            f = np.frombuffer(ref.lz4f_compress_port(c.tobytes()), np.uint8)
            assert out.size >= self.frame_bound(c.size)
            out[: f.size] = f
            res.append(ChunkResult(frame=out[: f.size], md5=ref.md5(c.tobytes()) if flags & 2 else None))
        return res


def _arena_factory(dev, mc, mb):
    return _ArenaContext(dev, mc, mb)


class _DecodeContext(_OracleContext):
    """Double for the destination side: SkyHipContext.decompress_batch over liblz4 + hashlib."""

    def decompress_batch(self, frames, raw_lens, want_md5=False, into=None):
        outs = [ref.lz4f_decompress(bytes(f), n) for f, n in zip(frames, raw_lens)]
        return (outs, [hashlib.md5(o).digest() for o in outs]) if want_md5 else outs


def _decode_factory(dev, mc, mb):
    return _DecodeContext(dev, mc, mb)


def _make_store(tmp_path, n, size=70_000):
    store = ChunkStore(tmp_path / "chunks")
    q_in, q_out = GatewayQueue(), GatewayQueue()
    store.add_partition("0", q_in)
    reqs = []
    rng = synth.rng_for(11)
    for i in range(n):
        cid = uuid.uuid4().hex
        data = (synth.gen_text(rng, size) if i % 2 else synth.gen_random(rng, size)).tobytes()
        store.get_chunk_file_path(cid).write_bytes(data)
        cr = ChunkRequest(chunk=Chunk(src_key=f"/src/{i}", dest_key=f"{i}", chunk_id=cid, chunk_length_bytes=size, partition_id="0"))
        reqs.append((cr, data))
    return store, q_in, q_out, reqs


def _drain(q: Queue, n, timeout=30, stamps=None):
    out, t0 = [], time.time()
    while len(out) < n and time.time() - t0 < timeout:
        try:
            out.append(q.get(timeout=0.2))
            if stamps is not None:
                stamps.append(time.time())
        except pyqueue.Empty:
            pass
    return out


def test_operator_batches_shards_devices_and_writes_sidecars(tmp_path, monkeypatch):
    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    n = 21
    store, q_in, q_out, reqs = _make_store(tmp_path, n)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "local:test", q_in, q_out, err_ev, err_q, store, n_processes=2, max_batch=4, device_ids=[0, 1],
                            context_factory=_factory)
    for cr, _ in reqs:
        assert store.add_chunk_request(cr)[1]
    stamps = []
    op.start_workers()
    done = _drain(q_out.q, n, stamps=stamps)
    op.stop_workers()
    assert not err_ev.is_set(), err_q.get() if not err_q.empty() else ""
    assert sorted(c.chunk.chunk_id for c in done) == sorted(cr.chunk.chunk_id for cr, _ in reqs)
    # structural, not wall-clock: the batching loop never sleeps per chunk -- the reference's yield_sleep_s (0.1 s after every
    # chunk, gateway_operator.py:102) is not referenced by the lane loop at all
    import inspect
    assert "yield_sleep_s" not in inspect.getsource(GatewayHipCompress._lane_loop) and "yield_sleep_s" not in inspect.getsource(GatewayHipCompress._take_batch)
    # two workers -> two devices, round-robin by worker id
    devs = {int(l.split()[1]) for l in (tmp_path / "dev.log").read_text().split("\n") if l}
    assert devs == {0, 1}
    # status records: registered, in_progress, complete(+metadata) per chunk
    recs = _drain(store.chunk_status_queue, 3 * n)
    comp = [r for r in recs if r["state"] == "complete"]
    assert len(comp) == n and all(r["handle"] == "gpu_compress_0" and r["uncompressed_size_bytes"] == 70_000 and r["compressed_size_bytes"] > 0 for r in comp)
    # sidecars + the cooperating sender's wire bytes; receiver-side checks of gateway_receiver.py:150-218
    for cr, data in reqs:
        assert store.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == data           # raw file untouched
        hdr, payload = hip_sender.wire_payload(store, cr, n_chunks_left_on_socket=0)
        h2 = WireProtocolHeader.from_bytes(hdr.to_bytes())
        assert h2.is_compressed and h2.data_len == len(payload) and h2.raw_data_len == len(data)
        out = ref.lz4f_decompress(payload, h2.raw_data_len)                                 # == lz4.frame.decompress
        assert out == data and len(out) == h2.raw_data_len
        assert hip_sender.chunk_digest(store, cr.chunk.chunk_id) == hashlib.md5(data).digest()
        hip_sender.cleanup_sidecars(store, cr.chunk.chunk_id)
        hdr, payload = hip_sender.wire_payload(store, cr, 3)                                # no sidecar -> raw, like use_compression=False
        assert not hdr.is_compressed and payload == data and hdr.n_chunks_left_on_socket == 3


def test_operator_zero_copy_staging_path(tmp_path, monkeypatch):
    """With a context that offers pinned staging the operator reads chunk files straight into the arena and writes
    the frames out of it; arenas grow, never shrink; results are what the plain path produces."""
    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    store, q_in, q_out, reqs = _make_store(tmp_path, 7, size=70_001)         # odd size: views are 256-byte aligned, lengths are not
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "local:test", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=3, device_ids=[0],
                            context_factory=_arena_factory, handoff="files")
    op.worker_id = 0
    crs = [cr for cr, _ in reqs]
    assert op.process_batch(crs[:2]) == [True, True]
    assert op.process_batch(crs[2:5]) == [True] * 3                           # larger batch: both arenas are replaced by bigger ones
    assert op.process_batch(crs[5:6]) == [True]                               # smaller batch: arenas are reused
    assert op.process(crs[6]) is True
    log = (tmp_path / "dev.log").read_text().split("\n")
    assert sum(l.startswith("arena") for l in log) == 4 and sum(l.startswith("release") for l in log) == 2
    for cr, data in reqs:
        frame = store.get_compressed_file_path(cr.chunk.chunk_id).read_bytes()
        assert ref.lz4f_decompress(frame, len(data)) == data
        assert hip_sender.chunk_digest(store, cr.chunk.chunk_id) == hashlib.md5(data).digest()
    assert [m["uncompressed_size_bytes"] for m in op._last_metadata] == [70_001]
    # size invariant of gateway_operator.py:352, both directions
    for bad in (b"short", reqs[0][1] + b"x"):
        store.get_chunk_file_path(crs[0].chunk.chunk_id).write_bytes(bad)
        with pytest.raises(AssertionError, match="should be 70001"):
            op.process_batch(crs[:1])
    op.worker_exit(0)


def test_arena_slots_grow_when_a_larger_chunk_follows_a_small_first_batch(tmp_path, monkeypatch, capsys):
    """slots sized by a first batch of small chunks must not disable the arena for the rest of the transfer."""
    from skyplane_amd.gateway import shm_arena

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    store, q_in, q_out, reqs = _make_store(tmp_path, 2, size=70_001)
    big = synth.gen_text(synth.rng_for(5), 300_000).tobytes()
    cid = uuid.uuid4().hex
    store.get_chunk_file_path(cid).write_bytes(big)
    big_cr = ChunkRequest(chunk=Chunk(src_key="k", dest_key="k", chunk_id=cid, chunk_length_bytes=len(big), partition_id="0"))
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "local:test", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=3, device_ids=[0],
                            context_factory=_arena_factory, arena_slots=4)
    op.worker_id = 0
    crs = [cr for cr, _ in reqs]
    assert op.process_batch(crs) == [True, True]
    small = shm_arena.open_payload(store.get_compressed_file_path(crs[0].chunk.chunk_id))
    assert small.arena is not None and small.arena.slot_bytes < len(big)
    assert op.process_batch([big_cr]) == [True]
    pl = shm_arena.open_payload(store.get_compressed_file_path(cid))
    assert pl.arena is not None and pl.arena.slot_bytes >= len(big) and pl.arena.path != small.arena.path      # in a slot of the NEW arena, not a plain file
    assert ref.lz4f_decompress(shm_arena.read_payload(store.get_compressed_file_path(cid)), len(big)) == big
    assert ref.lz4f_decompress(shm_arena.read_payload(store.get_compressed_file_path(crs[0].chunk.chunk_id)), 70_001) == reqs[0][1]      # the old arena still serves its frames
    assert "new arena with slots of" in capsys.readouterr().out
    op.worker_exit(0)


def test_operator_shared_arena_handoff(tmp_path, monkeypatch):
    """SURVEY 8f item 2, "pinned shared memory instead of tmpfs files": with handoff="arena" (the default) a frame is produced straight into a slot of
    one shared arena file in the chunk directory; `<id>.chunk.lz4f` is a pointer to the slot; the cooperating sender sendfile()s the slot and unlinks
    the pointer, which frees the slot; when every slot is still waiting for its sender the operator falls back to one payload file per chunk."""
    import socket
    from skyplane_amd.gateway import shm_arena

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    store, q_in, q_out, reqs = _make_store(tmp_path, 8, size=70_001)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "local:test", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=3, device_ids=[0],
                            context_factory=_arena_factory, arena_slots=5)
    op.worker_id = 0
    crs = [cr for cr, _ in reqs]
    assert op.process_batch(crs[:3]) == [True] * 3 and op.process_batch(crs[3:6]) == [True] * 3
    arenas = list((tmp_path / "chunks").glob("_arena_*.shm"))
    assert len(arenas) == 1                                                      # one arena per lane, not one file per chunk
    kinds = []
    for cr, data in reqs[:6]:
        p = store.get_compressed_file_path(cr.chunk.chunk_id)
        pl = shm_arena.open_payload(p)
        kinds.append(pl.arena is not None)
        if pl.arena is not None:
            assert p.stat().st_size < 200 and pl.arena.path == arenas[0]         # a pointer, not the frame
        assert ref.lz4f_decompress(shm_arena.read_payload(p), len(data)) == data
        hdr, payload = hip_sender.wire_payload(store, cr, n_chunks_left_on_socket=0)
        assert hdr.is_compressed and hdr.data_len == len(payload) and ref.lz4f_decompress(payload, len(data)) == data
    assert kinds == [True] * 5 + [False]                                         # five slots, the sixth chunk fell back to a payload file
    # the sender: slot pages -> socket.  release=True frees the slot only when the peer HAS the bytes: sendfile is zero-copy, the socket's queue still
    # references the arena's pages when it returns (round 3 unlinked the pointer right away)
    a, b = socket.socketpair()
    a.setsockopt(socket.SOL_SOCKET, socket.SO_SNDBUF, 1 << 20)
    got = bytearray()
    sent = hip_sender.send_chunk(a, store, crs[0], n_chunks_left_on_socket=0, release=True)
    ptr = store.get_compressed_file_path(crs[0].chunk.chunk_id)
    assert ptr.exists() and hip_sender.release_acked(a) == 1                     # nothing read yet on the other end: the slot stays taken
    import threading
    rd = threading.Thread(target=lambda: [got.extend(x) for x in iter(lambda: b.recv(1 << 16), b"")])
    rd.start()
    hip_sender.drain_releases(a, timeout=10.0)
    assert not ptr.exists()
    a.close(); rd.join(); b.close()
    h = WireProtocolHeader.from_bytes(bytes(got[:53]))
    assert h.is_compressed and h.data_len == sent == len(got) - 53 and ref.lz4f_decompress(bytes(got[53:]), 70_001) == reqs[0][1]
    assert not store.get_compressed_file_path(crs[0].chunk.chunk_id).exists()
    # a connection dies with a frame unacknowledged; the reconnect gets the SAME descriptor number and sends the frame again.  The new
    # socket must start from an empty ledger (the old one was keyed by fileno and handed its byte count and pending list to the newcomer, which then
    # freed the slot on the strength of bytes the dead connection had counted), and the slot goes back only when the LATER send is acknowledged.
    ptr1 = store.get_compressed_file_path(crs[1].chunk.chunk_id)
    c1, d1 = socket.socketpair()
    fd_old = c1.fileno()
    hip_sender.send_chunk(c1, store, crs[1], n_chunks_left_on_socket=0, release=True)
    assert hip_sender.release_acked(c1) == 1 and ptr1.exists()
    assert hip_sender.forget(c1) == 1 and ptr1.exists()                          # the error path: ledger dropped, nothing unlinked
    c1.close(); d1.close()
    c2, d2 = socket.socketpair()
    if c2.fileno() != fd_old:                                                    # (make the descriptor number collide whatever the allocator did)
        os.dup2(c2.fileno(), fd_old); c2.close(); c2 = socket.socket(fileno=fd_old)
    assert hip_sender.release_acked(c2) == 0                                     # nothing inherited
    hip_sender.send_chunk(c2, store, crs[1], n_chunks_left_on_socket=0, release=True)
    assert hip_sender.release_acked(c2) == 1 and ptr1.exists()                   # the retry is in flight: the slot stays taken
    got2 = bytearray()
    rd2 = threading.Thread(target=lambda: [got2.extend(x) for x in iter(lambda: d2.recv(1 << 16), b"")])
    rd2.start()
    hip_sender.drain_releases(c2, timeout=10.0)
    assert not ptr1.exists()
    c2.close(); rd2.join(); d2.close()
    assert ref.lz4f_decompress(bytes(got2[53:]), len(reqs[1][1])) == reqs[1][1]
    # the same frame pending on two LIVE sockets (a retry while the first connection still drains): the first acknowledgement must not free the slot
    ptr2 = store.get_compressed_file_path(crs[2].chunk.chunk_id)
    e1, f1 = socket.socketpair(); e2, f2 = socket.socketpair()
    hip_sender.send_chunk(e1, store, crs[2], n_chunks_left_on_socket=0, release=True)
    hip_sender.send_chunk(e2, store, crs[2], n_chunks_left_on_socket=0, release=True)
    sink = bytearray()
    r1 = threading.Thread(target=lambda: [sink.extend(x) for x in iter(lambda: f1.recv(1 << 16), b"")]); r1.start()
    t_end = time.monotonic() + 10.0
    while hip_sender.release_acked(e1) and time.monotonic() < t_end:
        time.sleep(0.001)
    assert hip_sender.release_acked(e1) == 0 and ptr2.exists()                    # acknowledged on e1, still in flight on e2
    r2 = threading.Thread(target=lambda: [sink.extend(x) for x in iter(lambda: f2.recv(1 << 16), b"")]); r2.start()
    hip_sender.drain_releases(e2, timeout=10.0)
    assert not ptr2.exists()
    hip_sender.forget(e1)
    e1.close(); e2.close(); r1.join(); r2.join(); f1.close(); f2.close()
    assert op.process_batch(crs[6:8]) == [True] * 2
    k2 = [shm_arena.open_payload(store.get_compressed_file_path(cr.chunk.chunk_id)).arena is not None for cr in crs[6:8]]
    assert k2 == [True, True]                                                    # the freed slots were reused
    # untrusted pointer files: wrong arena name, range outside the arena
    bad = tmp_path / "chunks" / "bad.chunk.lz4f"
    bad.write_bytes(shm_arena._PTR.pack(shm_arena.MAGIC, 4096, 10, 11) + b"../../passwd")
    with pytest.raises(shm_arena.ArenaError):
        shm_arena.open_payload(bad)
    name = arenas[0].name.encode()
    bad.write_bytes(shm_arena._PTR.pack(shm_arena.MAGIC, 1 << 40, 10, len(name)) + name)
    with pytest.raises(shm_arena.ArenaError):
        shm_arena.open_payload(bad)
    op.worker_exit(0)


def test_decompress_operator_waits_decodes_verifies(tmp_path, monkeypatch):
    """Destination side: the receiver in deferred mode leaves wire payloads as sidecars, the
    gpu_decompress operator waits for them like GatewayWaitReceiver waits for chunk files, decodes in batches, checks
    length and digest, writes <id>.chunk."""
    import socket
    import threading
    from skyplane_amd.gateway.operators import hip_receiver
    from skyplane_amd.gateway.operators.gateway_operator import GatewayHipDecompress

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    n = 9
    src, _, _, reqs = _make_store(tmp_path / "src", n, size=50_000)
    for i, (cr, data) in enumerate(reqs):                                     # what gpu_compress leaves on the source side
        if i != 4:                                                            # chunk 4 travels uncompressed
            src.get_compressed_file_path(cr.chunk.chunk_id).write_bytes(ref.lz4f_compress_port(data))
        cr.chunk.md5_hash = [hashlib.md5(data).digest(), hashlib.md5(data).hexdigest(), None][i % 3]
    dst = ChunkStore(tmp_path / "dst" / "chunks")
    q_in, q_out = GatewayQueue(), GatewayQueue()
    dst.add_partition("0", q_in)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipDecompress("gpu_decompress_0", "local:dst", q_in, q_out, err_ev, err_q, dst, n_processes=1, max_batch=4, device_ids=[0],
                              context_factory=_decode_factory)
    for cr, _ in reqs:
        dst.add_chunk_request(cr)                                             # registered before anything has arrived
    op.start_workers()
    time.sleep(0.3)
    assert q_out.q.empty()                                                    # nothing decoded out of thin air
    a, b = socket.socketpair()
    rx = threading.Thread(target=lambda: hip_receiver.recv_chunks(b, dst, None))
    rx.start()
    hip_sender.send_chunks(a, src, [cr for cr, _ in reqs])
    rx.join(20)
    done = _drain(q_out.q, n)
    op.stop_workers()
    a.close(); b.close()
    assert not err_ev.is_set(), err_q.get() if not err_q.empty() else ""
    assert sorted(c.chunk.chunk_id for c in done) == sorted(cr.chunk.chunk_id for cr, _ in reqs)
    for cr, data in reqs:
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == data
        assert not dst.get_compressed_file_path(cr.chunk.chunk_id).exists()   # payload consumed
    recs = [r for r in _drain(dst.chunk_status_queue, 200, timeout=2) if r["state"] == "complete"]
    assert len(recs) == n and sum("md5_hex" in r for r in recs) >= 5


class _InPlaceDecodeContext(_ArenaContext):
    """Destination double WITH the staging interface: decompress_batch writes into `into` and returns views of it (what SkyHipContext does by DMA), and
    logs every register_host so that the test can see the slot files being page-locked once."""

    def register_host(self, buf):
        Path(os.environ["SKYTEST_DEVLOG"]).open("a").write(f"register {buf.size}\n")

    def unregister_host(self, buf):
        pass

    def decompress_batch(self, frames, raw_lens, want_md5=False, into=None):
        outs = []
        for f, n, o in zip(frames, raw_lens, into):
            d = np.frombuffer(ref.lz4f_decompress(bytes(f), n), np.uint8)
            o[: d.size] = d
            outs.append(o[: d.size])
        return (outs, [hashlib.md5(o.tobytes()).digest() for o in outs]) if want_md5 else outs


def test_decoded_chunks_are_published_as_hard_links_to_page_locked_slot_files(tmp_path, monkeypatch):
    """Raw side of the destination hand-off: the device writes a decoded chunk into a slot FILE's pages and
    <id>.chunk is a hard link to it -- an ordinary file of the right length for GatewayWaitReceiver / write_object_store (gateway_operator.py:125-149,
    :625-645), no write() of the chunk.  Deleting <id>.chunk (what the daemon does when the chunk is done) frees the slot; a chunk of another length and a
    batch that finds no free slot take the plain write path."""
    from skyplane_amd.gateway.operators.gateway_operator import GatewayHipDecompress

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    size = 40_000
    src, _, _, reqs = _make_store(tmp_path / "src", 7, size=size)
    short = (reqs[6][0], reqs[6][1][:12_345])                                  # an object's short tail
    short[0].chunk.chunk_length_bytes = len(short[1])
    reqs[6] = short
    dst = ChunkStore(tmp_path / "dst" / "chunks")
    q_in, q_out = GatewayQueue(), GatewayQueue()
    dst.add_partition("0", q_in)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipDecompress("gpu_decompress_0", "local:dst", q_in, q_out, err_ev, err_q, dst, n_processes=1, max_batch=4, device_ids=[0],
                              context_factory=lambda d, mc, mb: _InPlaceDecodeContext(d, mc, mb), out_slots=3)
    op.worker_id = 0
    for cr, data in reqs:
        cr.chunk.md5_hash = hashlib.md5(data).hexdigest()
        dst.get_compressed_file_path(cr.chunk.chunk_id).write_bytes(ref.lz4f_compress_port(data))
    first = [cr for cr, _ in reqs[:4]]
    assert op.process_batch(first) == [True] * 4
    slot_files = sorted((tmp_path / "dst" / "chunks").glob("_outslot_*"))
    assert len(slot_files) == 3 and all(f.stat().st_size == size for f in slot_files)
    inodes = {f.stat().st_ino for f in slot_files}
    linked = [dst.get_chunk_file_path(cr.chunk.chunk_id).stat().st_ino in inodes for cr in first]
    assert linked == [True, True, True, False]                                  # three slots: the fourth chunk was written the plain way
    for cr, data in reqs[:4]:
        f = dst.get_chunk_file_path(cr.chunk.chunk_id)
        assert f.read_bytes() == data and f.stat().st_size == size               # a file like any other for whoever uploads it
        assert not dst.get_compressed_file_path(cr.chunk.chunk_id).exists()
    log = (tmp_path / "dev.log").read_text().splitlines()
    assert log.count(f"register {size}") == 3                                    # page-locked once, when the slots were made
    # nothing is free while the chunks wait for their upload ...
    assert op.process_batch([reqs[4][0]]) == [True]
    assert dst.get_chunk_file_path(reqs[4][0].chunk.chunk_id).stat().st_ino not in inodes
    # ... the daemon deleting two finished chunks frees two slots (link count back to 1); the short tail never fits one
    dst.get_chunk_file_path(first[0].chunk.chunk_id).unlink()
    dst.get_chunk_file_path(first[2].chunk.chunk_id).unlink()
    assert op.process_batch([reqs[5][0], reqs[6][0]]) == [True, True]
    assert dst.get_chunk_file_path(reqs[5][0].chunk.chunk_id).stat().st_ino in inodes
    assert dst.get_chunk_file_path(reqs[6][0].chunk.chunk_id).stat().st_ino not in inodes
    assert dst.get_chunk_file_path(reqs[5][0].chunk.chunk_id).read_bytes() == reqs[5][1]
    assert dst.get_chunk_file_path(reqs[6][0].chunk.chunk_id).read_bytes() == reqs[6][1]
    assert dst.get_chunk_file_path(first[1].chunk.chunk_id).read_bytes() == reqs[1][1]      # a published chunk is not touched by later batches
    assert log.count(f"register {size}") == 3
    op.worker_exit(0)
    assert not list((tmp_path / "dst" / "chunks").glob("_outslot_*"))
    assert dst.get_chunk_file_path(first[1].chunk.chunk_id).read_bytes() == reqs[1][1]      # ... nor by the slots going away


def test_slot_files_follow_the_transfers_chunk_length(tmp_path, monkeypatch, capsys):
    """A lane whose FIRST batch held only an object's short tail must not keep tail-sized slot files for the rest of the transfer (the arena's lesson,
    ADVICE r4): once nothing is published in them they are made again for the length the chunks really have."""
    from skyplane_amd.gateway.operators.gateway_operator import GatewayHipDecompress

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    size = 30_000
    src, _, _, reqs = _make_store(tmp_path / "src", 4, size=size)
    tail = (reqs[0][0], reqs[0][1][:7_777])
    tail[0].chunk.chunk_length_bytes = len(tail[1])
    reqs[0] = tail
    dst = ChunkStore(tmp_path / "dst" / "chunks")
    q_in, q_out = GatewayQueue(), GatewayQueue()
    dst.add_partition("0", q_in)
    op = GatewayHipDecompress("gpu_decompress_0", "local:dst", q_in, q_out, Event(), Queue(), dst, n_processes=1, max_batch=4, device_ids=[0],
                              context_factory=lambda d, mc, mb: _InPlaceDecodeContext(d, mc, mb), out_slots=2)
    op.worker_id = 0
    for cr, data in reqs:
        dst.get_compressed_file_path(cr.chunk.chunk_id).write_bytes(ref.lz4f_compress_port(data))
    assert op.process_batch([reqs[0][0]]) == [True]                               # the tail alone: slots of 7777 bytes
    assert {f.stat().st_size for f in (tmp_path / "dst" / "chunks").glob("_outslot_*")} == {7_777}
    assert op.process_batch([reqs[1][0]]) == [True]                               # a full chunk while the tail is still published: plain write, same slots
    assert {f.stat().st_size for f in (tmp_path / "dst" / "chunks").glob("_outslot_*")} == {7_777}
    dst.get_chunk_file_path(reqs[0][0].chunk.chunk_id).unlink()                   # the daemon is done with the tail
    assert op.process_batch([reqs[2][0], reqs[3][0]]) == [True, True]
    slots = list((tmp_path / "dst" / "chunks").glob("_outslot_*"))
    assert {f.stat().st_size for f in slots} == {size} and "made again" in capsys.readouterr().out
    inodes = {f.stat().st_ino for f in slots}
    assert all(dst.get_chunk_file_path(reqs[k][0].chunk.chunk_id).stat().st_ino in inodes for k in (2, 3))
    for cr, data in reqs[1:]:
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == data
    op.worker_exit(0)


def test_decompress_operator_checksum_mismatch_is_an_error(tmp_path, monkeypatch):
    from skyplane_amd.gateway.operators.gateway_operator import GatewayHipDecompress

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    store, q_in, q_out, reqs = _make_store(tmp_path, 2, size=20_000)
    for cr, data in reqs:
        store.get_compressed_file_path(cr.chunk.chunk_id).write_bytes(ref.lz4f_compress(data))
        store.get_chunk_file_path(cr.chunk.chunk_id).unlink()
        cr.chunk.md5_hash = hashlib.md5(data).digest()
    reqs[1][0].chunk.md5_hash = hashlib.md5(b"something else").digest()
    err_ev, err_q = Event(), Queue()
    op = GatewayHipDecompress("gpu_decompress_0", "r", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=8, device_ids=[0], context_factory=_decode_factory)
    for cr, _ in reqs:
        store.add_chunk_request(cr)
    op.start_workers()
    assert err_ev.wait(20)
    tb = err_q.get(timeout=5)
    op.stop_workers()
    assert "checksum mismatch" in tb


def test_operator_error_path_matches_reference(tmp_path, monkeypatch):
    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    store, q_in, q_out, reqs = _make_store(tmp_path, 2)
    store.get_chunk_file_path(reqs[1][0].chunk.chunk_id).write_bytes(b"short")            # size mismatch -> assertion (like :352)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "r", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=8, device_ids=[0], context_factory=_factory)
    for cr, _ in reqs:
        store.add_chunk_request(cr)
    op.start_workers()
    assert err_ev.wait(20)
    tb = err_q.get(timeout=5)
    op.stop_workers()
    assert "AssertionError" in tb and "should be" in tb


def test_operator_fails_loudly_without_gpu(tmp_path):
    """No test double, no GPU in this container: the worker must raise into the reference's error path, never
    fall back to a CPU codec."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    store, q_in, q_out, reqs = _make_store(tmp_path, 1)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "r", q_in, q_out, err_ev, err_q, store, n_processes=1, device_ids=[0])
    store.add_chunk_request(reqs[0][0])
    op.start_workers()
    assert err_ev.wait(60)
    tb = err_q.get(timeout=5)
    op.stop_workers()
    assert "SkyHipError" in tb or "ImportError" in tb
    assert q_out.size() == 0 and not store.get_compressed_file_path(reqs[0][0].chunk.chunk_id).exists()


def test_base_operator_contract(tmp_path):
    class Echo(GatewayOperator):
        yield_sleep_s = 0.0

        def process(self, chunk_req, **args):
            return True

    store, q_in, q_out, reqs = _make_store(tmp_path, 3, size=10)
    err_ev, err_q = Event(), Queue()
    op = Echo("echo_0", "r", q_in, q_out, err_ev, err_q, store, n_processes=1)
    for cr, _ in reqs:
        store.add_chunk_request(cr)
    op.start_workers()
    assert len(_drain(q_out.q, 3)) == 3
    op.stop_workers()
    states = [r["state"] for r in _drain(store.chunk_status_queue, 9)]
    assert states.count("registered") == 3 and states.count("in_progress") == 3 and states.count("complete") == 3


def test_and_queue_fans_out():
    q = GatewayANDQueue()
    q.register_handle("a"); q.register_handle("b")
    q.put(1)
    assert q.get_handle_queue("a").q.get(timeout=5) == 1 and q.get_handle_queue("b").q.get(timeout=5) == 1
    with pytest.raises(ValueError):
        q.put_nowait(2)


def test_program_node_and_registration(tmp_path):
    node = gateway_program.GatewayGpuCompress(num_workers=8, max_batch=16, cdc=True)
    node.set_handle("n1")
    d = node.to_dict()
    assert d["op_type"] == "gpu_compress" and d["num_workers"] == 8 and d["max_batch"] == 16 and d["cdc"] is True and d["children"] == []
    store = ChunkStore(tmp_path / "c")
    op = gateway_program.create_operator(d, "gpu_compress_n1", "r", GatewayQueue(), GatewayQueue(), Event(), Queue(), store)
    assert isinstance(op, GatewayHipCompress) and op.n_processes == 8 and op.max_batch == 16 and op.cdc
    with pytest.raises(ValueError):
        gateway_program.create_operator({"op_type": "nope"}, "h", "r", None, None, None, None, store)
    from skyplane_amd.gateway.operators.gateway_operator import GatewayHipDecompress
    d2 = gateway_program.GatewayGpuDecompress(num_workers=2, max_batch=64, verify_md5=False).to_dict()
    op2 = gateway_program.create_operator(d2, "gpu_decompress_n2", "r", GatewayQueue(), GatewayQueue(), Event(), Queue(), store)
    assert isinstance(op2, GatewayHipDecompress) and op2.n_processes == 2 and op2.max_batch == 64 and op2.verify_md5 is False


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from skyplane_amd import _lib

    lib = _lib.load()
    header = (ROOT / "include" / "skyhip.h").read_text()
    declared = set(re.findall(r"\b(skyhip_[a-z0-9_]+)\s*\(", header))
    assert declared and declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.skyhip_abi_version() == 1
    assert lib.skyhip_frame_bound(8 << 20) == 8389139          # the reference's worst case for 8 MiB
    assert lib.skyhip_frame_bound(0) == 19 and lib.skyhip_frame_bound(1) == 24
    assert b"device" in lib.skyhip_strerror(-6)
    h = ctypes.c_void_p()
    assert lib.skyhip_create(0, 0, 1, ctypes.byref(h)) == -1     # argument validation needs no GPU
    # the host-side helper library (plain C: the device store's fingerprint map) against ITS header
    from skyplane_amd import _hostlib

    hl = _hostlib.load()
    declared = set(re.findall(r"\b(skyhost_[a-z0-9_]+)\s*\(", (ROOT / "include" / "skyhost.h").read_text()))
    assert declared == {"skyhost_map_new", "skyhost_map_free", "skyhost_map_count", "skyhost_map_put", "skyhost_map_get"}
    for name in declared:
        assert hasattr(hl, name), name


def _gloo_worker(rank, world, port, n_chunks, out_q):
    import torch.distributed as dist

    from skyplane_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_indices(n_chunks, rank, world)
    local, worst = shard.timed_region(lambda: time.sleep(0.05 * (rank + 1)), steps=2, dist=dist)
    out_q.put((rank, mine, local, worst))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_rank_sharding_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket

    with socket.socket() as sk:          # a free port chosen by the kernel
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 13, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    (r0, s0, l0, w0), (r1, s1, l1, w1) = res
    assert sorted(s0 + s1) == list(range(13)) and not set(s0) & set(s1) and s0 == list(range(0, 13, 2))
    # every rank reports the same, slowest-rank time; the slow rank slept 2 x 0.1 s inside the timed region
    assert w0 == pytest.approx(w1) and w0 >= max(l0, l1) - 1e-6 and w0 >= 0.19


def test_bench_py_rank_logic_world2_gloo():
    """bench.py's OWN multi-rank path (the code the driver launches with --gpus N): two ranks under torch.distributed.run, gloo,
    the shipping kernels under the emulator as the device.  Checks the configs[3] workload label, the rank-dependent streams, the
    all-ranks verification and the max-over-ranks timing -- on a box without a GPU."""
    import json
    import socket
    import subprocess
    import sys

    from tests.emu import emulib
    emulib.lib()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", "2", "--context", "emu", "--chunk-bytes", "131072", "--chunks", "8", "--unit-mib", "1", "--steps", "2",
           "--warmup", "1", "--max-batch", "4", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 2 and r["config"]["workload"].startswith("configs[3]: 2 MI355X")
    assert r["verified"]["all_ranks_ok"] and r["verified"]["digests_vs_hashlib"] == 8 and r["verified"]["frames_vs_liblz4"] == 8
    assert r["value"] > 0 and abs(r["value"] - 2 * 8 * 131072 * 2 / (r["ms_per_step"] * 2 / 1e3) / 2**30) < 0.01 * r["value"] + 1e-3
    # configs[3] "per-GPU + aggregate": one row per rank from that rank's own clock and kernel timers, the aggregate beside them
    assert [g["rank"] for g in r["per_gpu"]] == [0, 1] and all(g["GiBps"] > 0 and g["elapsed_s"] > 0 and "lz4_ms_per_step" in g and "frac" in g for g in r["per_gpu"])
    assert r["aggregate"]["GiBps"] == r["value"] and r["aggregate"]["sum_of_per_gpu_GiBps"] >= r["value"] - 0.002 and r["aggregate"]["slowest_rank"] in (0, 1)
    assert r["roofline"]["step"]["algorithmic_bytes"] > 2 * 8 * 131072 and 0 <= r["roofline"]["step"]["frac"] < 1 and r["roofline"]["step"]["ms"] > 0


def test_bench_py_bare_gpus2_launches_itself():
    """`python bench.py --gpus 2` with no launcher around it (the shape the driver uses for --gpus 1) must become two ranks on its own and print
    one JSON line; run with two resident halves so that the second half's frames replace the first's in the slots (configs[3])."""
    import json
    import subprocess
    import sys

    from tests.emu import emulib
    emulib.lib()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--context", "emu", "--chunk-bytes", "131072", "--chunks", "8", "--unit-mib", "1",
           "--steps", "1", "--warmup", "0", "--max-batch", "2", "--halves", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["chunks_per_gpu"] == 8 and "2 resident half" in r["config"]["workload"]
    # every digest of the step, every frame still resident (the second half's) checked on every rank
    assert r["verified"]["all_ranks_ok"] and r["verified"]["digests_vs_hashlib"] == 8 and r["verified"]["frames_vs_liblz4"] == 4


def test_bench_py_bare_gpus8_dry_run():
    """The shape of the driver's 8-GPU scaling run, dry: `python bench.py --gpus 8` becomes eight ranks (gloo, the emulator as the device), every
    rank holds its own share of the node's queue, every rank verifies, rank 0 alone prints ONE JSON line on stdout and the heartbeat goes to stderr."""
    import json
    import subprocess
    import sys

    from tests.emu import emulib
    emulib.lib()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--context", "emu", "--chunk-bytes", "131072", "--chunks", "4", "--unit-mib", "1",
           "--steps", "1", "--warmup", "0", "--max-batch", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and r["config"]["workload"].startswith("configs[3]: 8 MI355X") and r["config"]["chunks_per_gpu"] == 4
    assert r["verified"]["all_ranks_ok"] and r["verified"]["digests_vs_hashlib"] == 4 and r["verified"]["frames_vs_liblz4"] == 4
    assert abs(r["value"] - 8 * 4 * 131072 / (r["ms_per_step"] / 1e3) / 2**30) < 0.01 * r["value"] + 1e-3      # whole-job aggregate over the eight ranks
    assert [g["rank"] for g in r["per_gpu"]] == list(range(8)) and all(g["GiBps"] > 0 for g in r["per_gpu"]) and r["aggregate"]["sum_of_per_gpu_GiBps"] >= r["value"] - 0.005
    beats = [l for l in p.stderr.splitlines() if l.startswith("[bench +")]
    assert any("timed region" in l for l in beats) and any("verifying" in l for l in beats) and "secondary" not in r


def test_steady_state_e2e_script_with_emulated_device():
    """scripts/e2e_steady.py end to end on the CPU: both operators run the shipping kernel source under the emulator;
    sender threads, loopback TCP, deferred receiver, digest registration and the final verification are the real thing."""
    import json
    import subprocess
    import sys

    from tests.emu import emulib
    emulib.lib()
    p = subprocess.run([sys.executable, str(ROOT / "scripts" / "e2e_steady.py"), "--context", "emu", "--chunks", "10", "--chunk-kib", "64",
                        "--connections", "2", "--max-batch", "4", "--workers", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["verified"] and r["chunks"] == 10 and r["workers"] == 2
    # the same with the harness standing in for write_object_store + the daemon's clean-up (every chunk checked and deleted on arrival: what frees the
    # destination's slot files), two lanes per worker by default
    p = subprocess.run([sys.executable, str(ROOT / "scripts" / "e2e_steady.py"), "--context", "emu", "--chunks", "12", "--chunk-kib", "64",
                        "--connections", "2", "--max-batch", "4", "--workers", "1", "--dst-consume"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["dst_consume"]["chunks_deleted_on_arrival"] == 14 and r["dst_consume"]["hashed_on_the_cpu"] >= 1
    # ... and with a reader stage inside the timed region that downloads into gpu_compress's page-locked source slots
    p = subprocess.run([sys.executable, str(ROOT / "scripts" / "e2e_steady.py"), "--context", "emu", "--chunks", "12", "--chunk-kib", "64",
                        "--connections", "2", "--max-batch", "4", "--workers", "1", "--src-reader", "slots", "--in-slots", "6"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["verified"] and r["src_reader_writes"]["slots"] >= 6 and r["src_reader_writes"]["slots"] + r["src_reader_writes"]["files"] == 14


def test_source_slots_are_claimed_once_each_and_consumed_in_place(tmp_path, monkeypatch):
    """shm_arena.InSlots / claim_slot (round 6: the source's raw side).  Eight readers race for four slots: every slot ends up with exactly one chunk name,
    the others get none and write ordinary files; a chunk of another length never takes a slot; the operator hands slot-resident chunks to its context as
    views of the slot's mapping (no read of the file) and everything still comes out right; deleting the chunk frees the slot."""
    import threading

    from skyplane_amd.gateway import shm_arena, sidecar

    monkeypatch.setenv("SKYTEST_DEVLOG", str(tmp_path / "dev.log"))
    size = 96 * 1024
    store, q_in, q_out, _ = _make_store(tmp_path, 0)
    op = GatewayHipCompress("gpu_compress_0", "r", q_in, q_out, Event(), Queue(), store, n_processes=1, max_batch=16, max_chunk_bytes=size, device_ids=[0],
                            context_factory=_arena_factory, in_slots=4, in_slot_chunk_bytes=size, handoff="files")
    ctx = op._context()
    ins = op._in_slots(ctx, [size])
    assert ins is not None and ins.n == 4
    d = store.get_chunk_file_path("x").parent
    datas = {uuid.uuid4().hex: synth.gen_text(synth.rng_for(5, k), size if k < 8 else size - 1000).tobytes() for k in range(9)}
    got = {}

    def reader(cid):
        path = store.get_chunk_file_path(cid)
        claimed = shm_arena.claim_slot(path, len(datas[cid]))
        got[cid] = claimed
        with open(path, "r+b" if claimed else "wb") as f:
            f.write(datas[cid])

    ths = [threading.Thread(target=reader, args=(cid,)) for cid in datas]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    # never two names on one slot, the odd length never qualifies; normally all four slots are taken (two readers that link one slot at the same moment both let
    # go, and a sweep that saw a collision is repeated -- a slot left free by three collisions in a row is allowed, a slot with two chunks is not)
    assert 3 <= sum(got.values()) <= 4 and not got[list(datas)[8]]
    links = sorted(os.stat(p).st_nlink for p in d.glob("_inslot_*"))
    assert all(n in (1, 2) for n in links) and links.count(2) == sum(got.values())
    reqs = [ChunkRequest(chunk=Chunk(src_key=c, dest_key=c, chunk_id=c, chunk_length_bytes=len(b), partition_id="0")) for c, b in datas.items()]
    handed = []
    real = ctx.process_batch
    ctx.process_batch = lambda chunks, **kw: (handed.extend(chunks), real(chunks, **kw))[1]
    assert op.process_batch(reqs) == [True] * 9 and op._tls.in_slot_hits == sum(got.values())
    slot_addrs = {v.ctypes.data for v in ins.views}
    assert sum(isinstance(c, np.ndarray) and c.ctypes.data in slot_addrs for c in handed) == sum(got.values())
    for cid, b in datas.items():
        assert ref.lz4f_decompress(sidecar.compressed_path(store, cid).read_bytes(), len(b)) == b
    winner = next(c for c, ok in got.items() if ok)
    os.unlink(store.get_chunk_file_path(winner))
    shm_arena._claim_cache.clear()
    assert shm_arena.claim_slot(store.get_chunk_file_path("again"), size)
    if sum(got.values()) == 4:
        assert not shm_arena.claim_slot(store.get_chunk_file_path("nomore"), size)
    op.worker_exit(0)
    op.process_exit(0)
    assert not list(d.glob("_inslot_*"))


def test_lanes_collect_before_a_call(tmp_path):
    """_take_batch: a trickle of requests is collected for up to fill_wait_s instead of becoming one device call per request (a call costs ~80 ms
    whatever it holds: the MD5 of an 8 MiB chunk is one serial chain; profiles/r3_e2e_fill_wait.txt), and a full batch does not wait at all."""
    import threading
    import time

    store, q_in, q_out, reqs = _make_store(tmp_path, 48)
    err_ev, err_q = Event(), Queue()
    op = GatewayHipCompress("gpu_compress_0", "r", q_in, q_out, err_ev, err_q, store, n_processes=1, max_batch=32, device_ids=[0], context_factory=_factory, fill_wait_s=0.3)
    feeder = threading.Thread(target=lambda: [(time.sleep(0.02), store.add_chunk_request(cr)) for cr, _ in reqs[:6]])
    feeder.start()
    got = []
    while not got:
        got = op._take_batch()
    feeder.join()
    assert len(got) >= 5                                            # ONE batch for the trickle (the sixth request may still be in the queue's pipe)
    for cr, _ in reqs[6:48]:
        store.add_chunk_request(cr)
    time.sleep(0.2)                                                 # (multiprocessing.Queue.put hands over to a feeder thread)
    t = time.monotonic()
    assert len(op._take_batch()) == 32 and time.monotonic() - t < 0.1
