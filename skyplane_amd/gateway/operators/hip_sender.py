"""Sender cooperation (SURVEY.md 8b "unavoidable"): what GatewaySender.process must do differently when the
gpu_compress operator ran upstream.  The reference sender re-reads the raw chunk, compresses it itself and only
then sets is_compressed (skyplane/gateway/operators/gateway_operator.py:343-372).  ``wire_payload`` is the
replacement for those lines: ship the pre-compressed sidecar when present, otherwise the raw bytes; the header,
the receiver (gateway_receiver.py:142-237) and chunk.py stay untouched.  INTEGRATION.md shows the ~10-line patch."""
from __future__ import annotations

import array
import errno
import fcntl
import os
import termios
import time
import weakref
from collections import deque
from typing import Optional, Tuple

from skyplane_amd.chunk import ChunkRequest, WireProtocolHeader
from skyplane_amd.gateway import shm_arena, sidecar
from skyplane_amd.gateway.chunk_store import ChunkStore


def wire_payload(chunk_store: ChunkStore, chunk_req: ChunkRequest, n_chunks_left_on_socket: int) -> Tuple[WireProtocolHeader, bytes]:
    chunk = chunk_req.chunk
    frame_path = sidecar.compressed_path(chunk_store, chunk.chunk_id)
    if frame_path.exists():
        data = shm_arena.read_payload(frame_path)       # the payload file, or the arena slot a pointer file names
        header = chunk.to_wire_header(n_chunks_left_on_socket=n_chunks_left_on_socket, wire_length=len(data),
                                      raw_wire_length=chunk.chunk_length_bytes, is_compressed=True)
        return header, data
    data = chunk_store.get_chunk_file_path(chunk.chunk_id).read_bytes()
    assert len(data) == chunk.chunk_length_bytes, f"chunk {chunk.chunk_id} has size {len(data)} but should be {chunk.chunk_length_bytes}"
    header = chunk.to_wire_header(n_chunks_left_on_socket=n_chunks_left_on_socket, wire_length=len(data), raw_wire_length=len(data), is_compressed=False)
    return header, data


# ---- when may an arena slot be handed back? ----------------------------------------------------------------------------------------------------
# os.sendfile() from a tmpfs mapping is zero-copy: when it returns, the socket's send queue REFERENCES the arena's pages -- for the data that has
# not left yet and for whatever TCP may still have to retransmit (sendfile(2): the file must stay unmodified until the peer has the data).  Unlinking
# the pointer file at that moment (round 3 did) lets gpu_compress DMA the next frame into the slot while the old frame's tail is still queued: silent
# corruption of the wire payload under a real RTT, and our frames carry no checksum.  So a released frame is only NOTED here with the socket's running
# byte count, and its pointer is unlinked when the kernel says that many bytes have been acknowledged (SIOCOUTQ = unacknowledged bytes of the send queue).
class _SockLedger:
    __slots__ = ("sent", "pending")

    def __init__(self):
        self.sent = 0                  # bytes handed to this socket by send_chunk so far
        self.pending = deque()         # (byte count at the end of the frame, pointer path) in send order


# Keyed by the socket OBJECT, weakly: a ledger keyed by sock.fileno() outlived its socket, and the reconnect that got the same descriptor
# number inherited the dead connection's byte count and pending list -- the retry of a chunk then freed its slot on the strength of bytes the OLD
# connection had counted.  A ledger now dies with its socket (or with `forget`), whatever number the kernel hands out next.
_LEDGERS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _ledger(sock, create: bool = False) -> Optional[_SockLedger]:
    led = _LEDGERS.get(sock)
    if led is None and create:
        led = _LEDGERS[sock] = _SockLedger()
    return led


def _unacked(sock) -> int:
    buf = array.array("i", [0])
    fcntl.ioctl(sock.fileno(), termios.TIOCOUTQ, buf)       # SIOCOUTQ on a TCP socket: bytes written and not yet acknowledged by the peer
    return int(buf[0])


def _unlink_quiet(path):
    try:
        path.unlink()
    except FileNotFoundError:
        pass


def _still_referenced(path, but_not) -> bool:
    """Is `path` (a pointer file = an arena slot) still waiting for acknowledgement on ANOTHER live socket?  A chunk that is sent again after a
    connection died has one entry per attempt; the slot goes back when the last of them is acknowledged -- or forgotten."""
    for sk, led in list(_LEDGERS.items()):
        if sk is not but_not and any(q == path for _c, q in led.pending):
            return True
    return False


def release_acked(sock) -> int:
    """Unlink the pointer files of this socket's released frames whose bytes the peer has acknowledged; returns how many are still waiting."""
    led = _ledger(sock)
    if led is None or not led.pending:
        return 0
    try:
        if sock.fileno() < 0:              # closed: what the kernel still has queued cannot be asked about any more -- wait for forget()
            return len(led.pending)
        acked = led.sent - _unacked(sock)
    except OSError as e:
        if e.errno in (errno.EBADF, errno.ENOTCONN, errno.EPIPE, errno.ECONNRESET):      # the socket is dead: what it had in flight is not acknowledged and never will be
            return len(led.pending)
        acked = led.sent                   # not a TCP socket (a test's socketpair): nothing is ever retransmitted, delivered = copied
    while led.pending and led.pending[0][0] <= acked:
        path = led.pending.popleft()[1]
        if not any(q == path for _c, q in led.pending) and not _still_referenced(path, sock):      # (the same chunk sent twice: the later entry decides)
            _unlink_quiet(path)
    return len(led.pending)


def forget(sock) -> int:
    """Call on EVERY close or error path of a socket that sent frames with ``release=True``: drops its ledger WITHOUT unlinking anything -- what was
    in flight on a dead connection was not acknowledged, its chunks will be sent again (same pointer file, another socket) and released there.
    Returns how many frames were still pending."""
    led = _LEDGERS.pop(sock, None)
    return len(led.pending) if led else 0


def drain_releases(sock, timeout: float = 30.0) -> None:
    """Call before closing a socket that sent frames with ``release=True``: waits (bounded) until the peer has acknowledged everything, then frees
    the slots; on timeout the pointers stay -- a slot that is never freed costs capacity, a slot freed too early costs correctness.  The ledger
    is dropped either way (`forget`): nothing of it may reach a later socket."""
    end = time.monotonic() + timeout
    while release_acked(sock):
        if time.monotonic() > end:
            break
        time.sleep(0.0005)
    forget(sock)


def send_chunk(sock, chunk_store: ChunkStore, chunk_req: ChunkRequest, n_chunks_left_on_socket: int, release: bool = False) -> int:
    """Header + payload of one chunk, the payload straight from its file to the socket (``socket.sendfile`` -> os.sendfile on a plain TCP
    socket): the same bytes ``wire_payload`` + ``sendall`` put on the wire without the copy through a Python ``bytes`` -- the frame was
    produced by the GPU, the CPU has no reason to touch it.  Returns the payload bytes sent.  (A TLS-wrapped socket, as the reference uses
    when e2ee/TLS is on, falls back to read + send inside ``sendfile`` itself.)"""
    chunk = chunk_req.chunk
    frame_path = sidecar.compressed_path(chunk_store, chunk.chunk_id)
    compressed = frame_path.exists()
    if compressed:
        # the frame lives in a payload file or -- gpu_compress(handoff="arena") -- in a slot of the shared arena that `<id>.chunk.lz4f` points to:
        # either way its pages go to the socket by sendfile; `release` frees the sidecar (= the arena slot) once the peer has acknowledged the bytes
        # (release_acked / drain_releases above), never at sendfile's return
        size = shm_arena.open_payload(frame_path).length
        header = chunk.to_wire_header(n_chunks_left_on_socket=n_chunks_left_on_socket, wire_length=size, raw_wire_length=chunk.chunk_length_bytes, is_compressed=True)
        header.to_socket(sock)
        sent = shm_arena.sendfile_payload(sock, frame_path)
        if sent != size:
            raise ConnectionError(f"chunk {chunk.chunk_id}: {sent} of {size} payload bytes sent")
        led = _ledger(sock, create=True)
        led.sent += len(header.to_bytes()) + size
        if release:
            led.pending.append((led.sent, frame_path))
        release_acked(sock)
        return size
    path = chunk_store.get_chunk_file_path(chunk.chunk_id)
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        if not compressed:
            assert size == chunk.chunk_length_bytes, f"chunk {chunk.chunk_id} has size {size} but should be {chunk.chunk_length_bytes}"
        header = chunk.to_wire_header(n_chunks_left_on_socket=n_chunks_left_on_socket, wire_length=size, raw_wire_length=chunk.chunk_length_bytes,
                                      is_compressed=compressed)
        header.to_socket(sock)
        sent = sock.sendfile(f, 0, size) if size else 0
    if sent != size:
        raise ConnectionError(f"chunk {chunk.chunk_id}: {sent} of {size} payload bytes sent")
    return size


def send_chunks(sock, chunk_store: ChunkStore, chunk_reqs) -> int:
    """The per-connection send loop of GatewaySender.process (gateway_operator.py:343-402) minus its HTTP
    pre-registration and retry-on-reconnect: header, then payload, for every chunk; returns wire bytes sent."""
    sent = 0
    for idx, chunk_req in enumerate(chunk_reqs):
        sent += send_chunk(sock, chunk_store, chunk_req, n_chunks_left_on_socket=len(chunk_reqs) - idx - 1)
    return sent


def chunk_digest(chunk_store: ChunkStore, chunk_id: str) -> Optional[bytes]:
    """The 16-byte digest Chunk.md5_hash is declared to carry (chunk.py:21), read from the side channel."""
    p = sidecar.digest_path(chunk_store, chunk_id)
    return bytes.fromhex(p.read_text().strip()) if p.exists() else None


def attach_digest(chunk_dict: dict, chunk_store: ChunkStore) -> dict:
    """Source side of the checksum plumbing.  The sender pre-registers its chunks with the destination as
    ``json.dumps([c.chunk.as_dict() ...])`` (gateway_operator.py:299); ``Chunk.md5_hash`` is declared bytes (chunk.py:21) and bytes do not
    survive json.dumps, so the digest gpu_compress left in ``<id>.chunk.md5`` rides along as its hex string.  INTEGRATION.md section 7a."""
    p = sidecar.digest_path(chunk_store, chunk_dict["chunk_id"])
    if p.exists():
        chunk_dict = dict(chunk_dict, md5_hash=p.read_text().strip())
    return chunk_dict


def verified_digest(chunk, chunk_store: ChunkStore) -> Optional[bytes]:
    """Destination side: what ``GatewayObjStoreWriteOperator.process`` passes to
    ``upload_object(check_md5=...)`` (gateway_operator.py:633-643).  The registered digest (hex, from the source GPU) is compared with
    the digest of what actually arrived -- written next to the chunk by the receiver's `# todo check hash` edit or by gpu_decompress --
    and returned as the 16 raw bytes the object stores expect (s3_interface.py:203 base64-encodes them into Content-MD5).  A mismatch
    raises: the upload must not happen."""
    want = chunk.md5_hash
    if want is None:
        return None
    want = bytes.fromhex(want) if isinstance(want, str) else bytes(want)
    got = chunk_digest(chunk_store, chunk.chunk_id)
    if got is not None and got != want:
        raise ValueError(f"[Gateway] chunk {chunk.chunk_id}: digest of the received bytes {got.hex()} differs from the digest the source registered {want.hex()}")
    return want


def cleanup_sidecars(chunk_store: ChunkStore, chunk_id: str):
    """gateway_daemon_api.py:125-127 unlinks only <id>.chunk; whoever completes the chunk removes the sidecars."""
    for p in (sidecar.compressed_path(chunk_store, chunk_id), sidecar.digest_path(chunk_store, chunk_id)):
        try:
            p.unlink()
        except FileNotFoundError:
            pass
