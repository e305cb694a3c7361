"""Sender cooperation (SURVEY.md 8b "unavoidable"): what GatewaySender.process must do differently when the
gpu_compress operator ran upstream.  The reference sender re-reads the raw chunk, compresses it itself and only
then sets is_compressed (skyplane/gateway/operators/gateway_operator.py:343-372).  ``wire_payload`` is the
replacement for those lines: ship the pre-compressed sidecar when present, otherwise the raw bytes; the header,
the receiver (gateway_receiver.py:142-237) and chunk.py stay untouched.  INTEGRATION.md shows the ~10-line patch."""
from __future__ import annotations

from typing import Optional, Tuple

from skyplane_amd.chunk import ChunkRequest, WireProtocolHeader
from skyplane_amd.gateway import sidecar
from skyplane_amd.gateway.chunk_store import ChunkStore


def wire_payload(chunk_store: ChunkStore, chunk_req: ChunkRequest, n_chunks_left_on_socket: int) -> Tuple[WireProtocolHeader, bytes]:
    chunk = chunk_req.chunk
    frame_path = sidecar.compressed_path(chunk_store, chunk.chunk_id)
    if frame_path.exists():
        data = frame_path.read_bytes()
        header = chunk.to_wire_header(n_chunks_left_on_socket=n_chunks_left_on_socket, wire_length=len(data),
                                      raw_wire_length=chunk.chunk_length_bytes, is_compressed=True)
        return header, data
    data = chunk_store.get_chunk_file_path(chunk.chunk_id).read_bytes()
    assert len(data) == chunk.chunk_length_bytes, f"chunk {chunk.chunk_id} has size {len(data)} but should be {chunk.chunk_length_bytes}"
    header = chunk.to_wire_header(n_chunks_left_on_socket=n_chunks_left_on_socket, wire_length=len(data), raw_wire_length=len(data), is_compressed=False)
    return header, data


def send_chunks(sock, chunk_store: ChunkStore, chunk_reqs) -> int:
    """The per-connection send loop of GatewaySender.process (gateway_operator.py:343-402) minus its HTTP
    pre-registration and retry-on-reconnect: header, then payload, for every chunk; returns wire bytes sent."""
    sent = 0
    for idx, chunk_req in enumerate(chunk_reqs):
        header, payload = wire_payload(chunk_store, chunk_req, n_chunks_left_on_socket=len(chunk_reqs) - idx - 1)
        header.to_socket(sock)
        sock.sendall(payload)
        sent += len(payload)
    return sent


def chunk_digest(chunk_store: ChunkStore, chunk_id: str) -> Optional[bytes]:
    """The 16-byte digest Chunk.md5_hash is declared to carry (chunk.py:21), read from the side channel."""
    p = sidecar.digest_path(chunk_store, chunk_id)
    return bytes.fromhex(p.read_text().strip()) if p.exists() else None


def cleanup_sidecars(chunk_store: ChunkStore, chunk_id: str):
    """gateway_daemon_api.py:125-127 unlinks only <id>.chunk; whoever completes the chunk removes the sidecars."""
    for p in (sidecar.compressed_path(chunk_store, chunk_id), sidecar.digest_path(chunk_store, chunk_id)):
        try:
            p.unlink()
        except FileNotFoundError:
            pass
