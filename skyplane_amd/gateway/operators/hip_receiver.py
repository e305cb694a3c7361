"""Destination side of a gateway-to-gateway stream with the decode step on the GPU.

Follows GatewayReceiver.recv_chunks (skyplane/gateway/operators/gateway_receiver.py:142-237) for the parts this
stage touches: read the 53-byte WireProtocolHeader, recv_into the payload in <= recv_block_size pieces (:177-188),
decompress when header.is_compressed (:195-201), write <chunk_dir>/<chunk_id>.chunk and check its size against
raw_data_len (:204-218), stop when n_chunks_left_on_socket == 0 (:235-237).  End-to-end encryption (nacl) and the
socket profiler queue are out of scope.  A size mismatch raises instead of retrying forever."""
from __future__ import annotations

import mmap
import os
import socket
from typing import Callable, List, Optional

from skyplane_amd.chunk import WireProtocolHeader
from skyplane_amd.gateway import shm_arena, sidecar
from skyplane_amd.gateway.chunk_store import ChunkStore

MB = 1024 * 1024


def frame_bound(raw_len: int) -> int:
    """Largest LZ4 frame a conforming sender can make of raw_len bytes (15-byte header, 4 bytes per 64 KiB block, EndMark)."""
    return 15 + raw_len + 4 * ((raw_len + 65535) // 65536) + 4


def make_arena(chunk_store: ChunkStore, tag: str, max_chunk_bytes: int, n_slots: int) -> "shm_arena.ArenaWriter":
    """The receiver's side of the shared-memory hand-off (gateway/shm_arena.py): one arena per receiver process, shared by its connections; slots
    hold one wire payload each (a frame or a dedup recipe of a chunk of at most max_chunk_bytes)."""
    slot = frame_bound(max_chunk_bytes) + 33 + 21 * (max_chunk_bytes // 1024 + 2)
    return shm_arena.ArenaWriter(chunk_store.get_chunk_file_path("x").parent, f"rx_{tag}_{os.getpid()}", slot, n_slots)


def recv_chunks(conn: socket.socket, chunk_store: ChunkStore, decompress: Optional[Callable[[bytes, int], bytes]], recv_block_size: int = 4 * MB,
                max_chunk_bytes: int = 1024 * MB, arena: Optional["shm_arena.ArenaWriter"] = None) -> List[str]:
    """`decompress(frame, raw_len)` stands where lz4.frame.decompress stands in the reference, e.g.
    ``lambda f, n: ctx.decompress_batch([f], [n])[0]`` with a SkyHipContext.
    ``decompress=None`` defers the decode to the batching ``gpu_decompress`` operator (GatewayHipDecompress): a
    compressed payload is left as ``<chunk_id>.chunk.lz4f`` (written under a temporary name and renamed, so its
    existence means it is complete) and no ``<chunk_id>.chunk`` is written here."""
    received: List[str] = []
    while True:
        header = WireProtocolHeader.from_socket(conn)
        # the header is untrusted input: bound both lengths before anything is allocated (the reference trusts them, :177-188)
        # (a dedup recipe -- gateway/dedup_wire.py -- is a frame plus 33 bytes and 21 per segment of at least 1 KiB)
        if header.raw_data_len > max_chunk_bytes or header.data_len > frame_bound(max_chunk_bytes) + 33 + 21 * (max_chunk_bytes // 1024 + 2):
            raise ValueError(f"[Gateway] chunk {header.chunk_id}: header announces {header.data_len} wire / {header.raw_data_len} raw bytes, limit {max_chunk_bytes}")
        if header.is_compressed and decompress is None:
            final = sidecar.compressed_path(chunk_store, header.chunk_id)
            slots = arena.take(1) if (arena is not None and 0 < header.data_len <= arena.arena.slot_bytes) else []
            if slots:
                # shared-memory hand-off: the socket's bytes land in a slot of the arena gpu_decompress uploads from (page-locked there); what the
                # chunk directory gets is a pointer file, and gpu_decompress unlinking it frees the slot
                try:
                    view, got = memoryview(arena.arena.slot(slots[0])), 0
                    while got < header.data_len:
                        n = conn.recv_into(view[got:header.data_len], min(header.data_len - got, recv_block_size))
                        if n == 0:
                            raise ConnectionError(f"socket closed after {got} of {header.data_len} bytes of chunk {header.chunk_id}")
                        got += n
                except BaseException:
                    arena.give_back(slots[0])
                    raise
                arena.publish(slots[0], final, header.data_len)
                received.append(header.chunk_id)
                if header.n_chunks_left_on_socket == 0:
                    return received
                continue
            # deferred decode through a file: stream the payload to its sidecar in recv_block_size pieces, never holding it whole
            tmp = final.with_suffix(".rxtmp")
            got = 0
            with open(tmp, "w+b") as f:
                if header.data_len:
                    # the socket's bytes land in the file's pages (one copy: kernel socket buffer -> page cache), not in a bounce buffer first
                    f.truncate(header.data_len)
                    with mmap.mmap(f.fileno(), header.data_len) as mm, memoryview(mm) as view:
                        while got < header.data_len:
                            n = conn.recv_into(view[got:], min(header.data_len - got, recv_block_size))
                            if n == 0:
                                raise ConnectionError(f"socket closed after {got} of {header.data_len} bytes of chunk {header.chunk_id}")
                            got += n
            os.replace(tmp, final)
            received.append(header.chunk_id)
            if header.n_chunks_left_on_socket == 0:
                return received
            continue
        payload = bytearray(header.data_len)
        view, got = memoryview(payload), 0
        while got < header.data_len:
            n = conn.recv_into(view[got:], min(header.data_len - got, recv_block_size))
            if n == 0:
                raise ConnectionError(f"socket closed after {got} of {header.data_len} bytes of chunk {header.chunk_id}")
            got += n
        data = decompress(bytes(payload), header.raw_data_len) if header.is_compressed else bytes(payload)
        if len(data) != header.raw_data_len:
            raise ValueError(f"[Gateway] chunk {header.chunk_id}: {len(data)} bytes after decoding, header says {header.raw_data_len}")
        # (temporary name + rename, never a truncating open of <id>.chunk: after a retransmission that name may be a hard link to one of gpu_decompress's
        # page-locked slot files, and truncating it would free the pinned pages under the slot -- ADVICE r5)
        final = chunk_store.get_chunk_file_path(header.chunk_id)
        tmp = final.with_name(final.name + ".rawtmp")
        tmp.write_bytes(data)
        os.replace(tmp, final)
        received.append(header.chunk_id)
        if header.n_chunks_left_on_socket == 0:
            return received
