"""Wire-compatible stand-ins for the reference's chunk data formats (skyplane/chunk.py), the part of the interface the
GPU stage must leave byte-for-byte unchanged (BASELINE.json: "skyplane/chunk.py header untouched").

Inside the reference tree the operator imports ``skyplane.chunk`` itself; this module exists so the operator and its
tests run stand-alone (the reference package cannot be imported offline).  tests/test_host_formats.py pins the 53
header bytes and the dict shapes against values produced by the reference's own classes (tests/golden/golden.json).

Interface mirrored (names, fields, defaults, error types):
  WireProtocolHeader  skyplane/chunk.py:95-167   big-endian: magic u64 "SKY_LARK", version u32 = 3, chunk id 16 B,
                                                 data_len u64, raw_data_len u64, is_compressed u8, n_chunks_left u64
  Chunk               skyplane/chunk.py:9-43
  ChunkRequest        skyplane/chunk.py:47-76    (from_dict takes a *Chunk* dict, :73-76)
  ChunkState          skyplane/chunk.py:79-92
"""
from __future__ import annotations

import dataclasses
import enum
import socket
import struct
from typing import Any, Dict, Optional

_WIRE = struct.Struct(">Q I 16s Q Q B Q")     # 8 + 4 + 16 + 8 + 8 + 1 + 8 = 53 bytes
_SKY_LARK = int.from_bytes(b"SKY_LARK", "big")
_PROTO = 3                                     # 1 = base, 2 = compression flag, 3 = uuid chunk ids


@dataclasses.dataclass
class WireProtocolHeader:
    """Per-chunk preamble on a gateway-to-gateway TCP stream."""

    chunk_id: str                  # 32 hex characters
    data_len: int                  # bytes that follow on the socket (after compression / encryption)
    raw_data_len: int              # bytes of the chunk itself
    is_compressed: bool
    n_chunks_left_on_socket: int

    @staticmethod
    def magic_hex() -> int:
        return _SKY_LARK

    @staticmethod
    def protocol_version() -> int:
        return _PROTO

    @staticmethod
    def length_bytes() -> int:
        return _WIRE.size

    def to_bytes(self) -> bytes:
        raw_id = bytes.fromhex(self.chunk_id)
        assert len(raw_id) == 16
        return _WIRE.pack(_SKY_LARK, _PROTO, raw_id, self.data_len, self.raw_data_len, 1 if self.is_compressed else 0, self.n_chunks_left_on_socket)

    @staticmethod
    def from_bytes(data: bytes) -> "WireProtocolHeader":
        assert len(data) == _WIRE.size, f"{len(data)} != {_WIRE.size}"
        magic, version, raw_id, wire_len, raw_len, compressed, left = _WIRE.unpack(data)
        if magic != _SKY_LARK:
            raise ValueError(f"Invalid magic number, got {magic:x} but expected {_SKY_LARK:x}")
        if version != _PROTO:
            raise ValueError(f"Invalid protocol version, got {version} but expected {_PROTO}")
        return WireProtocolHeader(raw_id.hex(), wire_len, raw_len, compressed != 0, left)

    def to_socket(self, sock: socket.socket) -> None:
        assert sock.sendall(self.to_bytes()) is None

    @staticmethod
    def from_socket(sock: socket.socket) -> "WireProtocolHeader":
        pending = bytearray()
        while len(pending) < _WIRE.size:
            piece = sock.recv(_WIRE.size - len(pending))
            if not piece:
                raise ConnectionError("socket closed in the middle of a chunk header")
            pending += piece
        return WireProtocolHeader.from_bytes(bytes(pending))


class _Ordered:
    """ChunkState members compare by their position in the life cycle."""

    def __lt__(self, other):
        return self.value < other.value

    def __le__(self, other):
        return self.value <= other.value

    def __gt__(self, other):
        return self.value > other.value

    def __ge__(self, other):
        return self.value >= other.value


class ChunkState(_Ordered, enum.Enum):
    registered = 1
    in_progress = 2
    failed = 3
    queued = 4
    complete = 5

    @staticmethod
    def from_str(s: str) -> "ChunkState":
        return ChunkState[s.lower()]


@dataclasses.dataclass
class Chunk:
    """A contiguous piece of one object."""

    src_key: str
    dest_key: str
    chunk_id: str
    chunk_length_bytes: int
    partition_id: Optional[str] = None
    mime_type: Optional[str] = None
    md5_hash: Optional[bytes] = None          # 16-byte digest (see INTEGRATION.md 7 for why it travels out of band)
    multi_part: Optional[bool] = False
    file_offset_bytes: Optional[int] = None
    part_number: Optional[int] = None
    upload_id: Optional[str] = None

    def as_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    @staticmethod
    def from_dict(d: Dict[str, Any]) -> "Chunk":
        return Chunk(**d)

    def to_wire_header(self, n_chunks_left_on_socket: int, wire_length: int, raw_wire_length: int, is_compressed: bool = False) -> WireProtocolHeader:
        return WireProtocolHeader(self.chunk_id, wire_length, raw_wire_length, is_compressed, n_chunks_left_on_socket)


@dataclasses.dataclass
class ChunkRequest:
    """Gateway-local envelope around a Chunk."""

    chunk: Chunk
    src_region: Optional[str] = None
    dst_region: Optional[str] = None
    src_type: Optional[str] = None            # "object_store" | "random" | "read_local"
    dst_type: Optional[str] = None            # "object_store" | "save_local"
    src_random_size_mb: Optional[int] = None
    src_object_store_bucket: Optional[str] = None
    dst_object_store_bucket: Optional[str] = None

    def __post_init__(self):
        needs = {("src", "object_store"): self.src_object_store_bucket, ("src", "random"): self.src_random_size_mb}
        required = needs.get(("src", self.src_type), True)
        assert required is not None
        if self.dst_type == "object_store":
            assert self.dst_object_store_bucket is not None

    def as_dict(self) -> Dict[str, Any]:
        d = dataclasses.asdict(self)
        d["chunk"] = self.chunk.as_dict()
        return d

    @staticmethod
    def from_dict(in_dict: Dict[str, Any]) -> "ChunkRequest":
        return ChunkRequest(chunk=Chunk.from_dict(in_dict))
