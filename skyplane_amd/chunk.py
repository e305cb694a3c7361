"""Wire-compatible mirror of the reference's chunk data formats (skyplane/chunk.py) -- the part of the
interface the GPU stage must leave byte-for-byte unchanged (BASELINE.json: "skyplane/chunk.py header untouched").

Inside the reference tree the operator imports ``skyplane.chunk`` itself; this module exists so the operator and
its tests run stand-alone (the reference package cannot be imported offline: skyplane/__init__.py pulls cloud
SDKs).  tests/test_host_formats.py pins ``WireProtocolHeader.to_bytes`` against bytes produced by the
reference's own class (tests/golden/golden.json).

Same names, fields, defaults and error behaviour as:
  Chunk               skyplane/chunk.py:9-43
  ChunkRequest        skyplane/chunk.py:47-76   (from_dict wraps a *Chunk* dict, :73-76)
  ChunkState          skyplane/chunk.py:79-92
  WireProtocolHeader  skyplane/chunk.py:95-167  (53 bytes, big-endian, magic "SKY_LARK", version 3)
"""
from __future__ import annotations

import socket
import struct
from dataclasses import asdict, dataclass
from enum import Enum, auto
from functools import total_ordering
from typing import Dict, Optional

_MAGIC = 0x534B595F4C41524B  # "SKY_LARK"
_VERSION = 3                 # v3 = uuid chunk ids
_HDR = struct.Struct(">QI16sQQBQ")
assert _HDR.size == 53


@dataclass
class Chunk:
    src_key: str
    dest_key: str
    chunk_id: str
    chunk_length_bytes: int
    partition_id: Optional[str] = None
    mime_type: Optional[str] = None
    md5_hash: Optional[bytes] = None  # 128 bits
    multi_part: Optional[bool] = False
    file_offset_bytes: Optional[int] = None
    part_number: Optional[int] = None
    upload_id: Optional[str] = None

    def to_wire_header(self, n_chunks_left_on_socket: int, wire_length: int, raw_wire_length: int, is_compressed: bool = False):
        return WireProtocolHeader(chunk_id=self.chunk_id, data_len=wire_length, raw_data_len=raw_wire_length, is_compressed=is_compressed,
                                  n_chunks_left_on_socket=n_chunks_left_on_socket)

    def as_dict(self):
        return asdict(self)

    @staticmethod
    def from_dict(d: Dict):
        return Chunk(**d)


@dataclass
class ChunkRequest:
    chunk: Chunk
    src_region: Optional[str] = None
    dst_region: Optional[str] = None
    src_type: Optional[str] = None
    dst_type: Optional[str] = None
    src_random_size_mb: Optional[int] = None
    src_object_store_bucket: Optional[str] = None
    dst_object_store_bucket: Optional[str] = None

    def __post_init__(self):
        if self.src_type == "object_store":
            assert self.src_object_store_bucket is not None
        elif self.src_type == "random":
            assert self.src_random_size_mb is not None
        if self.dst_type == "object_store":
            assert self.dst_object_store_bucket is not None

    def as_dict(self):
        out = asdict(self)
        out["chunk"] = self.chunk.as_dict()
        return out

    @staticmethod
    def from_dict(in_dict: Dict):
        return ChunkRequest(chunk=Chunk.from_dict(in_dict))


@total_ordering
class ChunkState(Enum):
    registered = auto()
    in_progress = auto()
    failed = auto()
    queued = auto()
    complete = auto()

    @staticmethod
    def from_str(s: str):
        return ChunkState[s.lower()]

    def __lt__(self, other):
        return self.value < other.value


@dataclass
class WireProtocolHeader:
    chunk_id: str
    data_len: int
    raw_data_len: int
    is_compressed: bool
    n_chunks_left_on_socket: int

    @staticmethod
    def magic_hex():
        return _MAGIC

    @staticmethod
    def protocol_version():
        return _VERSION

    @staticmethod
    def length_bytes():
        return _HDR.size

    @staticmethod
    def from_bytes(data: bytes):
        assert len(data) == _HDR.size, f"{len(data)} != {_HDR.size}"
        magic, version, cid, data_len, raw_len, comp, n_left = _HDR.unpack(data)
        if magic != _MAGIC:
            raise ValueError(f"Invalid magic number, got {magic:x} but expected {_MAGIC:x}")
        if version != _VERSION:
            raise ValueError(f"Invalid protocol version, got {version} but expected {_VERSION}")
        return WireProtocolHeader(chunk_id=cid.hex(), data_len=data_len, raw_data_len=raw_len, is_compressed=bool(comp), n_chunks_left_on_socket=n_left)

    def to_bytes(self):
        cid = bytes.fromhex(self.chunk_id)
        assert len(cid) == 16
        return _HDR.pack(_MAGIC, _VERSION, cid, self.data_len, self.raw_data_len, int(bool(self.is_compressed)), self.n_chunks_left_on_socket)

    @staticmethod
    def from_socket(sock: socket.socket):
        buf = b""
        while len(buf) < _HDR.size:
            got = sock.recv(_HDR.size - len(buf))
            if not got:
                raise ConnectionError("socket closed while reading chunk header")
            buf += got
        return WireProtocolHeader.from_bytes(buf)

    def to_socket(self, sock: socket.socket):
        assert sock.sendall(self.to_bytes()) is None
