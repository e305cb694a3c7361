"""Host-side mirror of the reference's data formats vs bytes produced by the reference's own chunk.py (golden)."""
import pickle

import pytest

from skyplane_amd.chunk import Chunk, ChunkRequest, ChunkState, WireProtocolHeader


def test_wire_header_bytes_equal_reference(golden):
    for h in golden["headers"]:
        c = Chunk(src_key="s", dest_key="d", chunk_id=h["chunk_id"], chunk_length_bytes=h["raw_wire_length"])
        hdr = c.to_wire_header(n_chunks_left_on_socket=h["n_left"], wire_length=h["wire_length"], raw_wire_length=h["raw_wire_length"], is_compressed=h["is_compressed"])
        b = hdr.to_bytes()
        assert b.hex() == h["hex"] and len(b) == WireProtocolHeader.length_bytes() == 53
        assert WireProtocolHeader.from_bytes(b) == hdr


def test_wire_header_rejects_bad_magic_and_version(golden):
    b = bytearray(bytes.fromhex(golden["headers"][0]["hex"]))
    bad = bytes([b[0] ^ 1]) + bytes(b[1:])
    with pytest.raises(ValueError):
        WireProtocolHeader.from_bytes(bad)
    bad = bytes(b[:11]) + bytes([2]) + bytes(b[12:])
    with pytest.raises(ValueError):
        WireProtocolHeader.from_bytes(bad)
    with pytest.raises(AssertionError):
        WireProtocolHeader.from_bytes(bytes(b[:-1]))


def test_chunk_dict_shapes_equal_reference(golden):
    g = golden["chunk_dict"]
    c = Chunk.from_dict(g["as_dict"])
    assert c.as_dict() == g["as_dict"]
    cr = ChunkRequest.from_dict(g["as_dict"])          # from_dict takes a *Chunk* dict (chunk.py:73-76)
    assert cr.as_dict() == g["request_as_dict"]
    assert [s.name for s in ChunkState] == g["states"]
    assert ChunkState.registered < ChunkState.complete and ChunkState.from_str("IN_PROGRESS") is ChunkState.in_progress
    assert pickle.loads(pickle.dumps(cr)) == cr         # crosses multiprocessing.Queue


def test_chunk_request_asserts_like_reference():
    c = Chunk(src_key="a", dest_key="b", chunk_id="00" * 16, chunk_length_bytes=1)
    with pytest.raises(AssertionError):
        ChunkRequest(chunk=c, src_type="object_store")
    with pytest.raises(AssertionError):
        ChunkRequest(chunk=c, dst_type="object_store")
