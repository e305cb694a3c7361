#!/usr/bin/env python3
"""Regenerates tests/golden/golden.json.  Run ONLY in the build container (needs /root/reference).

Sources of truth, all executed here rather than re-typed:
  * wire headers   -- the reference's own skyplane/chunk.py, loaded by file path (the package
                      __init__ pulls cloud SDKs that are not installed), Chunk.to_wire_header(...).to_bytes()
                      (chunk.py:29-36, 141-155).
  * md5            -- hashlib.md5 exactly as skyplane/obj_store/s3_interface.py:181-192 drives it, plus the
                      RFC 1321 A.5 suite.
  * lz4 frames     -- the system liblz4 (1.9.3) through the call pattern of lz4.frame.compress(data)
                      (gateway_operator.py:359; python-lz4 defaults).  python-lz4 itself is not installed;
                      oracle/ref.py binds the same C library with the same preferences.
The GPU box has no /root/reference: tests only read golden.json.
"""
import hashlib
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import ref  # noqa: E402
from skyplane_amd import synth  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_chunk", "/root/reference/skyplane/chunk.py")
ref_chunk = importlib.util.module_from_spec(spec)
sys.modules["ref_chunk"] = ref_chunk
spec.loader.exec_module(ref_chunk)

out = {"liblz4_version": ref.liblz4_version(), "headers": [], "md5_rfc1321": {}, "cases": {}, "chunk_dict": None}

hdr_args = [
    ("000102030405060708090a0b0c0d0e0f", 0, 6672176, 8388608, True),
    ("ffffffffffffffffffffffffffffffff", 127, 8389139, 8388608, True),
    ("00000000000000000000000000000000", 0, 0, 0, False),
    ("0123456789abcdeffedcba9876543210", 2**40, 2**63 - 1, 2**63 - 1, False),
    ("a3f1c2d4e5b60718293a4b5c6d7e8f90", 5, 19, 0, True),
]
for cid, nleft, wl, rl, comp in hdr_args:
    c = ref_chunk.Chunk(src_key="s", dest_key="d", chunk_id=cid, chunk_length_bytes=rl)
    h = c.to_wire_header(n_chunks_left_on_socket=nleft, wire_length=wl, raw_wire_length=rl, is_compressed=comp)
    b = h.to_bytes()
    assert ref_chunk.WireProtocolHeader.from_bytes(b) == h
    out["headers"].append({"chunk_id": cid, "n_left": nleft, "wire_length": wl, "raw_wire_length": rl, "is_compressed": comp, "hex": b.hex()})

# Chunk.as_dict / ChunkRequest.from_dict shape (chunk.py:38-43, 66-76)
c = ref_chunk.Chunk(src_key="/a/b", dest_key="b", chunk_id="000102030405060708090a0b0c0d0e0f", chunk_length_bytes=8388608, partition_id="0",
                    file_offset_bytes=16777216, part_number=3, upload_id="u", multi_part=True)
d = c.as_dict()
cr = ref_chunk.ChunkRequest.from_dict(d)
out["chunk_dict"] = {"as_dict": d, "request_as_dict": cr.as_dict(), "states": [s.name for s in ref_chunk.ChunkState]}

for s, hx in [(b"", "d41d8cd98f00b204e9800998ecf8427e"), (b"a", "0cc175b9c0f1b6a831c399e269772661"), (b"abc", "900150983cd24fb0d6963f7d28e17f72"),
              (b"message digest", "f96b697d7cb7938d525a2f31aaf161d0"), (b"abcdefghijklmnopqrstuvwxyz", "c3fcd3d76192e4007dfb496cca67e13b"),
              (b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", "d174ab98d277d9f5a5611c2c9f419d9f"),
              (b"1234567890" * 8, "57edf4a22be3c955ac49da2e2107b67a")]:
    assert hashlib.md5(s).hexdigest() == hx
    out["md5_rfc1321"][s.decode()] = hx

for name, data in synth.small_cases().items():
    frame = ref.lz4f_compress(data)
    assert ref.lz4f_decompress(frame, len(data)) == data
    ent = {"len": len(data), "md5": ref.hashlib_md5(data).hex(), "data_sha256": hashlib.sha256(data).hexdigest(),
           "liblz4_frame_len": len(frame), "liblz4_frame_sha256": hashlib.sha256(frame).hexdigest()}
    if len(frame) <= 2048:
        ent["liblz4_frame_hex"] = frame.hex()
    out["cases"][name] = ent

# one full-size chunk class sample: pin generator determinism + reference digests at 8 MiB
big = synth.silesia_like(synth.CHUNK_BYTES, config_id=2)
fr = ref.lz4f_compress(big)
out["chunk_8MiB_silesia_like"] = {"md5": ref.hashlib_md5(big).hex(), "data_sha256": hashlib.sha256(big).hexdigest(), "liblz4_frame_len": len(fr),
                                  "liblz4_frame_sha256": hashlib.sha256(fr).hexdigest()}

# Gear CDC spec vectors (ours; frozen here)
g = ref.gear_table()
out["gear"] = {"table_sha256": hashlib.sha256(g.tobytes()).hexdigest(), "table_first4": [int(x) for x in g[:4]],
               "params": {"min": ref.CDC_MIN, "avg": ref.CDC_AVG, "max": ref.CDC_MAX, "mask_s": ref.CDC_MASK_S, "mask_l": ref.CDC_MASK_L},
               "cuts_mixed_200k": [int(x) for x in ref.gear_cdc(synth.small_cases()["mixed_200k"])],
               "cuts_dedup_1MiB": [int(x) for x in ref.gear_cdc(synth.dedup_stream(1 << 20))]}

(Path(__file__).parent / "golden.json").write_text(json.dumps(out, indent=1))
print("wrote golden.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})
