"""Pins the CPU oracle (oracle/) against the golden vectors (tests/golden/golden.json) -- CPU only.

The oracle is only trustworthy as the GPU path's checker once its restatements agree with
(a) what the reference's own chunk.py emitted, (b) hashlib, (c) the system liblz4 the reference wraps.
"""
import hashlib

import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth


def test_md5_rfc1321_suite(golden):
    for msg, hx in golden["md5_rfc1321"].items():
        assert ref.md5(msg.encode()).hex() == hx


def test_md5_restatement_matches_hashlib_on_all_cases(golden, small_cases):
    for name, data in small_cases.items():
        assert hashlib.sha256(data).hexdigest() == golden["cases"][name]["data_sha256"], f"generator drifted: {name}"
        assert ref.md5(data).hex() == golden["cases"][name]["md5"] == hashlib.md5(data).hexdigest(), name


@pytest.mark.parametrize("n", [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 121, 127, 128, 129, 1000, 65536, 65537])
def test_md5_padding_boundaries(n):
    d = synth.gen_random(synth.rng_for(0, n), n).tobytes()
    assert ref.md5(d) == hashlib.md5(d).digest()


def test_wire_header_matches_reference_chunk_py(golden):
    for h in golden["headers"]:
        got = ref.wire_header(h["chunk_id"], h["wire_length"], h["raw_wire_length"], h["is_compressed"], h["n_left"])
        assert got.hex() == h["hex"]
        assert len(got) == 53


def test_liblz4_is_the_pinned_reference_dependency(golden):
    assert ref.liblz4_version() == golden["liblz4_version"]


def test_lz4_reference_frames_reproduce_and_decode(golden, small_cases):
    """liblz4 here produces the very frames recorded in golden.json, and BOTH decoders (liblz4 itself
    and the C restatement, strict mode) return the original bytes."""
    for name, data in small_cases.items():
        g = golden["cases"][name]
        frame = ref.lz4f_compress(data)
        assert len(frame) == g["liblz4_frame_len"] and hashlib.sha256(frame).hexdigest() == g["liblz4_frame_sha256"], name
        if "liblz4_frame_hex" in g:
            assert frame.hex() == g["liblz4_frame_hex"]
        assert ref.lz4f_decompress(frame, len(data)) == data
        dec, info = ref.lz4f_decode(frame, len(data), strict=True)
        assert dec == data, name
        # python-lz4 defaults: linked 64 KiB blocks + content size; liblz4 itself switches to
        # block-independent when the whole input fits one block, and treats contentSize==0 as "unknown"
        want = 0x60 if len(data) == 0 else (0x68 if len(data) <= 65536 else 0x48)
        assert info["flg"] == want and info["bd"] == 0x40, name


def test_golden_frames_decode_without_liblz4(golden, small_cases):
    """The stored hex frames decode with the restatement alone (no liblz4 involved)."""
    n = 0
    for name, g in golden["cases"].items():
        if "liblz4_frame_hex" in g:
            dec, _ = ref.lz4f_decode(bytes.fromhex(g["liblz4_frame_hex"]), g["len"], strict=True)
            assert dec == small_cases[name]
            n += 1
    assert n >= 5


def test_port_compressor_roundtrips_through_both_decoders(small_cases):
    for name, data in small_cases.items():
        frame = ref.lz4f_compress_port(data)
        assert ref.lz4f_decompress(frame, len(data)) == data, name
        dec, info = ref.lz4f_decode(frame, len(data), strict=True)
        assert dec == data and info["flg"] == 0x68


def test_decoder_rejects_malformed_frames(small_cases):
    data = small_cases["text_5000"]
    f = bytearray(ref.lz4f_compress_port(data))
    bad = bytearray(f); bad[0] ^= 1
    with pytest.raises(ref.OracleError):
        ref.lz4f_decode(bytes(bad), len(data))
    bad = bytearray(f); bad[14] ^= 0x10  # header checksum
    with pytest.raises(ref.OracleError):
        ref.lz4f_decode(bytes(bad), len(data))
    with pytest.raises(ref.OracleError):
        ref.lz4f_decode(bytes(f[:-1]), len(data))  # truncated endmark
    with pytest.raises(ref.OracleError):
        ref.lz4f_decode(bytes(f) + b"\0", len(data))  # trailing byte
    bad = bytearray(f); bad[6] ^= 1  # content size (also breaks HC)
    with pytest.raises(ref.OracleError):
        ref.lz4f_decode(bytes(bad), len(data))
    # liblz4 rejects the same corruptions
    with pytest.raises(ref.OracleError):
        ref.lz4f_decompress(bytes(f[:-1]), len(data))


def test_strict_end_of_block_rules():
    # token: 0 literals + match len 4 at offset 1 right at the end => violates "last sequence is literals only"
    blk = bytes([0x10, 0x41, 0x01, 0x00, 0x00])  # 1 literal 'A', match(off=1,len=4), then token 0x00 (0 literals)
    out = np.empty(64, np.uint8)
    src = np.frombuffer(blk, np.uint8)
    r = ref.sko().sko_lz4_block_decode(src.ctypes.data, src.size, out.ctypes.data, 64, 0, None, 0)
    assert r == 5
    r = ref.sko().sko_lz4_block_decode(src.ctypes.data, src.size, out.ctypes.data, 64, 1, None, 0)
    assert r < 0


def test_full_chunk_golden(golden):
    big = synth.silesia_like(synth.CHUNK_BYTES, config_id=2)
    g = golden["chunk_8MiB_silesia_like"]
    assert hashlib.sha256(big.tobytes()).hexdigest() == g["data_sha256"]
    assert ref.md5(big).hex() == g["md5"]
    fr = ref.lz4f_compress(big)
    assert len(fr) == g["liblz4_frame_len"]
    dec, info = ref.lz4f_decode(fr, big.size, strict=True)
    assert dec == big.tobytes() and info["blocks"] == 128


def test_gear_spec_frozen(golden, small_cases):
    g = ref.gear_table()
    assert hashlib.sha256(g.tobytes()).hexdigest() == golden["gear"]["table_sha256"]
    assert [int(x) for x in g[:4]] == golden["gear"]["table_first4"]
    cuts = ref.gear_cdc(small_cases["mixed_200k"])
    assert [int(x) for x in cuts] == golden["gear"]["cuts_mixed_200k"]
    assert [int(x) for x in ref.gear_cdc(synth.dedup_stream(1 << 20))] == golden["gear"]["cuts_dedup_1MiB"]


def test_gear_cdc_properties():
    d = synth.dedup_stream(4 << 20)
    cuts = ref.gear_cdc(d)
    seg = np.diff(np.concatenate([[0], cuts]))
    assert cuts[-1] == d.size and (seg[:-1] >= ref.CDC_MIN).all() and (seg <= ref.CDC_MAX).all()
    # numpy restatement of H(i): windowed sum over the last 64 bytes
    G = ref.gear_table()
    i = int(cuts[3]) - 1
    h = 0
    for k in range(64):
        h = (h + (int(G[d[i - k]]) << k)) & (2**64 - 1)
    seglen = int(seg[3])
    mask = ref.CDC_MASK_L if seglen >= ref.CDC_AVG else ref.CDC_MASK_S
    assert seglen == ref.CDC_MAX or (h & mask) == 0
    # content-defined: shifting the stream by a prefix re-synchronises the cuts
    pre = synth.gen_random(synth.rng_for(0, 5), 1234)
    cuts2 = ref.gear_cdc(np.concatenate([pre, d]))
    common = np.intersect1d(cuts + 1234, cuts2)
    assert common.size > 0.9 * cuts.size


def test_dedup_spec():
    fps = np.zeros((6, 16), np.uint8)
    fps[:, 0] = [1, 2, 1, 3, 2, 1]
    first = ref.dedup_first(fps, base_index=100)
    assert first.tolist() == [100, 101, 100, 103, 101, 100]
