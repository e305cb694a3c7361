"""CPU tests that execute the SHIPPING kernel source (skyplane_amd/csrc/*.inc) under the fiber SIMT emulator
(tests/emu) and check it against the oracle.  These are logic tests of the kernels for a box without a GPU;
the parity tests proper are the `-m gpu` ones in test_gpu_parity.py, which call libskyhip.so through its C ABI."""
import hashlib

import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth
from tests.emu import emulib
from tests.model import lz4smodel


def _check(chunks, frames, md5s):
    for i, (d, f, m) in enumerate(zip(chunks, frames, md5s)):
        assert m == hashlib.md5(d).digest(), f"md5 chunk {i}"
        assert ref.lz4f_decompress(f, len(d)) == d, f"liblz4 rejects/mismatches chunk {i}"  # == gateway_receiver.py:196
        dec, info = ref.lz4f_decode(f, len(d), strict=True)                               # strict format rules
        assert dec == d and info["flg"] == 0x68 and info["bd"] == 0x40
        assert len(f) <= emulib.frame_bound(len(d))
        lz4smodel.check_frame(d, f)       # the slice-parallel parse is deterministic: byte-identical to its sequential model


def test_emu_small_cases_batch(small_cases):
    chunks = list(small_cases.values())
    frames, md5s, _ = emulib.process(chunks)
    _check(chunks, frames, md5s)


def test_emu_sub_batch_prefix_base(small_cases):
    chunks = [small_cases[k] for k in ("records_131077", "empty", "abc_run", "one")]
    frames, md5s, _ = emulib.process(chunks, blk_skew=37)
    _check(chunks, frames, md5s)
    frames0, _, _ = emulib.process(chunks, blk_skew=0)
    assert frames == frames0


def test_emu_empty_frame_is_valid():
    frames, md5s, _ = emulib.process([b""])
    assert len(frames[0]) == 19 and md5s[0].hex() == "d41d8cd98f00b204e9800998ecf8427e"
    assert ref.lz4f_decompress(frames[0], 0) == b""


@pytest.mark.parametrize("n", [1, 4, 5, 11, 12, 13, 14, 55, 56, 57, 63, 64, 65, 119, 120, 121, 127, 128, 129, 4095, 65535, 65536, 65537, 131071, 131072, 131073])
def test_emu_ragged_lengths(n):
    rng = synth.rng_for(0, n)
    for gen in (synth.gen_text, synth.gen_sparse, synth.gen_random):
        d = gen(rng, n).tobytes()
        frames, md5s, _ = emulib.process([d])
        _check([d], frames, md5s)


def test_emu_every_class_compresses_like_the_reference():
    """Decoded output must be identical; compressed size must stay close to liblz4's with the reference's default
    (block-linked) preferences: within 10 % for every class (ours are independent 64 KiB blocks)."""
    for name in synth.CLASSES:
        d = synth.gen_class(name, 512 * 1024, synth.rng_for(9)).tobytes()
        frames, md5s, _ = emulib.process([d])
        _check([d], frames, md5s)
        assert len(frames[0]) <= 1.10 * len(ref.lz4f_compress(d)) + 64, name


def test_emu_long_matches_and_overlap():
    pats = [bytes(200_000), b"\x01" * 70_000 + b"\x02" * 70_000, (b"0123456789abcdef" * 9000), b"ab" * 40_000 + bytes(5) + b"ab" * 30_000,
            synth.gen_random(synth.rng_for(0, 1), 1000).tobytes() * 150]
    frames, md5s, _ = emulib.process(pats)
    _check(pats, frames, md5s)
    assert len(frames[0]) < 1200


def _emit_corner_cases():
    """Blocks aimed at the emit step: literal runs around 15 / 64 / 255-multiples (token nibble, extension bytes, the workgroup-wide copy of runs
    longer than 64), match lengths around 19 / 274 / 1039 (one, two, five extension bytes), runs that start or end a block, neighbours that meet
    inside an image dword."""
    rng = np.random.default_rng(7)

    def rnd(n):
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()

    cases = [rnd(1000) + bytes(60000) + rnd(300) + b"abcdefgh" * 400,
             rnd(70) + bytes(100) + rnd(200) + bytes(300) + rnd(1100) + b"xy" * 3000 + rnd(5000) + (rnd(37) * 500),
             b"".join(rnd(int(rng.integers(1, 400))) + bytes(int(rng.integers(4, 2000))) for _ in range(60))[:65536],
             b"".join(rnd(int(rng.integers(60, 90))) + (b"Q" * int(rng.integers(4, 30))) for _ in range(700))[:65536],
             (rnd(255 * 3 + 15 + 5) + bytes(19) + rnd(15) + bytes(270 + 4) + rnd(14) + bytes(269 + 4) + rnd(16) + bytes(1020 + 15 + 4 + 3)) * 8]
    unit = rnd(23) + b"ab" * 40 + rnd(9) + bytes(50)
    for n in (13, 64, 65, 100, 4095, 4096, 4097, 65535, 65536):
        cases.append((unit * (n // len(unit) + 1))[:n])
    return cases + [b"".join(cases)]


def test_emu_emit_corner_cases():
    cases = _emit_corner_cases()
    frames, md5s, _ = emulib.process(cases)
    _check(cases, frames, md5s)


def test_emu_incompressible_blocks_are_stored_raw():
    d = synth.gen_random(synth.rng_for(0, 2), 3 * 65536 + 100).tobytes()
    frames, md5s, cs = emulib.process([d])
    _check([d], frames, md5s)
    _, info = ref.lz4f_decode(frames[0], len(d))
    assert info["raw_blocks"] == 4 and len(frames[0]) == emulib.frame_bound(len(d))
    assert (cs >= np.array([65536, 65536, 65536, 100])).all()


def test_emu_full_chunk_8MiB(golden):
    d = synth.silesia_like(synth.CHUNK_BYTES, config_id=2).tobytes()
    assert hashlib.sha256(d).hexdigest() == golden["chunk_8MiB_silesia_like"]["data_sha256"]
    frames, md5s, _ = emulib.process([d])
    _check([d], frames, md5s)
    assert md5s[0].hex() == golden["chunk_8MiB_silesia_like"]["md5"]
    assert len(frames[0]) < 1.15 * golden["chunk_8MiB_silesia_like"]["liblz4_frame_len"]
