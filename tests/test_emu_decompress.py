"""CPU tests: the shipping LZ4 frame DEcompressor source (skyplane_amd/csrc/lz4d_kernel.inc) under the SIMT emulator.
Parity = what lz4.frame.decompress does at gateway_receiver.py:195-201: frames made by the reference's dependency
(liblz4, block-LINKED, python-lz4 defaults) and by this library's compressor (block-independent) decode to the raw bytes;
malformed frames are rejected like liblz4 rejects them."""
import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth
from tests.emu import emulib


@pytest.fixture(autouse=True, params=["resolve+chain", "link"])
def link_path(request):
    """Block-linked frames take one of two ways through the library (sky_lz4d_run picks by batch size): every test of this file runs through both."""
    emulib.set_link_resolve(request.param == "resolve+chain")
    yield request.param
    emulib.set_link_resolve(True)


def test_emu_decode_reference_frames(small_cases):
    names = list(small_cases)
    frames = [ref.lz4f_compress(small_cases[k]) for k in names]       # == lz4.frame.compress(data): linked blocks
    rc, outs, status = emulib.decompress(frames, [len(small_cases[k]) for k in names])
    assert rc == 0
    for k, o, s, f in zip(names, outs, status, frames):
        assert s == 0 and o == small_cases[k], k      # includes the empty chunk, whose frame has no content size


def test_emu_decode_own_frames_roundtrip(small_cases):
    chunks = [v for v in small_cases.values()]
    frames, _, _ = emulib.process(chunks, flags=1)
    rc, outs, status = emulib.decompress(frames, [len(c) for c in chunks])
    assert rc == 0 and status == [0] * len(chunks)
    assert outs == chunks


@pytest.mark.parametrize("name", synth.CLASSES)
def test_emu_decode_every_class_linked_and_independent(name):
    d = synth.gen_class(name, 300_000, synth.rng_for(9)).tobytes()
    linked = ref.lz4f_compress(d)
    indep = ref.lz4f_compress_port(d)
    (ours,), _, _ = emulib.process([d], flags=1)
    rc, outs, status = emulib.decompress([linked, indep, ours], [len(d)] * 3)
    assert rc == 0 and status == [0, 0, 0] and outs == [d, d, d]


def test_emu_decode_overlap_periods_and_long_runs():
    pats = [bytes(200_000), b"ab" * 50_000, b"abc" * 40_000, b"0123456" * 20_000, (bytes(range(70)) * 3000), b"x" * 13 + b"y" * 70_000,
            synth.gen_random(synth.rng_for(0, 1), 1000).tobytes() * 150]
    frames = [ref.lz4f_compress(p) for p in pats]
    rc, outs, status = emulib.decompress(frames, [len(p) for p in pats])
    assert rc == 0 and outs == pats


def test_emu_decode_rejects_what_liblz4_rejects(small_cases):
    d = small_cases["mixed_200k"]
    good = ref.lz4f_compress_port(d)
    bad = []
    b = bytearray(good); b[0] ^= 1; bad.append(bytes(b))                 # magic
    b = bytearray(good); b[14] ^= 0x10; bad.append(bytes(b))             # header checksum
    bad.append(good[:-1])                                                # truncated EndMark
    bad.append(good + b"\\0")                                             # trailing byte
    b = bytearray(good); b[15 + 4 + 1] = 0; b[15 + 4 + 2] = 0            # first block: second byte... corrupt an offset to 0
    bad.append(bytes(b))
    b = bytearray(good); b[6] ^= 1; bad.append(bytes(b))                 # content size (breaks HC)
    rc, outs, status = emulib.decompress([good] + bad, [len(d)] * (1 + len(bad)))
    assert rc == -8 and status[0] == 0 and outs[0] == d
    for i, f in enumerate(bad):
        liblz4_ok = True
        try:
            liblz4_ok = ref.lz4f_decompress(f, len(d)) == d
        except ref.OracleError:
            liblz4_ok = False
        assert (status[1 + i] == 0) == liblz4_ok or outs[1 + i] in (b"", d), (i, status[1 + i])
        if status[1 + i] == 0:
            assert outs[1 + i] == d
    assert all(s != 0 for s in status[1:5])


def test_emu_decode_capacity_too_small(small_cases):
    d = small_cases["text_5000"]
    rc, outs, status = emulib.decompress([ref.lz4f_compress(d)], [len(d) - 1])
    assert rc == -8 and status == [7] and outs == [b""]


def test_emu_decode_full_chunk():
    d = synth.silesia_like(2 << 20, config_id=2).tobytes()
    rc, outs, status = emulib.decompress([ref.lz4f_compress(d)], [len(d)])
    assert rc == 0 and outs[0] == d


def test_emu_decode_frames_without_content_size():
    """python-lz4 store_size=False / liblz4 streaming producers: the length comes out of the last block."""
    rng = synth.rng_for(21)
    datas = [b"", b"a", synth.gen_class("text", 1000, rng).tobytes(), synth.gen_class("records", 65536, rng).tobytes(),
             synth.gen_class("text", 65537, rng).tobytes(), synth.gen_class("random", 150_000, rng).tobytes(), bytes(200_001)]
    for linked in (True, False):
        frames = [ref.lz4f_compress(d, store_size=False, block_linked=linked) for d in datas]
        assert all(not (f[4] & 0x08) for f in frames)
        # capacity == exact length, and capacity with slack: both must report the true decoded length
        for slack in (0, 77, 70_000):
            rc, outs, status = emulib.decompress(frames, [len(d) + slack for d in datas])
            assert rc == 0 and status == [0] * len(datas), (linked, slack, status)
            assert outs == datas
        # one byte short: rejected, nothing reported
        short = [i for i, d in enumerate(datas) if len(d)]
        rc, outs, status = emulib.decompress([frames[i] for i in short], [len(datas[i]) - 1 for i in short])
        assert rc == -8 and all(s != 0 for s in status) and all(o == b"" for o in outs)


@pytest.mark.parametrize("bsid,bmax", [(5, 256 << 10), (6, 1 << 20), (7, 4 << 20)])
def test_emu_decode_large_block_sizes(bsid, bmax):
    d = synth.gen_class("records", bmax + 12345, synth.rng_for(22, bsid)).tobytes()
    frames = [ref.lz4f_compress(d, block_size_id=bsid), ref.lz4f_compress(d, block_size_id=bsid, block_linked=False),
              ref.lz4f_compress(d, store_size=False, block_size_id=bsid)]
    rc, outs, status = emulib.decompress(frames, [len(d)] * 3)
    assert rc == 0 and status == [0, 0, 0] and outs == [d, d, d]


def test_frames_with_short_blocks_in_the_middle():
    """A producer that flushes (LZ4F_flush; python-lz4's LZ4FrameCompressor.flush) closes blocks early: the frame holds blocks shorter than the
    block maximum in its middle.  lz4.frame.decompress (gateway_receiver.py:195-201) takes them; round 2's decoder did not.
    Linked and independent, with and without a content size -- the last one is only found out while decoding (second, sequential pass)."""
    d = synth.gen_class("text", 300_000, synth.rng_for(5)).tobytes() + bytes(70_000) + synth.gen_class("records", 100_000, synth.rng_for(6)).tobytes()
    cuts = [0, 70_000, 70_001, 200_000, 200_013, 330_000, len(d)]
    pieces = [d[a:b] for a, b in zip(cuts, cuts[1:])]
    frames, want = [], []
    for linked in (True, False):
        for sized in (True, False):
            f = ref.lz4f_compress_stream(pieces, store_size=sized, block_linked=linked)
            assert ref.lz4f_decompress(f, len(d)) == d
            frames.append(f); want.append(d)
    frames.append(ref.lz4f_compress(d)); want.append(d)                       # a regular frame in the same batch
    rc, outs, status = emulib.decompress(frames, [len(w) for w in want])
    assert rc == 0 and status == [0] * len(frames) and outs == want
    # truncated / corrupted irregular frames are still rejected, and nothing is written past the capacity
    bad = frames[3][:-5]
    rc, outs, status = emulib.decompress([bad, frames[0]], [len(d), len(d) - 1])
    assert rc != 0 and status[0] != 0 and status[1] != 0


def link_stress_cases():
    """Blocks built to stress the link pass (sky_lz4_link): more matches than one super-batch holds (> 7168 per 64 KiB block), chains of matches that each
    copy the output of the one before, matches that reach across the block boundary, long runs next to very short matches."""
    rng = np.random.default_rng(77)
    words = [rng.integers(0, 256, 5, dtype=np.uint8).tobytes() for _ in range(200)]
    many = b"".join(words[int(k)] + bytes([int(b)]) for k, b in zip(rng.integers(0, 200, 40_000), rng.integers(0, 256, 40_000)))     # a match every 6 bytes
    chain = bytearray(rng.integers(0, 256, 37, dtype=np.uint8).tobytes())
    while len(chain) < 200_000:                                        # every piece copies a recent piece and changes one byte: long dependency chains
        back = int(rng.integers(20, 37)); n = int(rng.integers(8, 30))
        chain += bytes(chain[-back:][:n]) + bytes([int(rng.integers(0, 256))])
    chain = bytes(chain)
    straddle = (rng.integers(0, 256, 65_500, dtype=np.uint8).tobytes() + b"Q" * 300) * 3 + bytes(70_000) + many[:50_000]
    # a four-byte match behind every literal: ~12 600 matches per 64 KiB block -- more than the 8192 descriptors sky_lz4_resolve holds in registers (the rest it
    # reads again from memory) and six super-batches of sky_lz4_link
    dense = np.tile(np.frombuffer(b"abcdX", np.uint8), 40_000).copy()
    dense[4::5] = rng.integers(0, 256, 40_000, dtype=np.uint8)
    return [many, chain, straddle, many[:70_000] + bytes(100_000) + chain[:70_000], dense.tobytes()]


def test_emu_linked_frames_with_many_and_chained_matches():
    """The link pass -- 16 wavefronts per frame, a match copied as soon as the matches it reads from are done -- against liblz4's linked frames."""
    cases = link_stress_cases()
    frames = [ref.lz4f_compress(c) for c in cases]                     # python-lz4's defaults: linked blocks
    rc, outs, status = emulib.decompress(frames, [len(c) for c in cases])
    assert rc == 0 and status == [0] * len(cases)
    assert outs == cases


def test_emu_decode_crafted_sequence_streams():
    """Frames built sequence by sequence (tests/_crafted_frames.py): length fields around every nibble / extension-byte boundary, 64 three-byte sequences per
    window, batches above the LDS staging size, matches that begin before a batch and end inside it, fields longer than a window, a frame without content size."""
    from tests._crafted_frames import crafted_frames
    import ctypes
    cases = crafted_frames(1)
    names = list(cases)
    stats = (ctypes.c_uint32 * 8).in_dll(emulib.lib(), "sky_d_stats")
    for i in range(8):
        stats[i] = 0
    rc, outs, status = emulib.decompress([cases[k][0] for k in names], [len(cases[k][1]) for k in names])
    assert rc == 0, dict(zip(names, status))
    for k, o, s in zip(names, outs, status):
        assert s == 0 and o == cases[k][1], k
    # the frames reached every path of the batch decoder (skyplane_amd/csrc/lz4d_kernel.inc, SKY_D_STAT): batches put together in LDS, batches too big for it,
    # batches with a match across their start, the 64th sequence of a window, sequences walked byte by byte inside and outside the window, copies from the batch itself
    seen = dict(zip(("batches", "staged", "too_big", "straddle", "drop64", "walk_one", "serial_seq", "inside"), list(stats)))
    assert all(v > 0 for v in seen.values()), seen
