"""CPU tests: the shipping Gear-CDC / segment-MD5 / dedup kernel source under the SIMT emulator vs the frozen
specification in oracle/skyoracle.c (sko_gear_cdc / sko_dedup).  CDC and dedup are NOT in the reference
(SURVEY fact 0.3): parity here is against our own spec ("parity unpinned")."""
import hashlib

import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth
from tests.emu import emulib


def _expect(chunks):
    cuts = [ref.gear_cdc(c) for c in chunks]
    fps = []
    for c, cu in zip(chunks, cuts):
        st = 0
        for e in cu:
            fps.append(hashlib.md5(c[st:int(e)]).digest())
            st = int(e)
    return cuts, fps


def _check(chunks, out, prev_fps=()):
    prefix, seg_end, fps, first, base, _ = out
    cuts, efps = _expect(chunks)
    for i, cu in enumerate(cuts):
        got = seg_end[prefix[i]:prefix[i + 1]]
        assert len(got) == len(cu) and (got == cu).all(), f"cuts of chunk {i}"
    assert [fps[i].tobytes() for i in range(len(fps))] == efps
    allfp = list(prev_fps) + efps
    exp = ref.dedup_first(np.frombuffer(b"".join(allfp), np.uint8).reshape(-1, 16)) if allfp else np.zeros(0, np.uint64)
    assert base == len(prev_fps)
    assert (exp[len(prev_fps):] == first).all()
    return allfp



@pytest.fixture(autouse=True, params=["staged", "lane-streamed"])
def segment_digest_kernel(request):
    """The library has two segment-digest kernels (sky_segment_md5x: rows staged through LDS, the default; sky_segment_md5: SKYHIP_SEGMD5_STAGED=0): every test of
    this file runs through both."""
    emulib.set_segmd5_staged(request.param == "staged")
    yield request.param
    emulib.set_segmd5_staged(True)


def test_emu_cdc_matches_spec_and_dedup_persists(small_cases):
    G = ref.gear_table()
    e = emulib.EmuCdc()
    chunks = [synth.dedup_stream(1 << 20).tobytes(), small_cases["mixed_200k"], b"", bytes(100_000), small_cases["rand_4096"], small_cases["one"]]
    seen = _check(chunks, e.run(chunks, G))
    chunks2 = [small_cases["mixed_200k"], synth.dedup_stream(1 << 19, config_id=5).tobytes(), small_cases["rand_4096"]]
    seen = _check(chunks2, e.run(chunks2, G), seen)
    assert len(seen) > 80


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 16384, 32767, 32768, 32769, 65536, 65537, 100_001])
def test_emu_cdc_ragged_lengths(n):
    G = ref.gear_table()
    rng = synth.rng_for(3, n)
    chunks = [synth.gen_random(rng, n).tobytes(), synth.gen_text(rng, n).tobytes(), bytes(n)]
    _check(chunks, emulib.EmuCdc().run(chunks, G))


def test_emu_cdc_candidate_overflow_slow_path():
    """A table in which one byte value hashes to 0 makes every position of a run a candidate: tiles overflow
    their candidate slots and the selection kernel must fall back to re-hashing -- results still match a
    CPU evaluation of the same spec with the same (artificial) table."""
    G = ref.gear_table().copy()
    G[0x41] = 0  # 'A' contributes nothing: H == 0 inside long runs of 'A'
    data = (b"A" * 70_000) + synth.gen_random(synth.rng_for(4), 30_000).tobytes() + (b"A" * 5000)
    prefix, seg_end, fps, first, base, cc = emulib.EmuCdc().run([data], G)
    assert (cc == 0xFFFFFFFF).any(), "expected at least one overflowed tile"
    # CPU evaluation with the modified table (python restatement of sko_gear_cdc)
    h, prev, cuts = 0, 0, []
    Gi = [int(x) for x in G]
    for i, b in enumerate(data):
        h = ((h << 1) + Gi[b]) & (2**64 - 1)
        ln = i + 1 - prev
        cut = ln >= ref.CDC_MAX or (ln >= ref.CDC_AVG and (h & ref.CDC_MASK_L) == 0) or (ref.CDC_MIN <= ln < ref.CDC_AVG and (h & ref.CDC_MASK_S) == 0) or i + 1 == len(data)
        if cut:
            cuts.append(i + 1)
            prev = i + 1
    assert seg_end.tolist() == cuts


def test_emu_cdc_without_dedup():
    G = ref.gear_table()
    c = synth.dedup_stream(1 << 18).tobytes()
    prefix, seg_end, fps, first, base, _ = emulib.EmuCdc().run([c], G, dedup=False)
    assert first is None and (seg_end == ref.gear_cdc(c)).all()


def test_emu_dedup_overfull_table_is_not_an_error():
    """a fingerprint table that cannot take more segments must not stop the transfer.  2^10 slots against ~4000 distinct
    segments: every call succeeds, a segment is either placed (first-seen index <= its own) or reported as not seen before."""
    gear = ref.gear_table()
    cdc = emulib.EmuCdc(slots_log2=10)
    rng = synth.rng_for(3, 77)
    seen_total = 0
    for _ in range(4):
        chunks = [synth.gen_random(rng, 4 << 20).tobytes() for _ in range(4)]
        prefix, seg_end, fps, first, base, _ = cdc.run(chunks, gear, dedup=True)
        n = int(prefix[-1])
        idx = np.arange(base, base + n, dtype=np.uint64)
        assert (first[:n] <= idx).all()
        seen_total += n
    assert seen_total > 3 * 1024


def test_emu_literal_streams_and_run_gather_match_a_host_gather():
    """The dedup-on-the-wire device kernels under the emulator: sky_lit_plan + sky_lit_gather (skyhip_dedup_literals) put every chunk's NEW segments back to
    back exactly as numpy does from the same cuts and first-seen indices, across two calls that share the table, with ragged and tiny chunks; and
    sky_gather_runs (skyhip_gather_md5) copies byte runs of any length and alignment to where they belong and nowhere else."""
    from skyplane_amd.gateway import dedup_wire

    gear = ref.gear_table()
    cdc = emulib.EmuCdc(slots_log2=12)
    stream = synth.dedup_stream(6 * 200_000, dup_fraction=0.5, config_id=3)
    chunks = [stream[i * 200_000:(i + 1) * 200_000].tobytes() for i in range(6)]
    chunks[4] = chunks[1]                                    # nothing but references in the second call
    chunks[5] = chunks[5][:70_001]
    chunks.append(b"x" * 13)
    for batch in (chunks[:3], chunks[3:]):
        prefix, seg_end, fps, first, base, _cc, streams = cdc.run(batch, gear, dedup=True, literals=True)
        for i, c in enumerate(batch):
            lens, kinds, _sl = dedup_wire.classify_segments(prefix.astype(np.uint64), seg_end, first, base, i)
            ends = np.cumsum(lens.astype(np.int64))
            want = b"".join(c[e - l:e] for e, l, k in zip(ends, lens, kinds) if k == dedup_wire.KIND_LITERAL)
            assert streams[i] == want, (i, len(streams[i]), len(want))
    assert streams[1] == b""                                  # chunks[4] == chunks[1]: every segment was seen in the first call
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, 300_000, dtype=np.uint8)
    dst = np.full(400_000, 0xEE, np.uint8)
    runs, pos = [], 7
    for n in [1, 2, 15, 16, 17, 63, 64, 65, 255, 4096, 70_001] + [int(x) for x in rng.integers(1, 3000, 40)]:
        o = int(rng.integers(0, src.size - n))
        runs.append((o, pos, n))
        pos += n + int(rng.integers(0, 3))                   # sometimes touching, sometimes a gap that must stay untouched
    emulib.gather_runs([src.ctypes.data + o for o, _, _ in runs], [dst.ctypes.data + d for _, d, _ in runs], [n for _, _, n in runs])
    want = np.full(400_000, 0xEE, np.uint8)
    for o, d, n in runs:
        want[d:d + n] = src[o:o + n]
    assert np.array_equal(dst, want)
