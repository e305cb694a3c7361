"""Dedup on the wire (skyplane_amd/gateway/dedup_wire.py) on a GPU-less box.

The device is the SIMT emulator running the shipping kernel sources (tests/emu): Gear CDC, segment fingerprints and the dedup table are the real
kernels' logic, LZ4 frames and MD5 too.  Covered: the recipe format and its rejection of malformed input, a stream with duplicates across chunks going
source operator -> sidecar payloads -> destination operator byte for byte (and saving bytes), chunks arriving in the wrong order (a reference before
its literal: "not ready", then resolved), epochs (table resets bound what the destination remembers), a reference that never resolves."""
import hashlib
import time
import uuid
from multiprocessing import Event, Queue

import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth
from skyplane_amd.chunk import Chunk, ChunkRequest
from skyplane_amd.gateway import dedup_wire, gateway_program, shm_arena, sidecar
from skyplane_amd.gateway.chunk_store import ChunkStore
from skyplane_amd.gateway.gateway_queue import GatewayQueue
from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress, GatewayHipDecompress
from tests.emu import emulib


class EmuDedupContext:
    """SkyHipContext's call shape over the emulator, with the CDC / dedup side of it (cdc_results, dedup_reset)."""

    def __init__(self, slots_log2=16):
        self.slots_log2 = slots_log2
        self.cdc = emulib.EmuCdc(slots_log2)
        self.last = None
        self.gear = ref.gear_table()
        self.resets = 0

    def process_batch(self, chunks, flags=3, frames_into=None):
        from skyplane_amd.hip_ops import ChunkResult

        raw = [bytes(c) for c in chunks]
        frames, md5s, _ = emulib.process(raw, flags=flags & 3) if flags & 3 else ([None] * len(raw), [None] * len(raw), None)
        if flags & 4:
            prefix, seg_end, fps, first, base, _ = self.cdc.run(raw, self.gear, dedup=bool(flags & 8))
            self.last = (prefix.astype(np.uint64), seg_end, fps, first, base)
        return [ChunkResult(frame=f if flags & 1 else None, md5=m if flags & 2 else None) for f, m in zip(frames, md5s)]

    def cdc_results(self, n, in_len):
        prefix, seg_end, fps, first, base = self.last
        assert len(prefix) == n + 1
        return prefix, seg_end, fps, first, base

    def dedup_reset(self):
        base = self.cdc.seg_base
        self.cdc = emulib.EmuCdc(self.slots_log2)
        self.cdc.seg_base = base
        self.resets += 1

    def decompress_batch(self, frames, raw_lens, want_md5=False, into=None):
        rc, outs, status = emulib.decompress([bytes(f) for f in frames], [int(r) for r in raw_lens])
        if rc != 0:
            raise ValueError(f"frame rejected: {status}")
        return (outs, emulib.process(outs, flags=2)[1]) if want_md5 else outs

    def close(self):
        pass


def _stores(tmp_path, chunks):
    src, dst = ChunkStore(str(tmp_path / "src")), ChunkStore(str(tmp_path / "dst"))
    reqs = []
    for c in chunks:
        cid = uuid.uuid4().hex
        src.get_chunk_file_path(cid).write_bytes(c)
        reqs.append(ChunkRequest(chunk=Chunk(src_key="k", dest_key="k", chunk_id=cid, chunk_length_bytes=len(c), md5_hash=hashlib.md5(c).hexdigest())))
    return src, dst, reqs


def _ops(src, dst, ctx_src, ctx_dst, **kw):
    ee, eq = Event(), Queue()
    comp = GatewayHipCompress("gpu_compress_0", "local:t", GatewayQueue(), GatewayQueue(), ee, eq, src, n_processes=1, max_batch=8, max_chunk_bytes=4 << 20,
                              device_ids=[0], context_factory=lambda d, mc, mb: ctx_src, dedup_wire=True, **kw)
    dec = GatewayHipDecompress("gpu_decompress_0", "local:t", GatewayQueue(), GatewayQueue(), ee, eq, dst, n_processes=1, max_batch=8, max_chunk_bytes=4 << 20,
                               device_ids=[0], context_factory=lambda d, mc, mb: ctx_dst)
    return comp, dec


def _ship(src, dst, reqs):
    """What sender + deferred receiver do: the payload sidecar of every chunk appears on the destination (is_compressed payload, unchanged bytes)."""
    for cr in reqs:
        sidecar.compressed_path(dst, cr.chunk.chunk_id).write_bytes(shm_arena.read_payload(sidecar.compressed_path(src, cr.chunk.chunk_id)))      # (a file or an arena slot)


def _dup_chunks(n=6, size=1 << 20):
    stream = synth.dedup_stream(n * size, dup_fraction=0.5, config_id=3)
    return [stream[i * size:(i + 1) * size].tobytes() for i in range(n)]


def test_recipe_format_roundtrip_and_rejection():
    rng = np.random.default_rng(1)
    lens = np.array([1000, 2000, 1500, 700], np.uint32)
    kinds = np.array([0, 1, 0, 1], np.uint8)
    fps = rng.integers(0, 256, (4, 16), dtype=np.uint8)
    lit = rng.integers(0, 256, 2500, dtype=np.uint8).tobytes()
    frame = ref.lz4f_compress_port(lit)
    blob = dedup_wire.encode_recipe(0xABCDEF0123456789, 7, lens, kinds, fps, frame, 2500)
    assert dedup_wire.is_recipe(blob) and not dedup_wire.is_recipe(frame) and len(blob) == dedup_wire.HEADER_BYTES + 4 * dedup_wire.SEG_BYTES + len(frame)
    r = dedup_wire.parse_recipe(blob, max_raw_len=5200)
    assert (r.lane, r.epoch, r.raw_len, r.lit_raw_len) == (0xABCDEF0123456789, 7, 5200, 2500)
    assert (r.segs["len"] == lens).all() and (r.segs["kind"] == kinds).all() and (r.segs["fp"] == fps).all() and bytes(r.lit_frame) == frame
    for bad in (blob[:20], blob[:-1], blob + b"x", blob[:4] + b"\x09" + blob[5:], b"SKYX" + blob[4:]):
        with pytest.raises(dedup_wire.RecipeError):
            dedup_wire.parse_recipe(bad)
    with pytest.raises(dedup_wire.RecipeError):
        dedup_wire.parse_recipe(blob, max_raw_len=5199)
    tampered = bytearray(blob)
    tampered[dedup_wire.HEADER_BYTES] ^= 1                    # a segment length no longer adds up
    with pytest.raises(dedup_wire.RecipeError):
        dedup_wire.parse_recipe(bytes(tampered))
    tampered = bytearray(blob)
    tampered[dedup_wire.HEADER_BYTES + 4] = 2                 # unknown kind
    with pytest.raises(dedup_wire.RecipeError):
        dedup_wire.parse_recipe(bytes(tampered))


def test_segment_store_keeps_two_epochs_per_lane():
    st = dedup_wire.SegmentStore()
    st.put_many(1, 0, [b"a" * 16], [b"x" * 10])
    st.put_many(1, 1, [b"b" * 16], [b"y" * 20])
    st.put_many(2, 0, [b"a" * 16], [b"z" * 5])
    assert st.get(1, 0, b"a" * 16) == b"x" * 10 and st.get(1, 1, b"a" * 16) is None and st.bytes_held == 35
    st.put_many(1, 2, [b"c" * 16], [b"w"])
    assert st.epochs_held(1) == [1, 2] and st.get(1, 0, b"a" * 16) is None and st.get(2, 0, b"a" * 16) == b"z" * 5 and st.bytes_held == 26


def test_dedup_wire_end_to_end_saves_bytes_and_rebuilds_exactly(tmp_path):
    chunks = _dup_chunks()
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, EmuDedupContext(), EmuDedupContext())
    assert all(comp.process_batch(reqs))
    payloads = [sidecar.compressed_path(src, cr.chunk.chunk_id).read_bytes() for cr in reqs]
    assert all(dedup_wire.is_recipe(p) for p in payloads)
    plain = sum(len(f) for f in emulib.process(chunks, flags=1)[0])
    ref_bytes = sum(m["dedup_reference_bytes"] for m in comp._last_metadata)
    assert ref_bytes > 0.25 * sum(map(len, chunks)), "the 50 %-duplicate stream should leave at least a quarter of its bytes out"
    assert sum(map(len, payloads)) < 0.8 * plain, (sum(map(len, payloads)), plain)
    _ship(src, dst, reqs)
    assert all(dec.process_batch(reqs))
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
        assert not sidecar.compressed_path(dst, cr.chunk.chunk_id).exists()
    assert all(m["md5_hex"] == hashlib.md5(c).hexdigest() and "dedup_reference_bytes" in m for m, c in zip(dec._last_metadata, chunks))


def test_dedup_wire_reference_before_literal_is_not_ready_then_resolves(tmp_path):
    chunks = _dup_chunks()
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, EmuDedupContext(), EmuDedupContext())
    assert all(comp.process_batch(reqs))
    _ship(src, dst, reqs)
    # the destination sees the LAST chunks first: whatever they reference in earlier chunks has not arrived
    late = reqs[3:]
    oks = dec.process_batch(late)
    assert not all(oks), "chunks that reference earlier chunks must wait for them"
    waiting = [cr for cr, ok in zip(late, oks) if not ok]
    assert all(sidecar.compressed_path(dst, cr.chunk.chunk_id).exists() for cr in waiting)       # payload kept for the retry
    assert all(dec.process_batch(reqs[:3]))
    assert all(dec.process_batch(waiting))                                                        # the worker loop's re-queue
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c


def test_dedup_wire_epochs_bound_the_destination_store(tmp_path):
    chunks = _dup_chunks(n=8, size=512 << 10)
    src, dst, reqs = _stores(tmp_path, chunks)
    ctx_src = EmuDedupContext()
    comp, dec = _ops(src, dst, ctx_src, EmuDedupContext(), dedup_epoch_bytes=1 << 20)      # a new epoch every two chunks
    for k in range(0, 8, 2):
        assert all(comp.process_batch(reqs[k:k + 2]))
    assert ctx_src.resets == 4
    recs = [dedup_wire.parse_recipe(sidecar.compressed_path(src, cr.chunk.chunk_id).read_bytes()) for cr in reqs]
    assert [r.epoch for r in recs] == [0, 0, 1, 1, 2, 2, 3, 3] and len({r.lane for r in recs}) == 1
    _ship(src, dst, reqs)
    for k in range(0, 8, 2):
        assert all(dec.process_batch(reqs[k:k + 2]))
    assert dec._segment_store().epochs_held(recs[0].lane) == [2, 3]
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c


def test_dedup_wire_unresolvable_reference_is_an_error_after_the_wait(tmp_path):
    import time

    chunks = _dup_chunks(n=4)
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, EmuDedupContext(), EmuDedupContext())
    assert all(comp.process_batch(reqs))
    _ship(src, dst, reqs)

    def needs_others(cr):
        r = dedup_wire.parse_recipe(sidecar.compressed_path(src, cr.chunk.chunk_id).read_bytes())
        own = {f.tobytes() for f in r.segs["fp"][r.segs["kind"] == dedup_wire.KIND_LITERAL]}
        return any(f.tobytes() not in own for f in r.segs["fp"][r.segs["kind"] == dedup_wire.KIND_REFERENCE])

    lonely = [cr for cr in reqs[1:] if needs_others(cr)]
    assert lonely, "the stream copies spans across chunk boundaries: some chunk must reference an earlier one"
    dec.dedup_wait_s = 0.05
    assert dec.process_batch(lonely[:1]) == [False]                   # first sight starts the clock
    time.sleep(0.1)
    with pytest.raises(ValueError, match="did not arrive"):
        dec.process_batch(lonely[:1])


def test_dedup_wire_corrupt_payload_and_digest_mismatch_are_errors(tmp_path):
    chunks = _dup_chunks(n=2)
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, EmuDedupContext(), EmuDedupContext())
    assert all(comp.process_batch(reqs))
    _ship(src, dst, reqs)
    p = sidecar.compressed_path(dst, reqs[0].chunk.chunk_id)
    blob = bytearray(p.read_bytes())
    p.write_bytes(bytes(blob[:-3]))                                  # truncated recipe
    with pytest.raises(dedup_wire.RecipeError):
        dec.process_batch(reqs[:1])
    good = dedup_wire.parse_recipe(bytes(blob))
    k = int(np.nonzero(good.segs["kind"] == 0)[0][0])
    blob[dedup_wire.HEADER_BYTES + k * dedup_wire.SEG_BYTES + 5] ^= 0xFF        # a literal's fingerprint: harmless for THIS chunk, its bytes are still right
    p.write_bytes(bytes(blob))
    assert dec.process_batch(reqs[:1]) == [True]
    # a wrong expected digest is caught by the whole-chunk check
    bad = ChunkRequest(chunk=Chunk(src_key="k", dest_key="k", chunk_id=reqs[1].chunk.chunk_id, chunk_length_bytes=len(chunks[1]), md5_hash="0" * 32))
    with pytest.raises(ValueError, match="checksum mismatch"):
        dec.process_batch([bad])


def test_program_nodes_carry_dedup_wire():
    d = gateway_program.GatewayGpuCompress(num_workers=2, dedup_wire=True, dedup_epoch_mb=2048).to_dict()
    assert d["dedup_wire"] is True and d["cdc"] is True and d["dedup"] is True and d["dedup_epoch_mb"] == 2048
    d2 = gateway_program.GatewayGpuDecompress(num_workers=4, dedup_wire=True).to_dict()
    assert d2["dedup_wire"] is True and d2["num_workers"] == 1


def test_dedup_wire_through_sockets_and_both_operator_loops():
    """scripts/e2e_steady.py --dedup-wire on the CPU: source operator (pipeline lanes) -> sender (sendfile) -> TCP -> deferred receiver -> destination
    operator, the emulator as the device on both sides; recipes arrive on three connections in whatever order, references wait for their literals through
    the worker loop's re-queue, every destination file is compared with its source and its digest."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    p = subprocess.run([sys.executable, str(root / "scripts" / "e2e_steady.py"), "--context", "emu", "--dedup-wire", "--chunks", "24", "--chunk-kib", "256",
                        "--connections", "3", "--max-batch", "4"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["verified"] is True and res["dedup_wire"] is True
    assert res["wire_ratio"] > 1.45, f"LZ4 alone reaches about 1.3 on this stream; with the duplicates left out: {res['wire_ratio']}"


def test_parse_recipe_never_fails_in_any_other_way():
    """Untrusted input: whatever the bytes, parse_recipe returns a consistent Recipe or raises RecipeError (never an IndexError, a numpy error, ...)."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    rng = np.random.default_rng(5)
    lens = np.array([300, 400, 50], np.uint32)
    good = dedup_wire.encode_recipe(1, 2, lens, [0, 1, 0], rng.integers(0, 256, (3, 16), dtype=np.uint8), ref.lz4f_compress_port(bytes(350)), 350)

    @settings(max_examples=300, deadline=None)
    @given(st.one_of(st.binary(max_size=200), st.tuples(st.integers(0, len(good) - 1), st.integers(0, 255)), st.integers(0, len(good))))
    def run(x):
        if isinstance(x, bytes):
            blob = dedup_wire.MAGIC + x
        elif isinstance(x, tuple):
            b = bytearray(good)
            b[x[0]] = x[1]
            blob = bytes(b)
        else:
            blob = good[:x]
        try:
            r = dedup_wire.parse_recipe(blob)
        except dedup_wire.RecipeError:
            return
        assert int(r.segs["len"].sum()) == r.raw_len and len(r.lit_frame) + dedup_wire.HEADER_BYTES + dedup_wire.SEG_BYTES * len(r.segs) == len(blob)

    run()


class ArenaEmuDedupContext(EmuDedupContext):
    """+ SkyHipContext's staging interface over ordinary memory (pinned_buffer / release_pinned / frame_bound, frames_into, into): the operators then take
    the same zero-copy branches as with the real library -- payloads read into one arena, decoded chunks returned as views of another."""

    def pinned_buffer(self, nbytes):
        return np.full(nbytes, 0xCD, np.uint8)

    def release_pinned(self, buf):
        pass

    def frame_bound(self, n):
        return emulib.frame_bound(n)

    def process_batch(self, chunks, flags=3, frames_into=None):
        res = super().process_batch(chunks, flags=flags)
        if frames_into is not None and flags & 1:
            for r, v in zip(res, frames_into):
                v[:len(r.frame)] = np.frombuffer(r.frame, np.uint8)
                r.frame = v[:len(r.frame)]
        return res

    def decompress_batch(self, frames, raw_lens, want_md5=False, into=None):
        res = super().decompress_batch(frames, raw_lens, want_md5=want_md5)
        outs, digs = res if want_md5 else (res, None)
        if into is not None:
            assert len(into) == len(outs) and all(v.size >= len(o) for v, o in zip(into, outs))
            views = []
            for v, o in zip(into, outs):
                v[:len(o)] = np.frombuffer(o, np.uint8)
                views.append(v[:len(o)])
            outs = views
        return (outs, digs) if want_md5 else outs


def test_staging_branches_with_plain_frames_recipes_and_both_in_one_batch(tmp_path):
    """The arena (pinned) branches of both operators, as the real library makes them take: a batch of plain frames, a batch of recipes, and a batch that
    holds both kinds (a deduplicating and a plain source feeding one destination)."""
    chunks = _dup_chunks(n=6, size=512 << 10)
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, ArenaEmuDedupContext(), ArenaEmuDedupContext())
    plain = GatewayHipCompress("gpu_compress_1", "local:t", GatewayQueue(), GatewayQueue(), Event(), Queue(), src, n_processes=1, max_batch=8,
                               max_chunk_bytes=4 << 20, device_ids=[0], context_factory=lambda d, mc, mb: ArenaEmuDedupContext())
    assert all(plain.process_batch(reqs[:2]))            # two chunks as plain frames ...
    assert all(comp.process_batch(reqs[2:]))             # ... four as recipes
    kinds = [dedup_wire.is_recipe(shm_arena.read_payload(sidecar.compressed_path(src, cr.chunk.chunk_id))) for cr in reqs]
    assert kinds == [False, False, True, True, True, True]
    _ship(src, dst, reqs)
    assert all(dec.process_batch(reqs[:1]))              # plain only
    assert all(dec.process_batch(reqs[1:4]))             # plain + recipes
    assert all(dec.process_batch(reqs[4:]))              # recipes only
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
    assert all("md5_hex" in m for m in dec._last_metadata)


class _HostBuf:
    """hip_ops.DeviceBuffer's surface over host memory: the "device address" is the array's address."""

    def __init__(self, arr):
        self.arr = arr

    @property
    def dptr(self):
        return self.arr.ctypes.data

    def __len__(self):
        return self.arr.size


class DeviceStoreEmuContext(ArenaEmuDedupContext):
    """+ the calls that let gpu_decompress put a recipe's chunk together ON THE DEVICE (skyhip_decompress_to_device / skyhip_gather_md5), here over host
    memory: literal streams stay where they were decoded, the segment store holds buffers with an address, chunks are gathered from byte runs."""

    def __init__(self):
        super().__init__()
        self.gathers = self.chains = self.segment_calls = 0

    def decompress_to_device(self, frames, raw_lens):
        return [_HostBuf(np.frombuffer(o, np.uint8).copy()) for o in ArenaEmuDedupContext.decompress_batch(self, frames, raw_lens)]

    def gather_md5(self, run_src, run_len, into, want_md5=True):
        import ctypes

        outs, digs = [], []
        for src, ln, dst in zip(run_src, run_len, into):
            blob = b"".join(ctypes.string_at(int(a), int(n)) for a, n in zip(src, ln))
            dst[:len(blob)] = np.frombuffer(blob, np.uint8)
            outs.append(dst[:len(blob)])
            digs.append(hashlib.md5(blob).digest())
        self.gathers += 1
        self.chains += len(into) if want_md5 else 0
        return outs, (digs if want_md5 else None)

    def segment_md5_device(self, addrs, lens):
        import ctypes

        self.segment_calls += 1
        return np.frombuffer(b"".join(hashlib.md5(ctypes.string_at(int(a), int(n))).digest() for a, n in zip(addrs, lens)), np.uint8).reshape(-1, 16).copy()


def test_recipes_are_put_together_from_device_resident_runs(tmp_path):
    """The destination's device path (round 5): literal streams decoded to "device" memory and kept there by the segment store, chunks gathered from runs
    of this and earlier streams, plain frames in the same batch decoded as before, a reference before its literal waits (its stream stays put), and the
    decoded chunks land in slot files published as hard links."""
    chunks = _dup_chunks(n=6, size=512 << 10)
    src, dst, reqs = _stores(tmp_path, chunks)
    dctx = DeviceStoreEmuContext()
    comp, dec = _ops(src, dst, ArenaEmuDedupContext(), dctx)
    plain = GatewayHipCompress("gpu_compress_1", "local:t", GatewayQueue(), GatewayQueue(), Event(), Queue(), src, n_processes=1, max_batch=8,
                               max_chunk_bytes=4 << 20, device_ids=[0], context_factory=lambda d, mc, mb: ArenaEmuDedupContext())
    assert all(plain.process_batch(reqs[:1]))            # one chunk as a plain frame ...
    assert all(comp.process_batch(reqs[1:]))             # ... five as recipes
    _ship(src, dst, reqs)
    oks = dec.process_batch(reqs[4:])                    # the last two first: what they reference in chunks 1-3 has not arrived
    assert not all(oks)
    waiting = [cr for cr, ok in zip(reqs[4:], oks) if not ok]
    assert all(dec.process_batch(reqs[:4]))              # plain + recipes in one batch
    assert all(dec.process_batch(waiting))
    assert dctx.gathers >= 2
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
    store = dec._segment_store()
    assert isinstance(store, dedup_wire.DeviceSegmentStore) and store._maps       # fingerprint -> address, in csrc/skyhost.c's map
    assert all(isinstance(b, _HostBuf) for m in store._maps.values() for b in m[1]) and store.bytes_held > 0      # ... and the buffers the addresses point into
    slot_inodes = {f.stat().st_ino for f in (tmp_path / "dst").glob("_outslot_*")}
    assert slot_inodes and any(dst.get_chunk_file_path(cr.chunk.chunk_id).stat().st_ino in slot_inodes for cr in reqs)
    dec.worker_exit(0)


def test_recipes_are_verified_segment_by_segment_not_by_a_chain(tmp_path):
    """dedup_verify="segments" (the default, round 6): a batch of recipes costs ONE call that digests its newly arrived literal segments and no whole-chunk
    chain; a literal stream that does not match its recipe's fingerprints is a checksum mismatch; dedup_verify="chunk" is round 5's rule."""
    chunks = _dup_chunks(n=5, size=512 << 10)
    for mode in ("segments", "chunk"):
        src, dst, reqs = _stores(tmp_path / mode, chunks)
        for cr, c in zip(reqs, chunks):
            cr.chunk.md5_hash = hashlib.md5(c).hexdigest()
        dctx = DeviceStoreEmuContext()
        comp, dec = _ops(src, dst, ArenaEmuDedupContext(), dctx)
        dec.dedup_verify = mode
        assert all(comp.process_batch(reqs))
        _ship(src, dst, reqs)
        assert all(dec.process_batch(reqs))
        if mode == "segments":
            assert dctx.segment_calls == 1 and dctx.chains == 0
            assert all(m.get("verified") == "segment fingerprints" for m in dec._last_metadata)
        else:
            assert dctx.segment_calls == 0 and dctx.chains == len(reqs)
        for cr, c in zip(reqs, chunks):
            assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
        dec.worker_exit(0)
    # one literal byte wrong on the wire: the frame still decodes (LZ4 frames carry no checksum here), the segment's fingerprint does not match
    src, dst, reqs = _stores(tmp_path / "bad", chunks)
    dctx = DeviceStoreEmuContext()
    comp, dec = _ops(src, dst, ArenaEmuDedupContext(), dctx)
    assert all(comp.process_batch(reqs))
    _ship(src, dst, reqs)
    real = dctx.decompress_to_device

    def flipped(frames, raw_lens):
        bufs = real(frames, raw_lens)
        bufs[0].arr[len(bufs[0]) // 2] ^= 0x40
        return bufs

    dctx.decompress_to_device = flipped
    with pytest.raises(ValueError, match="checksum mismatch, literal segment"):
        dec.process_batch(reqs)
    dec.worker_exit(0)


def test_recipe_with_a_zero_length_segment_is_refused():
    segs = np.zeros(2, dedup_wire.SEG_DTYPE)
    segs["len"] = [0, 10]
    head = dedup_wire._HDR.pack(dedup_wire.MAGIC, dedup_wire.VERSION, 1, 0, 2, 10, 0, 0) + segs.tobytes()
    segs["kind"] = [0, 1]
    with pytest.raises(dedup_wire.RecipeError, match="length zero"):
        dedup_wire.parse_recipe(dedup_wire._HDR.pack(dedup_wire.MAGIC, dedup_wire.VERSION, 1, 0, 2, 10, 0, 0) + segs.tobytes())
    assert head


def test_file_segment_store_is_shared_between_processes(tmp_path):
    import multiprocessing as mp

    st = dedup_wire.FileSegmentStore(tmp_path / "seg")
    st.put_chunk(7, 0, [b"a" * 16, b"b" * 16], [0, 3], [3, 4], b"xyz1234")

    def child(q):
        other = dedup_wire.FileSegmentStore(tmp_path / "seg")
        q.put((other.get(7, 0, b"b" * 16), other.get(7, 0, b"c" * 16)))
        other.put_chunk(7, 0, [b"c" * 16], [0], [2], b"CC")

    q = mp.get_context("fork").Queue()
    p = mp.get_context("fork").Process(target=child, args=(q,))
    p.start()
    assert q.get(timeout=30) == (b"1234", None)
    p.join(30)
    assert st.get(7, 0, b"c" * 16) == b"CC"                 # appended by the other process, found by reading the index on
    hit = st.get_many(7, 0, [b"a" * 16, b"b" * 16])
    assert hit[0][0] is hit[1][0] and (hit[0][1], hit[0][2], hit[1][1], hit[1][2]) == (0, 3, 3, 4)      # one mapping per literal stream: runs stay runs
    st.put_chunk(7, 2, [b"d" * 16], [0], [1], b"D")
    assert st.epochs_held(7) == [2] and not list((tmp_path / "seg").glob("L*-0-*"))                      # epoch 0 retired, its files gone


def test_dedup_wire_two_destination_worker_processes_share_the_file_store(tmp_path):
    """gpu_decompress(dedup_store="files", n_processes=2): whichever worker gets a chunk finds the segments the other one stored."""
    import queue as pyqueue
    import time

    chunks = _dup_chunks(n=8, size=256 << 10)
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, _ = _ops(src, dst, EmuDedupContext(), EmuDedupContext())
    assert all(comp.process_batch(reqs))
    _ship(src, dst, reqs)
    ee, eq = Event(), Queue()
    q_in, q_out = GatewayQueue(), GatewayQueue()
    dst.add_partition("0", q_in)
    dec = GatewayHipDecompress("gpu_decompress_0", "local:t", q_in, q_out, ee, eq, dst, n_processes=2, max_batch=2, max_chunk_bytes=4 << 20, device_ids=[0],
                               pipeline_depth=1, dedup_store="files", context_factory=lambda d, mc, mb: EmuDedupContext())
    for cr in reversed(reqs):
        q_in.put(cr)
    dec.start_workers()
    done, t0 = [], time.time()
    while len(done) < len(reqs) and time.time() - t0 < 120 and not ee.is_set():
        try:
            done.append(q_out.get_nowait())
        except pyqueue.Empty:
            time.sleep(0.01)
    dec.stop_workers()
    assert not ee.is_set(), eq.get() if not eq.empty() else ""
    assert len(done) == len(reqs)
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
    d = gateway_program.GatewayGpuDecompress(num_workers=4, dedup_wire=True, dedup_store="files").to_dict()
    assert d["num_workers"] == 4 and d["dedup_store"] == "files"


def test_recipe_format_is_pinned_by_a_golden_vector():
    """tests/golden/recipe_v1.bin (tests/golden/make_recipe_golden.py): the encoder still produces it byte for byte and the parser reads it back."""
    from pathlib import Path

    from tests.golden import make_recipe_golden

    want = (Path(__file__).parent / "golden" / "recipe_v1.bin").read_bytes()
    blob, lit = make_recipe_golden.build()
    assert blob == want
    r = dedup_wire.parse_recipe(want)
    assert (r.lane, r.epoch, r.raw_len, r.lit_raw_len, len(r.segs)) == (0x0123456789ABCDEF, 3, 25052, 18908, 5)
    assert r.segs["kind"].tolist() == [0, 1, 0, 0, 1] and ref.lz4f_decode(bytes(r.lit_frame), r.lit_raw_len)[0] == lit
    assert want[:5] == b"SKYD\x01" and want[5:13] == bytes.fromhex("efcdab8967452301")


def test_dedup_wire_ragged_and_empty_chunks(tmp_path):
    """Chunks below the smallest segment, an empty one, one that is a single run, one made of nothing but copies of an earlier one."""
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes()
    chunks = [b"", b"tiny", base, bytes(50_000), base, base[:1023], base[5:40_000] + base[:3]]
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, EmuDedupContext(), EmuDedupContext())
    assert all(comp.process_batch(reqs))
    recs = [dedup_wire.parse_recipe(sidecar.compressed_path(src, cr.chunk.chunk_id).read_bytes()) for cr in reqs]
    assert len(recs[0].segs) == 0 and recs[0].raw_len == 0
    assert recs[4].lit_raw_len == 0 and recs[4].segs["kind"].all(), "the second copy of a chunk is references only"
    _ship(src, dst, reqs)
    assert all(dec.process_batch(reqs))
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c


# ---- round 3: the advisor's findings on the destination store ----
def test_waiting_recipe_stores_its_literals_once_and_is_not_decoded_again(tmp_path):
    """A chunk that waits for a reference comes back many times: its literal stream must be stored ONCE (dedup_store="files": one .lit file, not one
    per retry) and decoded once (the decoded literals wait with it)."""
    chunks = _dup_chunks()
    src, dst, reqs = _stores(tmp_path, chunks)
    ctx_dst = EmuDedupContext()
    calls = {"n": 0}
    real = ctx_dst.decompress_batch

    def counting(frames, raw_lens, want_md5=False, into=None):
        calls["n"] += len(frames)
        return real(frames, raw_lens, want_md5=want_md5, into=into)

    ctx_dst.decompress_batch = counting
    ee, eq = Event(), Queue()
    comp = GatewayHipCompress("gpu_compress_0", "local:t", GatewayQueue(), GatewayQueue(), ee, eq, src, n_processes=1, max_batch=8, max_chunk_bytes=4 << 20,
                              device_ids=[0], context_factory=lambda d, mc, mb: EmuDedupContext(), dedup_wire=True)
    dec = GatewayHipDecompress("gpu_decompress_0", "local:t", GatewayQueue(), GatewayQueue(), ee, eq, dst, n_processes=1, max_batch=8, max_chunk_bytes=4 << 20,
                               device_ids=[0], context_factory=lambda d, mc, mb: ctx_dst, dedup_store="files")
    assert comp.pipeline_depth == 1, "one lane (one fingerprint table) per deduplicating worker unless asked otherwise"
    assert all(comp.process_batch(reqs))
    _ship(src, dst, reqs)
    late = [cr for cr in reqs[3:]]
    oks = dec.process_batch(late)
    waiting = [cr for cr, ok in zip(late, oks) if not ok]
    assert waiting
    seg_dir = dst.get_chunk_file_path("x").parent / "_segments"
    n_lit, n_dec = len(list(seg_dir.glob("*.lit"))), calls["n"]
    for _ in range(5):                                                     # the worker loop's retries
        assert not any(dec.process_batch(waiting))
    assert len(list(seg_dir.glob("*.lit"))) == n_lit and calls["n"] == n_dec
    assert all(dec.process_batch(reqs[:3])) and all(dec.process_batch(waiting))
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
    dec.process_exit(0)
    assert not seg_dir.exists(), "the segment store leaves with the worker"


def test_segment_stores_are_bounded_and_distrust_epochs(tmp_path):
    for store in (dedup_wire.SegmentStore(max_bytes=3000, idle_s=0, live_grace_s=0), dedup_wire.FileSegmentStore(tmp_path / "seg", max_bytes=3000, idle_s=0, live_grace_s=0)):
        fp = lambda k: bytes([k]) * 16      # noqa: E731
        for lane in (1, 2, 3, 4):           # 1000 bytes per lane, lanes long silent (grace 0): the fourth pushes the least recently used lane out
            store.put_chunk(lane, 0, [fp(lane)], [0], [1000], bytes([lane]) * 1000)
            if lane == 2:
                assert store.get(1, 0, fp(1)) is not None       # touch lane 1: lane 2 is the oldest from now on
        with pytest.raises(dedup_wire.StoreEvicted):            # ... and a reference to what the budget took fails at once, it does not wait
            store.get(2, 0, fp(2))
        assert store.get(1, 0, fp(1)) == b"\x01" * 1000 and store.get(4, 0, fp(4)) is not None
        # the mark is not sticky.  The silent lane resumes in the same epoch and sends its segments again: references to what has arrived
        # SINCE the eviction resolve, only a reference to what is really gone fails
        store.put_chunk(2, 0, [fp(22)], [0], [10], b"z" * 10)
        assert store.get(2, 0, fp(22)) == b"z" * 10
        with pytest.raises(dedup_wire.StoreEvicted):
            store.get_many(2, 0, [fp(22), fp(2)])
        # a known lane that claims an epoch far ahead is refused (it would retire everything the lane holds); a small step is normal
        with pytest.raises(dedup_wire.RecipeError):
            store.put_chunk(1, 0xFFFFFFFF, [fp(9)], [0], [10], b"x" * 10)
        assert store.get(1, 0, fp(1)) is not None
        store.put_chunk(1, 2, [fp(9)], [0], [10], b"y" * 10)
        assert store.epochs_held(1) == [2] and store.get(1, 0, fp(1)) is None
        store.cleanup()
    # idle groups go when something else is written
    s = dedup_wire.SegmentStore(idle_s=0.05)
    s.put_chunk(7, 0, [b"a" * 16], [0], [4], b"abcd")
    time.sleep(0.12)
    s.put_chunk(8, 0, [b"b" * 16], [0], [4], b"efgh")
    assert s.get(7, 0, b"a" * 16) is None and s.get(8, 0, b"b" * 16) == b"efgh"
    # a literal stream retired by another worker between the index look-up and the open is a miss, not an exception
    f = dedup_wire.FileSegmentStore(tmp_path / "seg2")
    f.put_chunk(5, 0, [b"c" * 16], [0], [4], b"ijkl")
    for p in (tmp_path / "seg2").glob("*.lit"):
        p.unlink()
    assert f.get_many(5, 0, [b"c" * 16]) == [None]


def test_live_epochs_survive_the_byte_budget_and_workers_share_a_lanes_epoch(tmp_path):
    """ADVICE r3.  (a) The byte budget never evicts what a sender may still reference: groups within keep_epochs of the top of a lane that is in use.
    (b) Several destination workers share one FileSegmentStore directory: a worker that saw none of a lane's chunks for a few epochs must not take the
    next one for a forged jump."""
    fp = lambda k: bytes([k]) * 16      # noqa: E731
    for store in (dedup_wire.SegmentStore(max_bytes=3000), dedup_wire.FileSegmentStore(tmp_path / "live", max_bytes=3000)):
        for lane in (1, 2, 3, 4):
            store.put_chunk(lane, 0, [fp(lane)], [0], [1000], bytes([lane]) * 1000)
        assert all(store.get(lane, 0, fp(lane)) == bytes([lane]) * 1000 for lane in (1, 2, 3, 4)) and store.over_budget_live
        store.put_chunk(1, 1, [fp(5)], [0], [1000], b"e" * 1000)                       # epoch 0 of lane 1 is still live (keep_epochs = 2)
        assert store.get(1, 0, fp(1)) is not None and store.get(1, 1, fp(5)) is not None
        store.put_chunk(1, 2, [fp(6)], [0], [1000], b"f" * 1000)                       # ... and retired when the lane reaches epoch 2
        assert store.get(1, 0, fp(1)) is None and store.epochs_held(1) == [1, 2]
        store.cleanup()
    a, b = dedup_wire.FileSegmentStore(tmp_path / "shared"), dedup_wire.FileSegmentStore(tmp_path / "shared")
    a.put_chunk(7, 0, [fp(1)], [0], [4], b"aaaa")
    b.put_chunk(7, 1, [fp(2)], [0], [4], b"bbbb")
    b.put_chunk(7, 2, [fp(3)], [0], [4], b"cccc")
    a.put_chunk(7, 3, [fp(4)], [0], [4], b"dddd")                                       # used to raise: "lane 0x7 jumps from epoch 0 to 3"
    assert a.get(7, 3, fp(4)) == b"dddd" and b.get(7, 3, fp(4)) == b"dddd" and a.get(7, 2, fp(3)) == b"cccc"
    with pytest.raises(dedup_wire.RecipeError):
        a.put_chunk(7, 40, [fp(9)], [0], [4], b"eeee")                                  # a real jump is still refused ...
    assert not list((tmp_path / "shared").glob("L*-40-*"))                              # ... and leaves no orphan stream file behind


def test_not_ready_chunks_wait_with_backoff_inside_the_lane(tmp_path):
    src, dst, reqs = _stores(tmp_path, [b"x" * 2000, b"y" * 2000])
    ee, eq = Event(), Queue()
    dec = GatewayHipDecompress("gpu_decompress_0", "local:t", GatewayQueue(), GatewayQueue(), ee, eq, dst, n_processes=1, max_batch=8, max_chunk_bytes=4 << 20,
                               device_ids=[0], context_factory=lambda d, mc, mb: EmuDedupContext())
    dec.worker_id = 0
    t0 = time.monotonic()
    for _ in range(4):
        dec._park(reqs[0])
        (due, n, cr), = dec._parked()
        dec._parked().clear()
    assert n == 3 and 0.07 <= due - t0 <= 0.2                              # 10, 20, 40, 80 ms
    dec._park(reqs[1])
    assert dec._take_batch() == [] and len(dec._parked()) == 1             # not due yet: stays parked, the loop does not spin on it
    time.sleep(0.02)
    assert [c.chunk.chunk_id for c in dec._take_batch()] == [reqs[1].chunk.chunk_id] and not dec._parked()


def test_native_fingerprint_map_and_device_store_bounds():
    """csrc/skyhost.c through DeviceSegmentStore: whole-array put / get, first value wins, the map grows, misses are counted; a lane moving on drops the
    old epoch's map together with the buffers it kept alive (= the device memory they own), like SegmentStore drops its dictionaries."""
    import gc
    import weakref

    class Buf:
        pass

    rng = np.random.default_rng(5)
    st = dedup_wire.DeviceSegmentStore(keep_epochs=2, max_bytes=1 << 40)
    n = 50_000                                               # (more than the map's first 16384 slots: it grows twice)
    fps = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    addrs = rng.integers(1, 1 << 48, n, dtype=np.uint64)
    lens = rng.integers(1, 16385, n, dtype=np.uint32)
    b0 = Buf()
    st.put_arrays(7, 0, fps, addrs, lens, b0)
    assert st.bytes_held == int(lens.sum())
    a, l, miss, keep = st.get_arrays(7, 0, fps[::3])
    assert len(keep) == 1
    assert miss == 0 and (a == addrs[::3]).all() and (l == lens[::3]).all()
    st.put_arrays(7, 0, fps[:10], addrs[:10] + np.uint64(5), lens[:10], Buf())      # the same fingerprints again: the first value stays
    a, l, miss, _ = st.get_arrays(7, 0, fps[:10])
    assert (a == addrs[:10]).all() and st.bytes_held == int(lens.sum())
    other = rng.integers(0, 256, (5, 16), dtype=np.uint8)
    a, l, miss, _ = st.get_arrays(7, 0, np.concatenate([other, fps[:2]]))
    assert miss == 5 and (a[:5] == 0).all() and (a[5:] == addrs[:2]).all()
    assert st.get_arrays(7, 1, fps[:4])[2] == 4 and st.get_arrays(8, 0, fps[:4])[2] == 4      # another epoch / lane: nothing
    ref0 = weakref.ref(b0)
    del b0, keep, _                                          # (a look-up's `keep` is what holds a group's buffers while the device reads them)
    st.put_arrays(7, 1, fps[:1], addrs[:1], lens[:1], Buf())
    assert st.epochs_held(7) == [0, 1] and ref0() is not None
    st.put_arrays(7, 2, fps[:1], addrs[:1], lens[:1], Buf())                          # epoch 0 is older than keep_epochs now
    gc.collect()
    assert st.epochs_held(7) == [1, 2] and ref0() is None and st.get_arrays(7, 0, fps[:4])[2] == 4
    st.cleanup()
    assert st.bytes_held == 0 and not st._maps
    # what a group PINS is what the budget counts: buffers cut from one device block hold the whole block, once per group
    class Block:
        nbytes = 1 << 20

    blk = Block()
    cut1, cut2 = Buf(), Buf()
    cut1.block = cut2.block = blk
    st.put_arrays(9, 0, fps[:3], addrs[:3], lens[:3], cut1)
    st.put_arrays(9, 0, fps[3:6], addrs[3:6], lens[3:6], cut2)
    assert st.bytes_held == 1 << 20
    st.put_arrays(9, 1, fps[6:7], addrs[6:7], lens[6:7], cut2)       # another group keeps a piece of the same block: it pins it too
    assert st.bytes_held == 2 << 20
    # an epoch as far ahead as the store keeps epochs is a reordered batch, not an attack (keep_epochs=2 -> jump of 2 is fine, 3 is refused)
    st.put_arrays(9, 3, fps[7:8], addrs[7:8], lens[7:8], Buf())
    with pytest.raises(dedup_wire.RecipeError, match="jumps"):
        st.put_arrays(9, 7, fps[8:9], addrs[8:9], lens[8:9], Buf())
    assert st.drop_not_live() == 0 and dedup_wire.DeviceSegmentStore()._b.max_epoch_jump == 4
    st.cleanup()


class DeviceSourceEmuContext(ArenaEmuDedupContext):
    """+ skyhip_dedup_literals' call shape on the host: the literal streams of the call just made, compressed into the caller's views."""

    def process_batch(self, chunks, flags=3, frames_into=None):
        if flags & 4:
            self._raw = [bytes(c) for c in chunks]
        return super().process_batch(chunks, flags=flags, frames_into=frames_into)

    def dedup_literals(self, in_lens, frames_into):
        prefix, cuts, fps, first, base = self.last
        lit_lens, frames = [], []
        for i, (raw, view) in enumerate(zip(self._raw, frames_into)):
            lens, kinds, _sl = dedup_wire.classify_segments(prefix, cuts, first, base, i)
            ends = np.cumsum(lens.astype(np.int64))
            lit = b"".join(raw[e - l:e] for e, l, k in zip(ends, lens, kinds) if k == dedup_wire.KIND_LITERAL)
            lit_lens.append(len(lit))
            if len(lit) in (0, len(raw)):
                frames.append(None)
                continue
            f = np.frombuffer(emulib.process([lit], flags=1)[0][0], np.uint8)
            view[:f.size] = f
            frames.append(view[:f.size])
        return lit_lens, frames


def test_source_lane_publishes_a_batch_while_the_next_one_is_on_the_device(tmp_path):
    """Round 5, source side of the dedup path: literal streams from the device call (dedup_literals), recipes written in two pieces (RecipeParts), and the
    lane's helper thread publishing batch k -- payload files, side-cars, completion records, output queue -- while batch k + 1 is read and launched; the
    staging areas alternate between two sets.  What arrives at the destination is byte for byte what the synchronous path produces."""
    chunks = _dup_chunks(n=9, size=256 << 10)
    src, dst, reqs = _stores(tmp_path, chunks)
    comp, dec = _ops(src, dst, DeviceSourceEmuContext(), ArenaEmuDedupContext())
    comp.worker_id = 0
    q_out = comp.output_queue
    for k in (0, 3, 6):                                   # three batches: the third one waits for the first to have left its staging set
        assert comp._process_in_lane(reqs[k:k + 3], 0) is None
    for fut in list(comp._tls.finishing.values()):
        fut.result(timeout=30)
    got = [q_out.q.get(timeout=10).chunk.chunk_id for _ in reqs]      # (a multiprocessing queue: what a thread put may take a moment to become visible)
    assert got == [cr.chunk.chunk_id for cr in reqs]          # completed in order, by the helper
    assert {"out0", "out1", "lit0", "lit1"} <= set(comp._arenas)
    payloads = [sidecar.compressed_path(src, cr.chunk.chunk_id).read_bytes() for cr in reqs]
    assert all(dedup_wire.is_recipe(p) for p in payloads)
    # the same stream through the synchronous path of a fresh operator: identical recipes
    src2, _dst2, reqs2 = _stores(tmp_path / "sync", chunks)
    comp2, _ = _ops(src2, _dst2, ArenaEmuDedupContext(), ArenaEmuDedupContext())
    comp2._tls.dedup_state = dict(comp._tls.dedup_state, epoch=0, bytes=0)        # same lane id in the recipe headers
    for k in (0, 3, 6):
        assert all(comp2.process_batch(reqs2[k:k + 3]))
    assert payloads == [sidecar.compressed_path(src2, cr.chunk.chunk_id).read_bytes() for cr in reqs2]
    _ship(src, dst, reqs)
    assert all(dec.process_batch(reqs))
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c
    import queue as pyqueue

    done = 0
    try:
        while done < 9:
            done += src.chunk_status_queue.get(timeout=10)["state"] == "complete"
    except pyqueue.Empty:
        pass
    assert done == 9
    comp.worker_exit(0)
