"""Parity tests proper: libskyhip.so (HIP, gfx950) through its C ABI vs the CPU oracle.  Run with -m gpu.

Bar: for every input (1) the reference's decode side -- liblz4 LZ4F_decompress, what
lz4.frame.decompress is at gateway_receiver.py:196 -- returns the raw bytes and consumes the whole frame,
(2) md5 == hashlib.md5(raw).digest(); both bit-exact.
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import ref  # noqa: E402
from skyplane_amd import synth  # noqa: E402
from tests.model import lz4smodel  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    torch.cuda.init()
    from skyplane_amd import hip_ops

    c = hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=8 << 20, max_batch=4)
    yield c
    c.close()


def _check(chunks, results):
    for i, (d, r) in enumerate(zip(chunks, results)):
        d = bytes(d)
        assert r.md5 == hashlib.md5(d).digest(), f"md5 mismatch chunk {i} len {len(d)}"
        assert ref.lz4f_decompress(r.frame, len(d)) == d, f"liblz4 decode mismatch chunk {i}"
        dec, info = ref.lz4f_decode(r.frame, len(d), strict=True)
        assert dec == d and info["flg"] == 0x68 and info["bd"] == 0x40
        lz4smodel.check_frame(d, r.frame)     # byte-identical to the sequential model of the parse (tests/model/lz4s_model.c)


def test_wave_primitives_selftest(ctx):
    assert ctx.selftest() == 0


def test_small_cases(ctx, small_cases, golden):
    chunks = list(small_cases.values())
    res = ctx.process_batch(chunks)          # 16 chunks with max_batch=4 -> exercises sub-batching
    _check(chunks, res)
    for (name, d), r in zip(small_cases.items(), res):
        assert r.md5.hex() == golden["cases"][name]["md5"]


# (55 / 56 / 57 / 119 / 120 / 121 and 8 MiB - 9 = 55 mod 64: the RFC 1321 padding edges -- the 0x80 byte and the 8-byte length fit the last block, need one more,
# or fill it exactly -- reached through the WHOLE-CHUNK digest kernel's own tail dispatch; sky_segment_md5 sees every residue elsewhere)
@pytest.mark.parametrize("n", [1, 12, 13, 55, 56, 57, 63, 64, 65, 119, 120, 121, 4095, 65535, 65536, 65537, 131073, 1 << 20, (1 << 20) + 77, (8 << 20) - 9])
def test_ragged_lengths(ctx, n):
    rng = synth.rng_for(0, n)
    chunks = [gen(rng, n).tobytes() for gen in (synth.gen_text, synth.gen_sparse, synth.gen_random, synth.gen_records)]
    _check(chunks, ctx.process_batch(chunks))


def test_every_class_ratio_close_to_reference(ctx):
    for name in synth.CLASSES:
        d = synth.gen_class(name, 4 << 20, synth.rng_for(9)).tobytes()
        (r,) = ctx.process_batch([d])
        _check([d], [r])
        assert len(r.frame) <= 1.10 * len(ref.lz4f_compress(d)) + 64, name      # vs the reference's default (block-linked) frames


def test_stream_ratio_within_3_percent_of_reference(ctx):
    """Egress bytes are what a Skyplane user pays for: on the bench's Silesia-like stream the frames must stay within 3 % of
    what the reference's own call -- lz4.frame.compress with python-lz4's defaults (block-linked) -- produces."""
    d = synth.silesia_like(32 << 20, config_id=2)
    chunks = [d[i:i + synth.CHUNK_BYTES].tobytes() for i in range(0, d.size, synth.CHUNK_BYTES)]
    res = ctx.process_batch(chunks)
    ours = sum(len(r.frame) for r in res)
    theirs = sum(len(ref.lz4f_compress(c)) for c in chunks)
    assert ours <= 1.03 * theirs, (ours, theirs)


def test_full_chunk_golden(ctx, golden):
    d = synth.silesia_like(synth.CHUNK_BYTES, config_id=2).tobytes()
    (r,) = ctx.process_batch([d])
    _check([d], [r])
    assert r.md5.hex() == golden["chunk_8MiB_silesia_like"]["md5"]


def test_long_matches_and_overlap(ctx):
    pats = [bytes(8 << 20), b"\x01" * 70_000 + b"\x02" * 70_000, b"0123456789abcdef" * 9000, b"ab" * 40_000 + bytes(5) + b"ab" * 30_000,
            synth.gen_random(synth.rng_for(0, 1), 1000).tobytes() * 150]
    _check(pats, ctx.process_batch(pats))


def test_emit_corner_cases(ctx):
    """Literal runs / match lengths around every extension-byte boundary, runs longer than 64 (copied by the whole workgroup), neighbouring lanes
    meeting inside an image dword: the same blocks the emulator runs (tests/test_emu_kernels.py)."""
    from tests.test_emu_kernels import _emit_corner_cases

    cases = _emit_corner_cases()
    _check(cases, ctx.process_batch(cases))


def test_md5_only_and_lz4_only(ctx, small_cases):
    from skyplane_amd import hip_ops

    chunks = [small_cases["mixed_200k"], small_cases["abc_run"]]
    r1 = ctx.process_batch(chunks, flags=hip_ops.F_MD5)
    r2 = ctx.process_batch(chunks, flags=hip_ops.F_LZ4)
    for d, a, b in zip(chunks, r1, r2):
        assert a.frame is None and a.md5 == hashlib.md5(d).digest()
        assert b.md5 is None and ref.lz4f_decompress(b.frame, len(d)) == d


def test_error_codes(ctx):
    from skyplane_amd import hip_ops

    with pytest.raises(hip_ops.SkyHipError) as e:
        ctx.process_batch([bytes((8 << 20) + 1)])
    assert e.value.code == -4


def test_device_resident_batch_matches_oracle(ctx):
    """Kernel-only path: 48 x 8 MiB chunks resident in HBM (tiled + rotated unit), every digest and every frame checked."""
    from skyplane_amd import hip_ops

    unit = synth.silesia_like(32 << 20, config_id=2)
    n, cb = 48, synth.CHUNK_BYTES
    d_unit = torch.from_numpy(unit).cuda()
    d_in = torch.empty(n * cb, dtype=torch.uint8, device="cuda")
    per = unit.size // cb
    host = []
    for t in range(n // per):
        rot = (t * 7919 * 4096 + t * 13) % unit.size
        d_in[t * unit.size:(t + 1) * unit.size] = torch.roll(d_unit, -rot)
        host.append(np.roll(unit, -rot))
    host = np.concatenate(host)
    bound = hip_ops.frame_bound(cb)
    stride = (bound + 255) & ~255
    d_out = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    in_off = np.arange(n, dtype=np.uint64) * cb
    in_len = np.full(n, cb, np.uint64)
    out_off = np.arange(n, dtype=np.uint64) * stride
    out_cap = np.full(n, stride, np.uint64)
    torch.cuda.synchronize()
    out_len, md5 = ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap)
    out_len2, md5b = ctx.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap)
    assert (out_len == out_len2).all() and (md5 == md5b).all()          # idempotent
    frames = d_out.cpu().numpy()
    for i in range(n):
        raw = host[i * cb:(i + 1) * cb].tobytes()
        assert md5[i].tobytes() == hashlib.md5(raw).digest(), i
        f = frames[int(out_off[i]):int(out_off[i]) + int(out_len[i])]
        assert ref.lz4f_decompress(f, cb) == raw, i
    t = ctx.timing()
    assert t.lz4_ms > 0 and t.md5_ms > 0 and t.lz4_in_bytes >= n * cb


def test_frames_written_in_place_equal_the_gathered_frames(monkeypatch):
    """Round 4: device-resident batches of >= 2 chunks per CU take sky_lz4s_frames (a workgroup per chunk at a time writes header, size words, blocks and
    EndMark straight into the frame), smaller ones the block queue + sky_frame_layout / sky_frame_gather.  Forced through BOTH here (SKYHIP_FRAMES_MIN is read
    when a context is created) over ragged, empty, tiny, incompressible and run-length chunks at misaligned input and frame offsets: the frames must be the
    same bytes, equal the sequential model block by block, decode with liblz4, and nothing may be written outside a frame's own length."""
    from skyplane_amd import hip_ops

    rng = synth.rng_for(0, 4242)
    sizes = [0, 1, 12, 13, 64, 4095, 65535, 65536, 65537, 131072 + 5, 200_000, (1 << 20) + 77, 3 * 65536, 65536 * 2 - 1, 70_001, 1 << 20]
    chunks = [synth.gen_class(synth.CLASSES[i % len(synth.CLASSES)], s, rng).tobytes() if s else b"" for i, s in enumerate(sizes)]
    chunks += [bytes(300_000), b"ab" * 40_000 + bytes(5) + b"ab" * 30_000, synth.gen_random(rng, 65536 * 2 + 9).tobytes(), b"\x01" * 70_000 + b"\x02" * 70_000]
    n = len(chunks)
    in_off, pos = np.zeros(n, np.uint64), 3
    for i, c in enumerate(chunks):
        in_off[i] = pos
        pos += len(c) + 7                                   # 7: odd gaps, every alignment of a chunk's first byte
    host_in = np.full(pos + 64, 0x5A, np.uint8)
    for i, c in enumerate(chunks):
        host_in[int(in_off[i]):int(in_off[i]) + len(c)] = np.frombuffer(c, np.uint8)
    in_len = np.array([len(c) for c in chunks], np.uint64)
    out_off, pos = np.zeros(n, np.uint64), 5
    for i, c in enumerate(chunks):
        out_off[i] = pos
        pos += hip_ops.frame_bound(len(c)) + 11
    out_cap = np.array([hip_ops.frame_bound(len(c)) for c in chunks], np.uint64)
    d_in = torch.from_numpy(host_in).cuda()
    got = {}
    for mode, fmin in (("in place", "1"), ("gathered", "0")):
        monkeypatch.setenv("SKYHIP_FRAMES_MIN", fmin)
        c = hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=2 << 20, max_batch=8)
        try:
            d_out = torch.full((pos + 64,), 0xEE, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            out_len, md5 = c.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, out_cap)
            t = c.timing()
            assert t.gather_ms == 0 if mode == "in place" else t.gather_ms > 0, mode       # the path that was asked for is the path that ran
            got[mode] = (out_len.copy(), md5.copy(), d_out.cpu().numpy())
        finally:
            c.close()
    (la, ma, fa), (lb, mb, fb) = got["in place"], got["gathered"]
    assert (la == lb).all() and (ma == mb).all()
    for i, c in enumerate(chunks):
        o, ln = int(out_off[i]), int(la[i])
        frame = fa[o:o + ln].tobytes()
        assert frame == fb[o:o + ln].tobytes(), f"chunk {i} ({len(c)} bytes): frames differ between the two paths"
        assert ma[i].tobytes() == hashlib.md5(c).digest()
        assert ref.lz4f_decompress(frame, len(c)) == c
        lz4smodel.check_frame(c, frame)
        nxt = int(out_off[i + 1]) if i + 1 < n else fa.size
        assert (fa[o + ln:nxt] == 0xEE).all() and (fa[:int(out_off[0])] == 0xEE).all(), f"chunk {i}: bytes written outside the frame"


# ---------------------------------------------------------------------------------------------------------
# Gear CDC + segment fingerprints + dedup table (new capability; spec = oracle/skyoracle.c, parity unpinned)
# ---------------------------------------------------------------------------------------------------------
def _cdc_expect(chunks):
    cuts = [ref.gear_cdc(c) for c in chunks]
    fps = []
    for c, cu in zip(chunks, cuts):
        st = 0
        for e in cu:
            fps.append(hashlib.md5(bytes(c[st:int(e)])).digest())
            st = int(e)
    return cuts, fps


@pytest.mark.parametrize("segment_kernel", ["staged", "lane-streamed"])
def test_cdc_fingerprints_dedup_match_spec(small_cases, segment_kernel, monkeypatch):
    """(through both segment-digest kernels: sky_segment_md5x -- rows staged through LDS, the default -- and sky_segment_md5, SKYHIP_SEGMD5_STAGED=0;
    a context reads the variable at its first CDC call)"""
    from skyplane_amd import hip_ops

    monkeypatch.setenv("SKYHIP_SEGMD5_STAGED", "1" if segment_kernel == "staged" else "0")
    with hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=8 << 20, max_batch=8) as c:
        flags = hip_ops.F_CDC | hip_ops.F_DEDUP | hip_ops.F_MD5 | hip_ops.F_LZ4
        seen = []
        for batch in ([synth.dedup_stream(8 << 20).tobytes(), small_cases["mixed_200k"], b"", bytes(100_000), small_cases["rand_4096"]],
                      [small_cases["mixed_200k"], synth.dedup_stream(1 << 20, config_id=5).tobytes(), bytes(100_000)]):
            res = c.process_batch(batch, flags=flags)
            cuts, efps = _cdc_expect(batch)
            for d, r, cu in zip(batch, res, cuts):
                assert (r.cuts == cu).all() and r.md5 == hashlib.md5(d).digest() and ref.lz4f_decompress(r.frame, len(d)) == d
            prefix, gcuts, fps, first, base = c.cdc_results(len(batch), [len(b) for b in batch])
            assert base == len(seen) and int(prefix[-1]) == len(efps)
            assert [fps[i].tobytes() for i in range(len(fps))] == efps
            allfp = seen + efps
            exp = ref.dedup_first(np.frombuffer(b"".join(allfp), np.uint8).reshape(-1, 16))
            assert (exp[len(seen):] == first).all()
            seen = allfp
        dup = float((first != np.arange(base, base + len(first))).mean())
        assert dup > 0.2          # the second batch repeats earlier content
        c.dedup_reset()
        res = c.process_batch([small_cases["mixed_200k"]], flags=hip_ops.F_CDC | hip_ops.F_DEDUP)
        _, _, _, first, base = c.cdc_results(1, [200_000])
        assert base == 0 and (first == np.arange(len(first))).all()


@pytest.mark.parametrize("n", [1, 64, 4096, 4097, 32768, 32769, 65537, 1 << 20])
def test_cdc_ragged(ctx, n):
    from skyplane_amd import hip_ops

    rng = synth.rng_for(3, n)
    chunks = [synth.gen_random(rng, n).tobytes(), synth.gen_text(rng, n).tobytes(), bytes(n)]
    res = ctx.process_batch(chunks, flags=hip_ops.F_CDC)
    for c, r in zip(chunks, res):
        assert (r.cuts == ref.gear_cdc(c)).all()


def test_cdc_device_path_config3_stream(ctx):
    """Config 3 shape: 50 %-duplicate synthetic stream as 8 MiB chunks, device resident; cut points, duplicate
    fraction and first-seen indices equal the CPU spec."""
    from skyplane_amd import hip_ops

    cb = synth.CHUNK_BYTES
    n = 6
    host = synth.dedup_stream(n * cb, dup_fraction=0.5, config_id=3)
    d_in = torch.from_numpy(host).cuda()
    in_off = np.arange(n, dtype=np.uint64) * cb
    in_len = np.full(n, cb, np.uint64)
    zero = np.zeros(n, np.uint64)
    ctx.dedup_reset()
    ctx.process_device(d_in.data_ptr(), in_off, in_len, 0, zero, zero, flags=hip_ops.F_CDC | hip_ops.F_DEDUP)
    prefix, cuts, fps, first, base = ctx.cdc_results(n, in_len)
    chunks = [host[i * cb:(i + 1) * cb] for i in range(n)]
    ecuts, efps = _cdc_expect(chunks)
    assert np.concatenate(ecuts).tolist() == cuts.tolist()
    assert [fps[i].tobytes() for i in range(len(fps))] == efps
    exp = ref.dedup_first(fps)
    assert (exp == first).all()
    dup_bytes = 0
    seg_len = np.diff(np.concatenate([[0], np.concatenate(ecuts)]))
    seg_len = np.where(seg_len > 0, seg_len, np.concatenate(ecuts))  # first segment of each chunk
    dup_bytes = int(seg_len[first != np.arange(len(first))].sum())
    assert dup_bytes > 0


def test_reference_default_chunk_size_64MiB_unaligned_device_offsets():
    """The reference's default multipart chunk is 64 MiB (skyplane/api/config.py:117).  One such chunk plus a short one,
    placed at odd byte offsets in the device buffers, through LZ4 + MD5 + CDC."""
    from skyplane_amd import hip_ops

    big = np.concatenate([synth.silesia_like(32 << 20, config_id=2), synth.mixed_chunks(4, 8 << 20, config_id=4).reshape(-1)])
    small = synth.gen_text(synth.rng_for(1), 12345)
    assert big.size == 64 << 20
    with hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=64 << 20, max_batch=2) as c:
        d_in = torch.zeros(big.size + small.size + 64, dtype=torch.uint8, device="cuda")
        in_off = np.array([3, 3 + big.size + 7], np.uint64)
        in_len = np.array([big.size, small.size], np.uint64)
        d_in[3:3 + big.size] = torch.from_numpy(big).cuda()
        d_in[int(in_off[1]):int(in_off[1]) + small.size] = torch.from_numpy(small).cuda()
        caps = np.array([hip_ops.frame_bound(big.size), hip_ops.frame_bound(small.size)], np.uint64)
        out_off = np.array([5, 5 + int(caps[0]) + 11], np.uint64)
        d_out = torch.zeros(int(out_off[1] + caps[1]) + 64, dtype=torch.uint8, device="cuda")
        out_len, md5 = c.process_device(d_in.data_ptr(), in_off, in_len, d_out.data_ptr(), out_off, caps, hip_ops.F_LZ4 | hip_ops.F_MD5 | hip_ops.F_CDC)
        frames = d_out.cpu().numpy()
        prefix, cuts, fps, first, base = c.cdc_results(2, in_len)
        for i, raw in enumerate((big, small)):
            assert md5[i].tobytes() == hashlib.md5(raw).digest()
            f = frames[int(out_off[i]):int(out_off[i]) + int(out_len[i])]
            assert ref.lz4f_decompress(f, raw.size) == raw.tobytes()
            assert (cuts[int(prefix[i]):int(prefix[i + 1])] == ref.gear_cdc(raw)).all()
        # and back through the GPU decompressor at odd offsets
        d_back = torch.zeros(big.size + small.size + 64, dtype=torch.uint8, device="cuda")
        olen = c.decompress_device(d_out.data_ptr(), out_off, out_len, d_back.data_ptr(), in_off, in_len)
        assert (olen == in_len).all() and torch.equal(d_back[3:3 + big.size].cpu(), torch.from_numpy(big))


def test_host_batch_pinned_buffers_and_pipelined_sub_batches(monkeypatch):
    """skyhip_process_batch with n > max_batch: one resident group whose LZ4 sub-batches start as their uploads land
    and whose frames leave as they are laid out; with a tiny SKYHIP_STAGE_BYTES the same call runs as several
    consecutive groups reusing the staging area.  Pinned (skyhip_host_alloc) and pageable buffers must give the same
    frames, and every frame/digest must be right."""
    import hashlib
    from skyplane_amd import hip_ops

    rng = synth.rng_for(41)
    sizes = [1 << 20, 0, 70_001, 65536, 13, (1 << 20) - 1, 300_000, 1, 65537, 999_999, 4096]
    chunks = [synth.gen_class(synth.CLASSES[i % len(synth.CLASSES)], s, rng).tobytes() if s else b"" for i, s in enumerate(sizes)]
    with hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=1 << 20, max_batch=4) as c:
        plain = c.process_batch(chunks, flags=3)
        arena_in = c.pinned_buffer(sum((s + 255) & ~255 for s in sizes) + 256)
        bounds = [c.frame_bound(s) for s in sizes]
        arena_out = c.pinned_buffer(sum((b + 255) & ~255 for b in bounds))
        views_in, views_out, pi, po = [], [], 0, 0
        for d, b in zip(chunks, bounds):
            v = arena_in[pi:pi + len(d)]
            v[:] = np.frombuffer(d, np.uint8)
            views_in.append(v)
            views_out.append(arena_out[po:po + b])
            pi += (len(d) + 255) & ~255
            po += (b + 255) & ~255
        for stage in (None, str(3 << 20), None):          # one group / three groups of max_batch / one group again
            if stage is None:
                monkeypatch.delenv("SKYHIP_STAGE_BYTES", raising=False)
            else:
                monkeypatch.setenv("SKYHIP_STAGE_BYTES", stage)
            arena_out[:] = 0
            pinned = c.process_batch(views_in, flags=3, frames_into=views_out)
            again = c.process_batch(chunks, flags=3)
            for d, a, b, g in zip(chunks, plain, pinned, again):
                assert bytes(b.frame) == a.frame == g.frame and a.md5 == b.md5 == g.md5 == hashlib.md5(d).digest()
                assert ref.lz4f_decompress(a.frame, len(d)) == d
        lz4_only = c.process_batch(views_in, flags=1, frames_into=views_out)
        assert [bytes(r.frame) for r in lz4_only] == [r.frame for r in plain] and all(r.md5 is None for r in lz4_only)
        md5_only = c.process_batch(chunks, flags=2)
        assert [r.md5 for r in md5_only] == [r.md5 for r in plain]
        c.release_pinned(arena_in)
        with pytest.raises(hip_ops.SkyHipError):
            c._check(c._lib.skyhip_host_free(c._h, 12345))          # not one of ours


def test_error_paths_leave_the_context_usable(ctx):
    """early returns of the C ABI (capacity errors, malformed frames) must not strand events or leave work in flight --
    provoke each a few dozen times, then the same context must still produce correct results."""
    from skyplane_amd import hip_ops
    d = synth.gen_text(synth.rng_for(5), 300_000).tobytes()
    small = np.empty(1000, np.uint8)                      # far below skyhip_frame_bound
    for _ in range(40):
        with pytest.raises(hip_ops.SkyHipError):
            ctx.process_batch([d, d, d], frames_into=[small, small, small])
        with pytest.raises((hip_ops.SkyHipError, ValueError)):
            ctx.decompress_batch([b"\x04\x22\x4d\x18garbage-not-a-frame", ref.lz4f_compress(d)[:-7]], [100, len(d)])
    # every checked HIP call of a host-buffer batch fails once (skyhip_debug_fault): the call reports an error, the next one is correct
    chunks = [d, d[:70_001], b"", d[::-1]]
    failed = 0
    for n in range(0, 400, 3):
        ctx._lib.skyhip_debug_fault(ctx._h, n)
        try:
            res = ctx.process_batch(chunks)
        except hip_ops.SkyHipError:
            failed += 1
            ctx._lib.skyhip_debug_fault(ctx._h, -1)
            _check(chunks, ctx.process_batch(chunks))
            continue
        ctx._lib.skyhip_debug_fault(ctx._h, -1)
        _check(chunks, res)          # the countdown outlived the call: nothing failed, and from here on nothing will
        break
    assert failed >= 10, failed
    _check([d, b"", d[:70_001]], ctx.process_batch([d, b"", d[:70_001]]))
    assert ctx.decompress_batch([ref.lz4f_compress(d)], [len(d)]) == [d]


def test_dedup_wire_operators_roundtrip_on_gpu(ctx, tmp_path):
    """Dedup on the wire with the real library behind both operators (tests/test_dedup_wire.py runs the same scenario on the emulator): recipes out
    of skyhip_cdc_results, literal streams compressed by a second LZ4-only call, rebuilt and digest-checked on the destination, fewer bytes than frames."""
    from skyplane_amd.gateway import dedup_wire, sidecar
    from tests import test_dedup_wire as T

    chunks = T._dup_chunks(n=8, size=1 << 20)
    src, dst, reqs = T._stores(tmp_path, chunks)
    ctx.dedup_reset()
    comp, dec = T._ops(src, dst, ctx, ctx)
    comp.max_batch = dec.max_batch = 4                     # the fixture's context
    for k in (0, 4):
        assert all(comp.process_batch(reqs[k:k + 4]))
    payloads = [sidecar.compressed_path(src, cr.chunk.chunk_id).read_bytes() for cr in reqs]
    assert all(dedup_wire.is_recipe(p) for p in payloads)
    plain = sum(len(r.frame) for k in (0, 4) for r in ctx.process_batch(chunks[k:k + 4], flags=1))
    assert sum(map(len, payloads)) < 0.8 * plain, (sum(map(len, payloads)), plain)
    T._ship(src, dst, reqs)
    oks = dec.process_batch(reqs[4:])                      # the later half first: references into the first half must wait
    assert all(dec.process_batch(reqs[:4]))
    waiting = [cr for cr, ok in zip(reqs[4:], oks) if not ok]
    assert all(dec.process_batch(waiting)) if waiting else True
    for cr, c in zip(reqs, chunks):
        assert dst.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == c



def test_dedup_literal_streams_are_put_together_on_the_device(ctx):
    """skyhip_dedup_literals (round 5: dedup on the wire without a host gather): for every chunk of the CDC + DEDUP call just made, the LZ4 frame of its NEW
    segments back to back, built from the chunks still resident on the device, decodes (liblz4) to exactly what numpy gathers from the host copy with the
    same cuts and first-seen indices; chunks without duplicates and chunks without new segments get no second frame."""
    from skyplane_amd import hip_ops
    from skyplane_amd.gateway import dedup_wire
    from tests import test_dedup_wire as T

    chunks = T._dup_chunks(n=4, size=1 << 20)
    chunks[2] = chunks[0]                                  # a chunk that is nothing but references
    chunks.append(synth.gen_random(synth.rng_for(5), 300_001).tobytes())      # ... and one without any duplicate (ragged length)
    ctx.dedup_reset()
    res = ctx.process_batch(chunks[:4], flags=hip_ops.F_LZ4 | hip_ops.F_MD5 | hip_ops.F_CDC | hip_ops.F_DEDUP)
    for rnd, batch in enumerate((chunks[:4], chunks[1:5])):
        if rnd:
            res = ctx.process_batch(batch, flags=hip_ops.F_LZ4 | hip_ops.F_MD5 | hip_ops.F_CDC | hip_ops.F_DEDUP)
        lens_in = np.array([len(c) for c in batch], np.uint64)
        prefix, cuts, fps, first, base = ctx.cdc_results(len(batch), lens_in)
        views = [np.empty(hip_ops.frame_bound(len(c)), np.uint8) for c in batch]
        lit_lens, frames = ctx.dedup_literals([len(c) for c in batch], views)
        seen_none = seen_frame = 0
        for i, c in enumerate(batch):
            lens, kinds, _sl = dedup_wire.classify_segments(prefix, cuts, first, base, i)
            ends = np.cumsum(lens.astype(np.int64))
            want = b"".join(c[e - l:e] for e, l, k in zip(ends, lens, kinds) if k == dedup_wire.KIND_LITERAL)
            assert lit_lens[i] == len(want), (rnd, i)
            if len(want) in (0, len(c)):
                assert frames[i] is None, (rnd, i)
                seen_none += 1
            else:
                assert ref.lz4f_decompress(frames[i].tobytes(), len(want)) == want, (rnd, i)
                seen_frame += 1
        assert seen_none and (seen_frame or rnd == 1), (rnd, seen_none, seen_frame)      # (round 1: every chunk but the random one was seen whole in round 0)
    # the call is only valid right after its CDC + DEDUP batch
    ctx.process_batch(chunks[:2], flags=hip_ops.F_LZ4)
    with pytest.raises(Exception):
        ctx.dedup_literals([len(c) for c in chunks[:2]], [np.empty(hip_ops.frame_bound(len(c)), np.uint8) for c in chunks[:2]])


def test_device_resident_literal_streams_and_gathered_chunks(ctx):
    """skyhip_decompress_to_device + skyhip_gather_md5 (dedup on the wire, destination side): frames decoded into device memory that Python owns, chunks put
    together on the device from byte runs of several such buffers -- any alignment, runs of 1 byte to megabytes, an empty chunk --, digested there and
    copied out; the memory outlives the context that allocated it and is readable by another context of the process."""
    from skyplane_amd import hip_ops

    rng = np.random.default_rng(9)
    streams = [synth.gen_text(synth.rng_for(3, k), n).tobytes() for k, n in enumerate((1 << 20, 70_001, 13, 300_000))]
    bufs = ctx.decompress_to_device([ref.lz4f_compress(s) for s in streams], [len(s) for s in streams])
    assert [len(b) for b in bufs] == [len(s) for s in streams]
    chunks_src, chunks_len, want = [], [], []
    for _c in range(5):
        src, ln, blob = [], [], b""
        for _r in range(int(rng.integers(1, 200))):
            k = int(rng.integers(0, len(streams)))
            n = int(min(rng.integers(1, 1 << int(rng.integers(1, 18))), len(streams[k])))
            o = int(rng.integers(0, len(streams[k]) - n + 1))
            src.append(bufs[k].dptr + o); ln.append(n); blob += streams[k][o:o + n]
            if len(blob) > (6 << 20):
                break
        chunks_src.append(np.array(src, np.uint64)); chunks_len.append(np.array(ln, np.uint32)); want.append(blob)
    chunks_src.append(np.zeros(0, np.uint64)); chunks_len.append(np.zeros(0, np.uint32)); want.append(b"")      # an empty chunk
    into = [np.empty(max(len(w), 1), np.uint8) for w in want]
    outs, digs = ctx.gather_md5(chunks_src, chunks_len, into)
    for o, d, w in zip(outs, digs, want):
        assert o.tobytes() == w and d == hashlib.md5(w).digest()
    # another context of the process reads the same device memory; the first context's buffers survive ITS closing
    other = hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=8 << 20, max_batch=4)
    tmp = hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=8 << 20, max_batch=4)
    (tb,) = tmp.decompress_to_device([ref.lz4f_compress(streams[1])], [len(streams[1])])
    tmp.close()
    outs, digs = other.gather_md5([np.array([tb.dptr, bufs[0].dptr + 5], np.uint64)], [np.array([len(tb), 1000], np.uint32)], [np.empty(len(tb) + 1000, np.uint8)])
    assert outs[0].tobytes() == streams[1] + streams[0][5:1005] and digs[0] == hashlib.md5(streams[1] + streams[0][5:1005]).digest()
    del tb, bufs
    other.close()


def test_segment_md5_of_device_resident_ranges(ctx):
    """skyhip_segment_md5_device (round 6): the digests of thousands of byte ranges that are already in device memory -- every residue of the RFC 1321
    padding, lengths 0 .. 32767, any alignment -- equal hashlib's; what gpu_decompress checks a recipe's literal segments with."""
    stream = synth.silesia_like(3 << 20, config_id=5, seg_min=3000, seg_max=40000).tobytes()
    (buf,) = ctx.decompress_to_device([ref.lz4f_compress(stream)], [len(stream)])
    rng = np.random.default_rng(11)
    lens = np.concatenate([np.arange(0, 200), [16384, 16383, 32767, 1024, 4096], rng.integers(1, 16385, 3000)]).astype(np.uint32)
    offs = rng.integers(0, len(stream) - 32768, lens.size).astype(np.uint64)
    got = ctx.segment_md5_device(np.uint64(buf.dptr) + offs, lens)
    for o, n, g in zip(offs, lens, got):
        assert g.tobytes() == hashlib.md5(stream[int(o):int(o) + int(n)]).digest(), (int(o), int(n))
    from skyplane_amd import hip_ops

    with pytest.raises(hip_ops.SkyHipError):
        ctx.segment_md5_device(np.array([buf.dptr], np.uint64), np.array([32768], np.uint32))      # a length the descriptor cannot hold


def test_device_segment_store_keeps_what_a_gather_still_reads(ctx):
    """Regression for GPU call r5t (round 5: a memory access fault at 1024 chunks): the device segment store resolves fingerprints to ADDRESSES; another lane
    then moves the group's lane past keep_epochs, the group is dropped and -- without an owner on the reader's side -- its device memory is freed (and handed
    out again) before the gather has read it.  get_arrays returns the group's buffers; while the caller holds them the bytes stay what they were."""
    import gc

    from skyplane_amd.gateway import dedup_wire

    store = dedup_wire.DeviceSegmentStore(keep_epochs=2, max_bytes=1 << 30)
    stream = synth.gen_text(synth.rng_for(4, 1), 4 << 20).tobytes()
    seg = 8192
    n = len(stream) // seg
    fps = np.frombuffer(b"".join(hashlib.md5(stream[k * seg:(k + 1) * seg]).digest() for k in range(n)), np.uint8).reshape(n, 16)
    (buf,) = ctx.decompress_to_device([ref.lz4f_compress(stream)], [len(stream)])
    store.put_arrays(7, 0, fps, np.uint64(buf.dptr) + np.arange(n, dtype=np.uint64) * seg, np.full(n, seg, np.uint32), buf)
    first_ptr = buf.dptr
    del buf
    addrs, lens, miss, keep = store.get_arrays(7, 0, fps)      # lane A resolved its references ...
    assert miss == 0 and keep
    other = bytes(len(stream))
    junk = []
    for epoch in (1, 2, 3):                                     # ... lane B carries the sender's lane three epochs on: epoch 0 is retired
        (b2,) = ctx.decompress_to_device([ref.lz4f_compress(other)], [len(other)])
        store.put_arrays(7, epoch, fps[:1], np.array([b2.dptr], np.uint64), np.array([seg], np.uint32), b2)
        junk.append(b2)
    gc.collect()
    assert store.epochs_held(7) == [2, 3]
    for _ in range(8):                                          # memory the allocator would hand out again if epoch 0's block had been freed
        (b3,) = ctx.decompress_to_device([ref.lz4f_compress(other)], [len(other)])
        junk.append(b3)
    assert all(j.dptr != first_ptr for j in junk)
    outs, digs = ctx.gather_md5([addrs], [lens], [np.empty(len(stream), np.uint8)])
    assert digs[0] == hashlib.md5(stream).digest() and outs[0].tobytes() == stream
    got = ctx.segment_md5_device(addrs, lens)
    assert (got == fps).all()
    del keep, junk
    store.cleanup()
