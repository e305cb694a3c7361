"""Run the shipping kernels ON THE GPU on buffers fenced by unmapped address ranges -- the device-side twin of tests/_guard_cases.py.

Executed by tests/test_gpu_guard.py in a child interpreter with SKYHIP_GUARD_ALLOC=1 (the library then places every device buffer it allocates for itself
against an unmapped page, with no slack) and AMD_SERIALIZE_KERNEL=3: an access one byte outside a buffer ends this process with "Memory access fault by
GPU node-N", which the parent reports as a failing test.  The caller-owned buffers (d_in / d_out of the device-resident calls) come from
skyhip_debug_guard_alloc: the last input byte is the last mapped byte, the last byte of the last frame region likewise.  TEST INFRASTRUCTURE ONLY.

usage: python tests/_gpu_guard_run.py {probe_in|probe_over|probe_under|lz4|batch|lz4d|cdc|dedup|frames512|smoke}
"""
import ctypes as C
import hashlib
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from oracle import ref  # noqa: E402
from tests import _guard_cases as gc  # noqa: E402  (patterns and sizes shared with the emulator's guard suite; importing it builds nothing)

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemset.restype = C.c_int
        _hip.hipDeviceSynchronize.restype = C.c_int
    return _hip


def dsync():
    """hipMemset (and a pageable hipMemcpy's last leg) may still be running when the call returns, and the library's streams are non-blocking: they do not
    wait for the null stream.  Everything this harness puts on the device is complete before the library sees it."""
    assert hip().hipDeviceSynchronize() == 0


class Guarded:
    """`nbytes` of device memory whose LAST byte is the last mapped byte (at_end) or whose FIRST byte is the first mapped one."""

    def __init__(self, lib, nbytes: int, at_end: bool = True, fill=None):
        self.lib, self.n = lib, int(nbytes)
        p = C.c_void_p()
        rc = lib.skyhip_debug_guard_alloc(max(self.n, 1), 1 if at_end else 0, C.byref(p))
        assert rc == 0, f"skyhip_debug_guard_alloc({nbytes}) -> {rc}"
        self.ptr = int(p.value)
        self.user = self.ptr + (max(self.n, 1) - self.n if at_end else 0)      # (a zero-byte buffer: one byte was mapped, the buffer is the empty range at its end)
        if fill is not None:
            assert hip().hipMemset(self.ptr, fill, max(self.n, 1)) == 0
            dsync()

    def upload(self, data, at: int = 0):
        b = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
        if b.size:
            assert hip().hipMemcpy(self.user + at, b.ctypes.data, b.size, 1) == 0
            dsync()

    def download(self, at: int, n: int) -> bytes:
        out = np.empty(n, np.uint8)
        if n:
            assert hip().hipMemcpy(out.ctypes.data, self.user + at, n, 2) == 0
        return out.tobytes()

    def free(self):
        assert self.lib.skyhip_debug_guard_free(self.ptr) == 0


def _ctx(max_chunk=1 << 20, max_batch=4):
    from skyplane_amd import hip_ops

    return hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=max_chunk, max_batch=max_batch)


def run_probe(kind: str):
    from skyplane_amd import _lib

    lib = _lib.load()
    with _ctx() as ctx:
        if kind == "probe_under":
            g = Guarded(lib, 5000, at_end=False, fill=0x5A)
            print("about to read one byte BEFORE a guarded buffer", flush=True)
            lib.skyhip_debug_guard_probe(ctx._h, g.user - 1)
            print("SURVIVED probe_under")
            return
        g = Guarded(lib, 5000, at_end=True, fill=0x5A)
        assert lib.skyhip_debug_guard_probe(ctx._h, g.user + 4999) == 0x5A and lib.skyhip_debug_guard_probe(ctx._h, g.user) == 0x5A
        if kind == "probe_over":
            print("about to read one byte PAST a guarded buffer", flush=True)
            lib.skyhip_debug_guard_probe(ctx._h, g.user + 5000)
            print("SURVIVED probe_over")
            return
        g.free()
    print("OK probe_in")


def _device_case(lib, ctx, chunks, at_end=True, flags=3):
    """One skyhip_process_device call: the chunks back to back, the last input byte against the fence (or the first one behind it); the frame regions
    back to back with exactly skyhip_frame_bound bytes each, the last region's last byte against the fence."""
    from skyplane_amd import hip_ops

    lens = np.array([len(c) for c in chunks], np.uint64)
    in_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    bounds = np.array([hip_ops.frame_bound(int(n)) for n in lens], np.uint64)
    out_off = np.concatenate([[0], np.cumsum(bounds)[:-1]]).astype(np.uint64)
    gin = Guarded(lib, int(lens.sum()), at_end=at_end)
    gout = Guarded(lib, int(bounds.sum()), at_end=True, fill=0xEE)
    gin.upload(b"".join(chunks))
    out_len, md5 = ctx.process_device(gin.user, in_off, lens, gout.user, out_off, bounds, flags)
    for i, d in enumerate(chunks):
        assert md5[i].tobytes() == hashlib.md5(d).digest(), (i, len(d))
        frame = gout.download(int(out_off[i]), int(out_len[i]))
        assert ref.lz4f_decompress(frame, len(d)) == d, (i, len(d))
    gin.free(); gout.free()


def run_lz4():
    """every size / pattern of the emulator's guard suite as the ONLY chunk of a device-resident call, by the block-queue kernel and (SKYHIP_FRAMES_MIN=1 in
    the environment of a second run) by the in-place one"""
    from skyplane_amd import _lib

    lib = _lib.load()
    n_cases = 0
    with _ctx(max_chunk=1 << 20, max_batch=4) as ctx:
        for n in gc.sizes_lz4():
            for name, data in gc.patterns(n, n + 7):
                if n >= gc.BLK - 1 and name in ("period3",):
                    continue
                for at_end in (True, False):
                    print(f"lz4 n={n} {name} at_end={at_end}", flush=True)
                    _device_case(lib, ctx, [data], at_end=at_end)
                    n_cases += 1
        _device_case(lib, ctx, [bytes(70000), b"x" * 13, b"", bytes(gc.BLK)])
    print(f"OK lz4 ({n_cases} cases, frames_min={os.environ.get('SKYHIP_FRAMES_MIN', 'default')})")


def run_batch():
    """the HOST-buffer calls (what the operator and smoke() use): the staging areas the library allocates are the guarded ones"""
    n_cases = 0
    with _ctx(max_chunk=1 << 20, max_batch=4) as ctx:
        for n in gc.sizes_lz4():
            for name, data in gc.patterns(n, n + 7):
                if n >= gc.BLK - 1 and name in ("period3",):
                    continue
                print(f"batch n={n} {name}", flush=True)
                (r,) = ctx.process_batch([data])
                assert r.md5 == hashlib.md5(data).digest() and ref.lz4f_decompress(r.frame, n) == data, (n, name)
                outs, digs = ctx.decompress_batch([ref.lz4f_compress(data)], [n], want_md5=True)
                assert outs == [data] and digs == [hashlib.md5(data).digest()], (n, name)
                n_cases += 1
        batch = [bytes(70000), b"x" * 13, b"", bytes(gc.BLK), np.random.default_rng(3).integers(0, 256, 200001, dtype=np.uint8).tobytes()]
        for r, d in zip(ctx.process_batch(batch), batch):
            assert r.md5 == hashlib.md5(d).digest() and ref.lz4f_decompress(r.frame, len(d)) == d
        outs, digs = ctx.decompress_batch([ref.lz4f_compress(d) for d in batch], [len(d) for d in batch], want_md5=True)
        assert outs == batch
    print(f"OK batch ({n_cases} cases)")


def run_lz4d():
    """frames -- liblz4's (block-linked, the reference sender's default), this library's, and the crafted sequence shapes -- decoded by
    skyhip_decompress_device with the last frame byte and the last output byte against the fences"""
    from skyplane_amd import _lib
    from tests._crafted_frames import crafted_frames

    lib = _lib.load()
    cases = []
    for n in (0, 1, 13, 64, 4096, gc.BLK - 1, gc.BLK, gc.BLK + 13, 2 * gc.BLK + 77, 300001):
        for name, data in gc.patterns(n, n + 5):
            if n >= gc.BLK - 1 and name in ("period3",):
                continue
            cases.append((f"liblz4 {name} {n}", ref.lz4f_compress(data), data))
    for name, (frame, data) in crafted_frames(2).items():
        cases.append((f"crafted {name}", frame, data))
    with _ctx(max_chunk=1 << 20, max_batch=4) as ctx:
        for r, d in zip(ctx.process_batch([c[2] for c in cases[:40:3]]), [c[2] for c in cases[:40:3]]):      # and our own frames of some of them
            cases.append((f"ours {len(d)}", r.frame, d))
        for label, frame, data in cases:
            print(f"lz4d {label}", flush=True)
            gin = Guarded(lib, len(frame), at_end=True)
            gout = Guarded(lib, len(data), at_end=True, fill=0xEE)
            gin.upload(frame)
            out_len = ctx.decompress_device(gin.user, np.array([0], np.uint64), np.array([len(frame)], np.uint64), gout.user,
                                            np.array([0], np.uint64), np.array([len(data)], np.uint64))
            assert ctx.last_decode_status[0] == 0 and int(out_len[0]) == len(data), (label, ctx.last_decode_status[0], int(out_len[0]))
            assert gout.download(0, len(data)) == data, label
            gin.free(); gout.free()
    print(f"OK lz4d ({len(cases)} frames)")


def run_cdc():
    from skyplane_amd import _lib, hip_ops

    lib = _lib.load()
    n_cases = 0
    with _ctx(max_chunk=1 << 20, max_batch=4) as ctx:
        for n in (1, 63, 64, 65, 4095, 4096, 4097, 32767, 32768, 32769, 100000, 3 * 65536):
            for name, data in gc.patterns(n, n + 11):
                if name in ("period3", "period4"):
                    continue
                want = [int(x) for x in ref.gear_cdc(data)]
                for at_end in (True, False):
                    print(f"cdc n={n} {name} at_end={at_end}", flush=True)
                    gin = Guarded(lib, n, at_end=at_end)
                    gin.upload(data)
                    ctx.dedup_reset()
                    lens = np.array([n], np.uint64)
                    ctx.process_device(gin.user, np.array([0], np.uint64), lens, 0, np.array([0], np.uint64), np.array([0], np.uint64),
                                       hip_ops.F_CDC | hip_ops.F_DEDUP | hip_ops.F_MD5)
                    prefix, cuts, fps, first, base = ctx.cdc_results(1, lens)
                    got = [int(x) for x in cuts[: int(prefix[1])]]
                    assert got == want, (n, name, at_end)
                    lo = 0
                    for k, hi_ in enumerate(got):
                        assert fps[k].tobytes() == hashlib.md5(data[lo:hi_]).digest(), (n, name, k)
                        lo = hi_
                    gin.free()
                    n_cases += 1
    print(f"OK cdc ({n_cases} cases)")


def run_dedup():
    """dedup on the wire on the device: literal streams put together from the staged chunks (skyhip_dedup_literals), frames decoded into device blocks,
    chunks gathered from runs that end on a block's last byte (skyhip_decompress_to_device / skyhip_gather_md5) -- staging areas, the literal buffer and
    the device blocks all end at an unmapped page"""
    from skyplane_amd import hip_ops, synth
    from skyplane_amd.gateway import dedup_wire

    n_cases = 0
    with _ctx(max_chunk=1 << 20, max_batch=4) as ctx:
        for size in (1 << 20, 300_001, 70_000):
            stream = synth.dedup_stream(4 * size, dup_fraction=0.5, config_id=3)
            chunks = [stream[i * size:(i + 1) * size].tobytes() for i in range(4)]
            chunks[3] = chunks[3][: size - 13]
            ctx.dedup_reset()
            ctx.process_batch(chunks, flags=hip_ops.F_LZ4 | hip_ops.F_MD5 | hip_ops.F_CDC | hip_ops.F_DEDUP)
            lens_in = np.array([len(c) for c in chunks], np.uint64)
            prefix, cuts, fps, first, base = ctx.cdc_results(4, lens_in)
            views = [np.empty(hip_ops.frame_bound(len(c)), np.uint8) for c in chunks]
            lit_lens, frames = ctx.dedup_literals([len(c) for c in chunks], views)
            lits = []
            for i, c in enumerate(chunks):
                lens, kinds, _sl = dedup_wire.classify_segments(prefix, cuts, first, base, i)
                ends = np.cumsum(lens.astype(np.int64))
                want = b"".join(c[e - l:e] for e, l, k in zip(ends, lens, kinds) if k == dedup_wire.KIND_LITERAL)
                assert lit_lens[i] == len(want)
                if frames[i] is not None:
                    assert ref.lz4f_decompress(frames[i].tobytes(), len(want)) == want
                lits.append(want)
            print(f"dedup literals size={size}", flush=True)
            bufs = ctx.decompress_to_device([ref.lz4f_compress(l) for l in lits if l], [len(l) for l in lits if l])
            live = [l for l in lits if l]
            # a chunk made of every stream's LAST bytes (the run ends where the block's mapping ends) and first bytes
            src, ln, blob = [], [], b""
            for b, l in zip(bufs, live):
                k = min(len(l), 4099)
                src += [b.dptr + len(l) - k, b.dptr]; ln += [k, min(len(l), 17)]; blob += l[len(l) - k:] + l[:min(len(l), 17)]
            outs, digs = ctx.gather_md5([np.array(src, np.uint64)], [np.array(ln, np.uint32)], [np.empty(len(blob), np.uint8)])
            assert outs[0].tobytes() == blob and digs[0] == hashlib.md5(blob).digest()
            del bufs
            n_cases += 1
    print(f"OK dedup ({n_cases} cases)")


_POOL_DATA = {}


def _w_check(args):
    i, frame, digest = args
    d = _POOL_DATA["chunks"][i]
    return hashlib.md5(d).digest() == digest and ref.lz4f_decompress(frame, len(d)) == d.tobytes()


def run_frames512():
    """ONE sky_lz4s_frames launch at production shape, taken by the library's own rule (>= 2 chunks of 8 MiB per CU): 512 x 8 MiB of the Silesia-like
    stream, input and frame slots against the fences, EVERY frame decoded with liblz4 and every digest compared with hashlib."""
    import multiprocessing as mp

    from skyplane_amd import synth

    n, cb = int(os.environ.get("GUARD_FRAMES_N", "512")), 8 << 20
    unit = synth.silesia_like(64 << 20, config_id=2)
    mixed = synth.mixed_chunks(8, cb, config_id=4)
    chunks = [(unit[(k % 8) * cb:(k % 8 + 1) * cb] if k % 5 else mixed[k % 8]) for k in range(n)]      # every fifth chunk from the mixed stream (raw blocks, sparse runs)
    chunks = [np.roll(c, 4099 * k) if k >= 8 else c for k, c in enumerate(chunks)]
    chunks[-1] = chunks[-1][: cb - 37]                       # a ragged last chunk: its last block ends 37 bytes short, right at the fence
    _POOL_DATA["chunks"] = chunks
    pool = mp.get_context("fork").Pool(min(16, len(os.sched_getaffinity(0))))      # forked BEFORE the HIP runtime exists in this process
    from skyplane_amd import _lib, hip_ops

    lib = _lib.load()
    with hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=cb, max_batch=64) as ctx:
        lens = np.array([c.size for c in chunks], np.uint64)
        in_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        bounds = np.array([hip_ops.frame_bound(int(x)) for x in lens], np.uint64)
        out_off = np.concatenate([[0], np.cumsum(bounds)[:-1]]).astype(np.uint64)
        gin, gout = Guarded(lib, int(lens.sum())), Guarded(lib, int(bounds.sum()), fill=0xEE)
        for c, o in zip(chunks, in_off):
            gin.upload(np.ascontiguousarray(c), int(o))
        ctx.reset_timing()
        out_len, md5 = ctx.process_device(gin.user, in_off, lens, gout.user, out_off, bounds, 3)
        tm = ctx.timing()
        assert tm.lz4_launches == 1, f"{tm.lz4_launches} compressor launches: the in-place path takes a batch of {n} chunks in one"
        jobs = [(i, gout.download(int(out_off[i]), int(out_len[i])), md5[i].tobytes()) for i in range(n)]
        bad = [i for i, ok in enumerate(pool.imap(_w_check, jobs, chunksize=4)) if not ok]
        assert not bad, f"frames / digests of chunks {bad[:8]} are wrong"
        gin.free(); gout.free()
    pool.close(); pool.join()
    print(f"OK frames512 ({n} chunks, {tm.lz4_ms:.1f} ms in one launch, ratio {float(lens.sum()) / float(out_len.sum()):.3f})")


def run_smoke():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g

    for _ in range(int(os.environ.get("GUARD_SMOKES", "3"))):
        g.smoke()
    print("OK smoke")


if __name__ == "__main__":
    what = sys.argv[1]
    {"probe_in": lambda: run_probe("probe_in"), "probe_over": lambda: run_probe("probe_over"), "probe_under": lambda: run_probe("probe_under"),
     "lz4": run_lz4, "batch": run_batch, "lz4d": run_lz4d, "cdc": run_cdc, "dedup": run_dedup, "frames512": run_frames512, "smoke": run_smoke}[what]()
