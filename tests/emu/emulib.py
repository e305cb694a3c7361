"""ctypes face of tests/emu/_build/libskyemu.so: the shipping kernel source run under the CPU SIMT
emulator.  TEST INFRASTRUCTURE ONLY -- lets `-m "not gpu"` tests execute the kernels' logic without a GPU."""
from __future__ import annotations

import ctypes as C
import mmap
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libskyemu.so"
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
        l = C.CDLL(str(_SO))
        l.emu_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        l.emu_process.restype = C.c_int
        l.emu_slot_bytes.restype = C.c_uint32
        l.emu_decompress.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5
        l.emu_decompress.restype = C.c_int
        l.emu_cdc.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_size_t] + [C.c_void_p] * 5 + [C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        l.emu_gather_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        l.emu_gather_runs.restype = None
        l.emu_cdc.restype = C.c_long
        _lib = l
    return _lib


_PAGE = mmap.PAGESIZE
_libc = C.CDLL(None, use_errno=True)


class Guarded:
    """n bytes with an inaccessible page right after them (tight="end") or right before them (tight="start"):
    a kernel that touches one byte outside its buffer takes SIGSEGV instead of silently reading a neighbour.
    (That is how a 16-byte over-read at the very end of a 64 GiB device buffer was reproduced on the CPU.)"""

    def __init__(self, n: int, tight: str = "end", fill: int = 0):
        npages = max(1, (n + _PAGE - 1) // _PAGE)
        self._mm = mmap.mmap(-1, (npages + 2) * _PAGE)
        self._all = np.frombuffer(self._mm, np.uint8)
        base = self._all.ctypes.data
        assert base % _PAGE == 0
        lo = _PAGE + (npages * _PAGE - n if tight == "end" else 0)
        self.view = self._all[lo:lo + n]
        self._all[_PAGE:_PAGE + npages * _PAGE] = fill
        self.addr = base + lo
        for a in (base, base + (npages + 1) * _PAGE):
            if _libc.mprotect(C.c_void_p(a), C.c_size_t(_PAGE), 0) != 0:
                raise OSError(C.get_errno(), "mprotect")


def frame_bound(n: int) -> int:
    return 15 + n + 4 * ((n + 65535) // 65536) + 4


def _pack_inputs(chunks, guard):
    """Lay the chunks out back to back (64-byte aligned starts).  guard=None: slack after the last chunk;
    "end"/"start": the packed buffer sits flush against an inaccessible page on that side."""
    n = len(chunks)
    in_off = np.zeros(n, np.uint64)
    pos = 0
    end = 0
    for i, c in enumerate(chunks):
        in_off[i] = pos
        end = pos + len(c)
        pos += (len(c) + 63) & ~63
    if guard is None:
        buf = np.zeros(max(pos, 64) + 64, np.uint8)
        addr, keep = buf.ctypes.data, buf
    else:
        keep = Guarded(end, guard)
        buf, addr = keep.view, keep.addr
    for i, c in enumerate(chunks):
        buf[int(in_off[i]):int(in_off[i]) + len(c)] = np.frombuffer(c, np.uint8)
    return in_off, addr, keep


def _process_once(chunks, flags=3, blk_skew=0, guard=None):
    """chunks: list of bytes. Returns (frames: list[bytes], md5s: list[bytes], csizes).
    guard: None | "end" | "start" -- see Guarded; input AND output buffers are then exactly as large as the C ABI
    promises (sum of chunk bytes / frame bounds) and fenced by inaccessible pages."""
    n = len(chunks)
    lens = np.array([len(c) for c in chunks], np.uint64)
    in_off, in_addr, _keep_in = _pack_inputs(chunks, guard)
    out_off = np.zeros(n, np.uint64)
    pos = 3 if guard is None else 0  # deliberately misaligned frame starts
    for i, c in enumerate(chunks):
        out_off[i] = pos
        pos += frame_bound(len(c)) + (5 if i + 1 < n or guard is None else 0)
    if guard is None:
        out = np.full(pos + 64, 0xEE, np.uint8)
        out_addr = out.ctypes.data
    else:
        _keep_out = Guarded(pos, guard, fill=0xEE)
        out, out_addr = _keep_out.view, _keep_out.addr
    flen = np.zeros(n, np.uint64)
    md5 = np.zeros((n, 16), np.uint8)
    nb = int(sum((len(c) + 65535) // 65536 for c in chunks))
    cs = np.zeros(max(nb, 1), np.uint32)
    rc = lib().emu_process(in_addr, in_off.ctypes.data, lens.ctypes.data, n, out_addr, out_off.ctypes.data, flen.ctypes.data,
                           md5.ctypes.data, flags, blk_skew, cs.ctypes.data)
    assert rc == 0
    frames = [out[int(out_off[i]):int(out_off[i]) + int(flen[i])].tobytes() for i in range(n)]
    # guard bytes between frames must be untouched
    for i in range(n):
        end = int(out_off[i]) + int(flen[i])
        nxt = int(out_off[i + 1]) if i + 1 < n else out.size
        if flags & 1:
            assert (out[end:nxt] == 0xEE).all(), f"frame {i} wrote past its length"
    return frames, [md5[i].tobytes() for i in range(n)], cs[:nb]


def process(chunks, flags=3, blk_skew=0, guard=None):
    """Both launch sequences of libskyhip under the emulator: block queue + scratch + frame layout / gather (small batches) and frames written in place
    by a workgroup per chunk (large device-resident batches).  Their frames must be the same bytes; the first path's results are returned."""
    res = _process_once(chunks, flags, blk_skew, guard)
    if flags & 1:
        again = _process_once(chunks, (flags & ~2) | 4, 0, guard)
        assert again[0] == res[0], "frames written in place differ from the frames the gather pass puts together"
    return res


class EmuCdc:
    """CDC + fingerprints + dedup under the emulator; keeps the dedup table across calls like a skyhip context."""

    def __init__(self, slots_log2=14):
        self.slots_log2 = slots_log2
        self.key_lo = np.zeros(1 << slots_log2, np.uint64)
        self.key_hi = np.zeros(1 << slots_log2, np.uint64)
        self.first = np.full(1 << slots_log2, 0xFFFFFFFFFFFFFFFF, np.uint64)
        self.seg_base = 0

    def run(self, chunks, gear, dedup=True, guard=None, literals=False):
        """literals=True (with dedup): also every chunk's literal stream -- its NEW segments back to back, put together by the shipping sky_lit_plan /
        sky_lit_gather kernels (skyhip_dedup_literals) -- as a seventh result, a list of bytes; with a guard the streams' buffer ends at a fenced page."""
        n = len(chunks)
        lens = np.array([len(c) for c in chunks], np.uint64)
        in_off, in_addr, _keep_in = _pack_inputs(chunks, guard)
        cap = int(sum(len(c) // 1024 + 2 for c in chunks))
        prefix = np.zeros(n + 1, np.uint32)
        seg_end = np.zeros(cap, np.uint32)
        fps = np.zeros((cap, 16), np.uint8)
        first = np.zeros(cap, np.uint64)
        ntiles = int(sum((len(c) + 32767) // 32768 for c in chunks))
        cc = np.zeros(max(ntiles, 1), np.uint32)
        gear = np.ascontiguousarray(gear, np.uint64)
        lit_addr, lit_stride, lit_len, lit_view = None, 0, np.zeros(max(n, 1), np.uint32), None
        if literals and dedup and n:
            lit_stride = max(int(lens.max()), 1)
            if guard is None:
                lit_view = np.full(n * lit_stride + 64, 0xEE, np.uint8)
                lit_addr = lit_view.ctypes.data
            else:
                _keep_lit = Guarded(n * lit_stride, guard, fill=0xEE)
                lit_view, lit_addr = _keep_lit.view, _keep_lit.addr
        tot = lib().emu_cdc(in_addr, in_off.ctypes.data, lens.ctypes.data, n, gear.ctypes.data, prefix.ctypes.data, seg_end.ctypes.data, cap,
                            fps.ctypes.data, first.ctypes.data, self.key_lo.ctypes.data, self.key_hi.ctypes.data, self.first.ctypes.data,
                            self.slots_log2, self.seg_base, int(dedup), cc.ctypes.data, lit_addr, lit_stride, lit_len.ctypes.data)
        assert tot >= 0, tot
        base = self.seg_base
        if dedup:
            self.seg_base += tot
        res = (prefix, seg_end[:tot], fps[:tot], (first[:tot] if dedup else None), base, cc[:ntiles])
        if literals and dedup:
            streams = [lit_view[i * lit_stride:i * lit_stride + int(lit_len[i])].tobytes() for i in range(n)]
            for i in range(n):      # nothing written behind a stream's end
                assert (lit_view[i * lit_stride + int(lit_len[i]):(i + 1) * lit_stride] == 0xEE).all(), f"chunk {i}: wrote past its literal stream"
            res += (streams,)
        return res


def gather_runs(src_addrs, dst_addrs, lens):
    """sky_gather_runs (skyhip_gather_md5's copy step) under the emulator: run k = lens[k] bytes from src_addrs[k] to dst_addrs[k] (host addresses)."""
    src = np.ascontiguousarray(src_addrs, np.uint64)
    dst = np.ascontiguousarray(dst_addrs, np.uint64)
    ln = np.ascontiguousarray(lens, np.uint32)
    assert src.size == dst.size == ln.size
    lib().emu_gather_runs(src.ctypes.data, dst.ctypes.data, ln.ctypes.data, int(ln.size))


def set_segmd5_staged(on):
    """Which of the library's two segment-digest kernels EmuCdc.run uses: True = sky_segment_md5x (rows staged through LDS: the library's default), False =
    sky_segment_md5 (every lane streams its own segment; SKYHIP_SEGMD5_STAGED=0)."""
    lib().emu_set_segmd5_staged(1 if on else 0)


def set_link_resolve(on):
    """Which of the library's two ways block-linked frames take in decompress(): True = sky_lz4_resolve + sky_lz4_chain (the library's choice up to
    SKYHIP_LINK_RESOLVE_MAX frames per call), False = sky_lz4_link (larger batches)."""
    lib().emu_set_link_resolve(1 if on else 0)


def decompress(frames, caps, guard=None):
    """frames: list[bytes]; caps: list[int] output capacities. Returns (rc, outputs list[bytes], status list[int]).
    guard: as in process()."""
    n = len(frames)
    in_len = np.array([len(f) for f in frames], np.uint64)
    in_off = np.zeros(n, np.uint64)
    pos = 5 if guard is None else 0  # misaligned on purpose
    end = 0
    for i, f in enumerate(frames):
        in_off[i] = pos
        end = pos + len(f)
        pos += len(f) + 3
    if guard is None:
        ibuf = np.full(pos + 64, 0x5A, np.uint8)
        iaddr = ibuf.ctypes.data
    else:
        _keep_in = Guarded(end, guard, fill=0x5A)
        ibuf, iaddr = _keep_in.view, _keep_in.addr
    for i, f in enumerate(frames):
        ibuf[int(in_off[i]):int(in_off[i]) + len(f)] = np.frombuffer(f, np.uint8)
    out_off = np.zeros(n, np.uint64)
    out_cap = np.array(caps, np.uint64)
    pos = 7 if guard is None else 0
    end = 0
    for i in range(n):
        out_off[i] = pos
        end = pos + int(caps[i])
        pos += int(caps[i]) + 9
    if guard is None:
        obuf = np.full(pos + 64, 0xEE, np.uint8)
        oaddr = obuf.ctypes.data
    else:
        _keep_out = Guarded(end, guard, fill=0xEE)
        obuf, oaddr = _keep_out.view, _keep_out.addr
    out_len = np.zeros(n, np.uint64)
    status = np.zeros(n, np.int32)
    rc = lib().emu_decompress(iaddr, in_off.ctypes.data, in_len.ctypes.data, n, oaddr, out_off.ctypes.data, out_cap.ctypes.data,
                              out_len.ctypes.data, status.ctypes.data)
    outs = [obuf[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)]
    for i in range(n):   # nothing written past the capacity
        end = int(out_off[i]) + int(caps[i])
        nxt = int(out_off[i + 1]) if i + 1 < n else obuf.size
        assert (obuf[end:nxt] == 0xEE).all(), f"frame {i}: wrote past its capacity"
    return rc, outs, status.tolist()
