"""Thin Python face of libskyhip.so -- the `skyplane/gateway/hip_ops` extension named by BASELINE.json.

Replaces, for one batch of chunks, the two CPU calls of the reference's source-gateway hot path:
  * ``lz4.frame.compress(data)``           skyplane/gateway/operators/gateway_operator.py:358-361
  * ``hashlib.md5(...).digest()``           skyplane/obj_store/s3_interface.py:181-192 (requested at gateway_operator.py:555-565)
and adds Gear CDC cut points / segment fingerprints / dedup lookups (new; not in the reference).

Everything here is plumbing over the C ABI in include/skyhip.h; all arithmetic runs in HIP kernels on gfx950.
There is no CPU fallback: construction raises if the extension or the GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import SkyHipError, Timing

F_LZ4, F_MD5, F_CDC, F_DEDUP = 1, 2, 4, 8
CDC_MIN_SEGMENT = 1024      # SKY_CDC_MIN of csrc/gear_kernel.inc: a chunk of n bytes has at most n // CDC_MIN_SEGMENT + 2 cut points


def frame_bound(raw_len: int) -> int:
    return int(_lib.load().skyhip_frame_bound(raw_len))


@dataclass
class ChunkResult:
    frame: Optional[bytes]   # LZ4 frame (what GatewaySender puts on the wire when is_compressed=True)
    md5: Optional[bytes]     # 16-byte digest == hashlib.md5(raw).digest()
    cuts: Optional[np.ndarray] = None  # CDC END offsets (uint32), last == len(raw)


class _DeviceBlock:
    """Device memory owned from Python (skyhip_dev_alloc): freed when the last DeviceBuffer cut from it goes -- also after its context was closed."""

    def __init__(self, ctx: "SkyHipContext", nbytes: int):
        p = C.c_void_p()
        ctx._check(ctx._lib.skyhip_dev_alloc(ctx._h, int(nbytes), C.byref(p)))
        self.ptr, self.nbytes, self._ctx = int(p.value), int(nbytes), ctx
        ctx._dev_blocks.add(self)

    def free(self):
        ctx, self._ctx = self._ctx, None
        if ctx is not None and self.ptr:
            ctx._lib.skyhip_dev_free(ctx._h if ctx._h else None, C.c_void_p(self.ptr))      # (the memory is the process's: it may outlive its context)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """`nbytes` bytes of device memory at `dptr` (a piece of a block it keeps alive): what the destination's segment store holds instead of bytes when the
    chunks are put together on the device."""

    __slots__ = ("block", "off", "nbytes")

    def __init__(self, block: _DeviceBlock, off: int, nbytes: int):
        self.block, self.off, self.nbytes = block, int(off), int(nbytes)

    @property
    def dptr(self) -> int:
        return self.block.ptr + self.off

    def __len__(self) -> int:
        return self.nbytes


class SkyHipContext:
    """One per worker process, created AFTER fork (HIP must never be initialised in the daemon parent)."""

    def __init__(self, device_id: int = 0, max_chunk_bytes: int = 64 << 20, max_batch: int = 8):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.skyhip_create(device_id, max_chunk_bytes, max_batch, C.byref(h))
        if rc != 0:
            raise SkyHipError(rc, self._lib.skyhip_strerror(rc).decode())
        self._h = h
        self._pinned = {}
        self._dev_blocks = weakref.WeakSet()      # device memory handed to Python (decompress_to_device): bookkeeping only, it is freed with its last reference
        self.device_id, self.max_chunk_bytes, self.max_batch = device_id, max_chunk_bytes, max_batch

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._pinned.clear()            # skyhip_destroy frees every pinned block still alive (device blocks handed to Python -- _DeviceBlock -- are the
                                            # process's: another lane's context may still be reading them through the shared segment store; they go
                                            # with their last reference)
            self._lib.skyhip_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            detail = self._lib.skyhip_last_hip_error(self._h).decode()
            raise SkyHipError(rc, self._lib.skyhip_strerror(rc).decode() + (f" [{detail}]" if detail else ""))

    @staticmethod
    def frame_bound(raw_len: int) -> int:
        return frame_bound(raw_len)

    # -- pinned host memory (zero-copy hand-off) ---------------------------------
    def pinned_buffer(self, nbytes: int) -> np.ndarray:
        """uint8 array over DMA-able host memory (skyhip_host_alloc).  Read chunk files straight into it
        (``f.readinto(buf[a:b])``) and pass views to process_batch: the H2D/D2H copies then run asynchronously and
        overlap the kernels of the neighbouring sub-batch.  Valid until release_pinned(buf) or close()."""
        p = C.c_void_p()
        self._check(self._lib.skyhip_host_alloc(self._h, int(nbytes), C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(p.value))
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def release_pinned(self, buf: np.ndarray):
        p = self._pinned.pop(buf.ctypes.data, None)
        if p is not None and self._h:
            self._check(self._lib.skyhip_host_free(self._h, C.c_void_p(p)))

    def register_host(self, buf: np.ndarray):
        """Page-lock memory the caller owns (a MAP_SHARED mapping of an arena file: gateway/shm_arena.py) so that copies to and from it are
        asynchronous DMA like those of a pinned_buffer.  Once per arena, not per chunk."""
        self._check(self._lib.skyhip_host_register(self._h, C.c_void_p(buf.ctypes.data), int(buf.nbytes)))

    def unregister_host(self, buf: np.ndarray):
        if self._h:
            self._check(self._lib.skyhip_host_unregister(self._h, C.c_void_p(buf.ctypes.data)))

    # -- host-buffer path (what the gateway operator uses) -------------------------------------------
    def process_batch(self, chunks: Sequence, flags: int = F_LZ4 | F_MD5, frames_into: Optional[Sequence[np.ndarray]] = None) -> List[ChunkResult]:
        """chunks: bytes-like objects or uint8 arrays (views of a pinned_buffer for the asynchronous path).
        frames_into: optional uint8 arrays (>= frame_bound(len) each, ideally pinned views) that receive the frames;
        ChunkResult.frame is then a view of them instead of a fresh bytes object (no copy on the way out)."""
        n = len(chunks)
        if n == 0:
            return []
        arrs = [np.frombuffer(c, np.uint8) if not isinstance(c, np.ndarray) else np.ascontiguousarray(c.reshape(-1).view(np.uint8)) for c in chunks]
        in_ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        in_len = (C.c_size_t * n)(*[a.size for a in arrs])
        outs, out_ptrs, out_cap, out_len = [], None, None, None
        if flags & F_LZ4:
            if frames_into is not None:
                outs = list(frames_into)
                if len(outs) != n or any(o.dtype != np.uint8 or not o.flags["C_CONTIGUOUS"] or not o.flags["WRITEABLE"] for o in outs):
                    raise ValueError("frames_into: one writable contiguous uint8 array per chunk")
            else:
                outs = [np.empty(frame_bound(a.size), np.uint8) for a in arrs]
            out_ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
            out_cap = (C.c_size_t * n)(*[o.size for o in outs])
            out_len = (C.c_size_t * n)()
        md5 = np.zeros((n, 16), np.uint8) if flags & F_MD5 else None
        cuts, cut_ptrs, cut_cap, n_cuts = [], None, None, None
        if flags & F_CDC:
            cuts = [np.empty(a.size // CDC_MIN_SEGMENT + 2, np.uint32) for a in arrs]
            cut_ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in cuts])
            cut_cap = (C.c_size_t * n)(*[c.size for c in cuts])
            n_cuts = (C.c_size_t * n)()
        rc = self._lib.skyhip_process_batch(self._h, n, in_ptrs, in_len, out_ptrs, out_cap, out_len, md5.ctypes.data if md5 is not None else None,
                                            cut_ptrs, cut_cap, n_cuts, flags)
        self._check(rc)
        res = []
        for i in range(n):
            frame = None
            if flags & F_LZ4:
                frame = outs[i][: out_len[i]] if frames_into is not None else outs[i][: out_len[i]].tobytes()
            res.append(ChunkResult(frame=frame,
                                   md5=md5[i].tobytes() if md5 is not None else None,
                                   cuts=cuts[i][: n_cuts[i]].copy() if flags & F_CDC else None))
        return res

    def dedup_literals(self, in_lens: Sequence[int], frames_into: Sequence[np.ndarray]):
        """Dedup on the wire, source side: right after ``process_batch(chunks, flags=F_LZ4 | ... | F_CDC | F_DEDUP)`` over the same chunks, the LZ4 frame
        of every chunk's literal stream (its new segments back to back), put together and compressed on the device from the chunks still resident there
        (skyhip_dedup_literals).  frames_into[i]: uint8 array of >= frame_bound(in_lens[i]) bytes, ideally pinned.  Returns (lit_lens, frames): frames[i]
        is a view of frames_into[i], or None where the chunk has no duplicates (the first call's frame is its literal stream) or no new segment."""
        n = len(in_lens)
        outs = list(frames_into)
        assert len(outs) == n and all(isinstance(o, np.ndarray) and o.dtype == np.uint8 for o in outs)
        out_ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        out_cap = (C.c_size_t * n)(*[o.size for o in outs])
        out_len = (C.c_size_t * n)()
        lit_len = (C.c_size_t * n)()
        self._check(self._lib.skyhip_dedup_literals(self._h, n, out_ptrs, out_cap, out_len, lit_len))
        return [int(x) for x in lit_len], [outs[i][: out_len[i]] if out_len[i] else None for i in range(n)]

    # -- device-resident path (bench / kernel-only measurements; pointers are raw device addresses) ----
    def process_device(self, d_in: int, in_off: np.ndarray, in_len: np.ndarray, d_out: int, out_off: np.ndarray, out_cap: np.ndarray,
                       flags: int = F_LZ4 | F_MD5, want_md5: bool = True):
        n = int(in_off.size)
        in_off = np.ascontiguousarray(in_off, np.uint64)
        in_len = np.ascontiguousarray(in_len, np.uint64)
        out_off = np.ascontiguousarray(out_off, np.uint64)
        out_cap = np.ascontiguousarray(out_cap, np.uint64)
        out_len = np.zeros(n, np.uint64)
        md5 = np.zeros((n, 16), np.uint8) if (flags & F_MD5 and want_md5) else None
        rc = self._lib.skyhip_process_device(self._h, n, C.c_void_p(d_in), in_off.ctypes.data, in_len.ctypes.data, C.c_void_p(d_out) if d_out else None,
                                             out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data,
                                             md5.ctypes.data if md5 is not None else None, flags)
        self._check(rc)
        return out_len, md5

    # -- decompression (destination gateway: lz4.frame.decompress at gateway_receiver.py:195-201) ----------
    def decompress_batch(self, frames: Sequence, raw_lens: Sequence[int], want_md5: bool = False, into: Optional[Sequence[np.ndarray]] = None):
        """frames[i] decodes into at most raw_lens[i] bytes (WireProtocolHeader.raw_data_len). Returns a list of bytes
        (views of `into` when given: uint8 arrays of >= raw_lens[i] bytes, ideally pinned); with want_md5 also the
        digests of the decoded bytes, computed on the device: (outs, digests).
        Raises SkyHipError(-8) if any frame is malformed (like lz4.frame.decompress raising)."""
        n = len(frames)
        if n == 0:
            return ([], []) if want_md5 else []
        arrs = [np.frombuffer(f, np.uint8) if not isinstance(f, np.ndarray) else np.ascontiguousarray(f.reshape(-1).view(np.uint8)) for f in frames]
        outs = list(into) if into is not None else [np.empty(max(int(r), 1), np.uint8) for r in raw_lens]
        if len(outs) != n or any(o.size < int(r) for o, r in zip(outs, raw_lens)):
            raise ValueError("into: one uint8 array of at least raw_lens[i] bytes per frame")
        in_ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        in_len = (C.c_size_t * n)(*[a.size for a in arrs])
        out_ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        out_cap = (C.c_size_t * n)(*[int(r) for r in raw_lens])
        out_len = (C.c_size_t * n)()
        status = (C.c_int32 * n)()
        md5 = np.zeros((n, 16), np.uint8) if want_md5 else None
        rc = self._lib.skyhip_decompress_batch_md5(self._h, n, in_ptrs, in_len, out_ptrs, out_cap, out_len, status,
                                                   md5.ctypes.data if md5 is not None else None)
        self.last_decode_status = list(status)
        self._check(rc)
        res = [outs[i][: out_len[i]] if into is not None else outs[i][: out_len[i]].tobytes() for i in range(n)]
        return (res, [md5[i].tobytes() for i in range(n)]) if want_md5 else res

    # -- dedup on the wire, destination side: literal streams that stay on the device (gateway/dedup_wire.py) ----
    def decompress_to_device(self, frames: Sequence, raw_lens: Sequence[int]) -> List["DeviceBuffer"]:
        """Decode frames[i] (host memory) into device memory that this call allocates -- ONE block for the batch -- and return a DeviceBuffer per frame.
        Nothing is copied back; the block is freed when the last DeviceBuffer cut from it is garbage-collected (or the context is closed)."""
        n = len(frames)
        if n == 0:
            return []
        arrs = [np.frombuffer(f, np.uint8) if not isinstance(f, np.ndarray) else np.ascontiguousarray(f.reshape(-1).view(np.uint8)) for f in frames]
        offs, tot = [], 0
        for r in raw_lens:
            offs.append(tot)
            tot += (int(r) + 255) & ~255
        block = _DeviceBlock(self, max(tot, 256))
        in_ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        in_len = (C.c_size_t * n)(*[a.size for a in arrs])
        dst = (C.c_void_p * n)(*[block.ptr + o for o in offs])
        out_cap = (C.c_size_t * n)(*[int(r) for r in raw_lens])
        out_len = (C.c_size_t * n)()
        status = (C.c_int32 * n)()
        rc = self._lib.skyhip_decompress_to_device(self._h, n, in_ptrs, in_len, dst, out_cap, out_len, status)
        self.last_decode_status = list(status)
        self._check(rc)
        return [DeviceBuffer(block, o, int(out_len[i])) for i, o in enumerate(offs)]

    def gather_md5(self, run_src: Sequence[np.ndarray], run_len: Sequence[np.ndarray], into: Sequence[np.ndarray], want_md5: bool = True):
        """Chunk i := the byte runs (DEVICE address run_src[i][k], run_len[i][k] bytes) back to back, put together on the device, copied into into[i]
        (ideally pinned) and digested there.  Returns (views of into, digests or None)."""
        n = len(into)
        assert len(run_src) == n and len(run_len) == n
        prefix = np.zeros(n + 1, np.uint64)
        prefix[1:] = np.cumsum([len(x) for x in run_len])
        src = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint64) for x in run_src]) if n else np.zeros(0, np.uint64), np.uint64)
        ln = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint32) for x in run_len]) if n else np.zeros(0, np.uint32), np.uint32)
        out_ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in into])
        out_cap = (C.c_size_t * n)(*[o.size for o in into])
        out_len = (C.c_size_t * n)()
        md5 = np.zeros((n, 16), np.uint8) if want_md5 else None
        self._check(self._lib.skyhip_gather_md5(self._h, n, prefix.ctypes.data, src.ctypes.data if src.size else None, ln.ctypes.data if ln.size else None,
                                                out_ptrs, out_cap, out_len, md5.ctypes.data if md5 is not None else None))
        return [into[i][: out_len[i]] for i in range(n)], ([md5[i].tobytes() for i in range(n)] if want_md5 else None)

    def segment_md5_device(self, addrs, lens) -> np.ndarray:
        """MD5 of byte ranges that are already in device memory (skyhip_segment_md5_device): addrs [n] uint64 DEVICE addresses, lens [n] uint32 (< 32768).
        Returns [n, 16] uint8.  Thousands of independent messages at once: how a destination checks a recipe's literal segments against their fingerprints."""
        addrs = np.ascontiguousarray(addrs, np.uint64)
        lens = np.ascontiguousarray(lens, np.uint32)
        n = int(lens.size)
        assert addrs.size == n
        fps = np.zeros((n, 16), np.uint8)
        if n:
            self._check(self._lib.skyhip_segment_md5_device(self._h, n, addrs.ctypes.data, lens.ctypes.data, fps.ctypes.data))
        return fps

    def decompress_device(self, d_in: int, in_off, in_len, d_out: int, out_off, out_cap):
        n = int(len(in_off))
        in_off = np.ascontiguousarray(in_off, np.uint64); in_len = np.ascontiguousarray(in_len, np.uint64)
        out_off = np.ascontiguousarray(out_off, np.uint64); out_cap = np.ascontiguousarray(out_cap, np.uint64)
        out_len = np.zeros(n, np.uint64)
        status = np.zeros(n, np.int32)
        rc = self._lib.skyhip_decompress_device(self._h, n, C.c_void_p(d_in), in_off.ctypes.data, in_len.ctypes.data, C.c_void_p(d_out),
                                                out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data)
        self.last_decode_status = status.tolist()
        self._check(rc)
        return out_len

    def decompress_ms(self, reset: bool = True) -> float:
        return float(self._lib.skyhip_decompress_ms(self._h, int(reset)))

    def cdc_results(self, n: int, in_len: np.ndarray):
        """CDC output of the last call with F_CDC: (cut_prefix[n+1], cuts, fingerprints[nseg,16], first_seen[nseg], seg_base)."""
        cap = int(sum(int(l) // CDC_MIN_SEGMENT + 2 for l in in_len))
        prefix = np.zeros(n + 1, np.uint64)
        cuts = np.zeros(max(cap, 1), np.uint32)
        fps = np.zeros((max(cap, 1), 16), np.uint8)
        first = np.zeros(max(cap, 1), np.uint64)
        base = np.zeros(1, np.uint64)
        rc = self._lib.skyhip_cdc_results(self._h, n, prefix.ctypes.data, cuts.ctypes.data, cap, fps.ctypes.data, first.ctypes.data, base.ctypes.data)
        self._check(rc)
        nseg = int(prefix[n])
        return prefix, cuts[:nseg], fps[:nseg], first[:nseg], int(base[0])

    def dedup_reset(self):
        self._check(self._lib.skyhip_dedup_reset(self._h))

    def selftest(self) -> int:
        return int(self._lib.skyhip_selftest(self._h))

    def timing(self) -> Timing:
        t = Timing()
        self._lib.skyhip_get_timing(self._h, C.byref(t))
        return t

    def reset_timing(self):
        self._lib.skyhip_reset_timing(self._h)
