"""Python face of the CPU oracle (test infrastructure only; never imported by the product).

Two independent things live here:

1. ``liblz4`` -- a ctypes binding of the *system* liblz4 (1.9.3 in this image), i.e. the very C
   library python-lz4 wraps.  ``lz4f_compress`` reproduces ``lz4.frame.compress(data)`` with
   python-lz4 4.3.2 defaults exactly as the reference calls it at
   skyplane/gateway/operators/gateway_operator.py:358-361 (zeroed LZ4F_preferences_t except
   frameInfo.contentSize = len(src): 64 KiB linked blocks, level 0, no checksums, FLG=0x48 BD=0x40).
   ``lz4f_decompress`` reproduces ``lz4.frame.decompress`` (gateway_receiver.py:195-201).
   This is "the reference run here" for the compress/decompress arithmetic (cpu_baseline kind
   "reference").

2. ``sko`` -- libskyoracle.so, the C restatement in skyoracle.c (MD5, LZ4 frame/block decoder with
   strict format checks, XXH32, wire header, Gear CDC spec, dedup spec, greedy LZ4 port).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libskyoracle.so"


def build(force: bool = False) -> Path:
    """gcc-compile skyoracle.c (seconds). Building the checker is not using it."""
    if force or not _SO.exists() or _SO.stat().st_mtime < (_HERE / "skyoracle.c").stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    return _SO


_sko = None


def sko() -> C.CDLL:
    global _sko
    if _sko is None:
        build()
        lib = C.CDLL(str(_SO))
        u8p = C.POINTER(C.c_uint8)
        lib.sko_md5.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.sko_md5.restype = None
        lib.sko_md5_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        lib.sko_md5_batch.restype = None
        lib.sko_xxh32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        lib.sko_xxh32.restype = C.c_uint32
        lib.sko_lz4_block_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        lib.sko_lz4_block_decode.restype = C.c_long
        lib.sko_lz4f_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        lib.sko_lz4f_decompress.restype = C.c_long
        lib.sko_lz4_block_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.sko_lz4_block_compress.restype = C.c_size_t
        lib.sko_lz4f_compress_port.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.sko_lz4f_compress_port.restype = C.c_size_t
        lib.sko_wire_header.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p]
        lib.sko_wire_header.restype = None
        lib.sko_gear_table.argtypes = [C.c_void_p]
        lib.sko_gear_table.restype = None
        lib.sko_gear_cdc.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
        lib.sko_gear_cdc.restype = C.c_size_t
        lib.sko_dedup.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        lib.sko_dedup.restype = C.c_int
        del u8p
        _sko = lib
    return _sko


def _buf(b) -> np.ndarray:
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    return np.ascontiguousarray(a.reshape(-1).view(np.uint8))


# ----------------------------------------------------------------------------------------------
# C restatement wrappers
# ----------------------------------------------------------------------------------------------
def md5(data) -> bytes:
    a = _buf(data)
    out = np.empty(16, np.uint8)
    sko().sko_md5(a.ctypes.data, a.size, out.ctypes.data)
    return out.tobytes()


def xxh32(data, seed: int = 0) -> int:
    a = _buf(data)
    return int(sko().sko_xxh32(a.ctypes.data, a.size, seed))


class OracleError(ValueError):
    pass


_ERR = {-1: "truncated", -2: "bad offset", -3: "output overrun", -4: "end-of-block rule violated", -5: "bad magic",
        -6: "bad FLG/BD", -7: "checksum mismatch", -8: "block larger than BD max", -9: "content size mismatch",
        -10: "trailing bytes after frame", -11: "output capacity too small"}


def lz4f_decode(frame, cap: int, strict: bool = True):
    """Decode a whole LZ4 frame with the C restatement. Returns (bytes, info dict)."""
    f = _buf(frame)
    out = np.empty(max(cap, 1), np.uint8)
    info = np.zeros(4, np.uint32)
    r = sko().sko_lz4f_decompress(f.ctypes.data, f.size, out.ctypes.data, cap, int(strict), info.ctypes.data)
    if r < 0:
        raise OracleError(f"oracle lz4f decode failed: {_ERR.get(int(r), r)}")
    return out[:r].tobytes(), {"flg": int(info[0]), "bd": int(info[1]), "blocks": int(info[2]), "raw_blocks": int(info[3])}


def lz4f_compress_port(data) -> bytes:
    a = _buf(data)
    cap = 15 + a.size + 4 * ((a.size + 65535) // 65536) + 4 + 64
    out = np.empty(cap, np.uint8)
    n = sko().sko_lz4f_compress_port(a.ctypes.data, a.size, out.ctypes.data, cap)
    if n == 0:
        raise OracleError("port compress failed")
    return out[:n].tobytes()


def wire_header(chunk_id_hex: str, data_len: int, raw_len: int, is_compressed: bool, n_left: int) -> bytes:
    cid = np.frombuffer(bytes.fromhex(chunk_id_hex), np.uint8).copy()
    assert cid.size == 16
    out = np.empty(53, np.uint8)
    sko().sko_wire_header(cid.ctypes.data, data_len, raw_len, int(is_compressed), n_left, out.ctypes.data)
    return out.tobytes()


def gear_table() -> np.ndarray:
    t = np.empty(256, np.uint64)
    sko().sko_gear_table(t.ctypes.data)
    return t


# Frozen CDC parameters (ours; not in the reference): min / avg / max = 1 KiB / 4 KiB / 16 KiB, masks on high bits.
CDC_MIN, CDC_AVG, CDC_MAX = 1024, 4096, 16384
CDC_MASK_S = 0xFFFC000000000000  # 14 bits: harder, used before the average size
CDC_MASK_L = 0xFFC0000000000000  # 10 bits: easier, used after it (subset of MASK_S)


def gear_cdc(data, min_size=CDC_MIN, avg_size=CDC_AVG, max_size=CDC_MAX, mask_s=CDC_MASK_S, mask_l=CDC_MASK_L) -> np.ndarray:
    a = _buf(data)
    cap = a.size // max(min_size, 1) + 2
    cuts = np.empty(cap, np.uint32)
    n = sko().sko_gear_cdc(a.ctypes.data, a.size, min_size, avg_size, max_size, mask_s, mask_l, cuts.ctypes.data, cap)
    assert n <= cap
    return cuts[:n].copy()


def dedup_first(fps: np.ndarray, base_index: int = 0) -> np.ndarray:
    fps = np.ascontiguousarray(fps, dtype=np.uint8).reshape(-1, 16)
    first = np.empty(fps.shape[0], np.uint64)
    rc = sko().sko_dedup(fps.ctypes.data, fps.shape[0], base_index, first.ctypes.data)
    assert rc == 0
    return first


# ----------------------------------------------------------------------------------------------
# The real third-party library the reference calls: liblz4 via ctypes
# ----------------------------------------------------------------------------------------------
class _FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_uint), ("blockMode", C.c_uint), ("contentChecksumFlag", C.c_uint), ("frameType", C.c_uint),
                ("contentSize", C.c_ulonglong), ("dictID", C.c_uint), ("blockChecksumFlag", C.c_uint)]


class _Prefs(C.Structure):
    _fields_ = [("frameInfo", _FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint), ("favorDecSpeed", C.c_uint),
                ("reserved", C.c_uint * 3)]


_lz4 = None


def liblz4() -> C.CDLL:
    global _lz4
    if _lz4 is None:
        last = None
        for cand in ("/usr/lib/x86_64-linux-gnu/liblz4.so.1", "liblz4.so.1", ctypes.util.find_library("lz4"), "/opt/conda/lib/liblz4.so.1"):
            if not cand:
                continue
            try:
                lib = C.CDLL(cand)
                break
            except OSError as e:  # pragma: no cover
                last = e
        else:  # pragma: no cover
            raise OSError(f"system liblz4 not found: {last}")
        lib.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(_Prefs)]
        lib.LZ4F_compressFrameBound.restype = C.c_size_t
        lib.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_Prefs)]
        lib.LZ4F_compressFrame.restype = C.c_size_t
        lib.LZ4F_isError.argtypes = [C.c_size_t]
        lib.LZ4F_isError.restype = C.c_uint
        lib.LZ4F_getErrorName.argtypes = [C.c_size_t]
        lib.LZ4F_getErrorName.restype = C.c_char_p
        lib.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        lib.LZ4F_createDecompressionContext.restype = C.c_size_t
        lib.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
        lib.LZ4F_freeDecompressionContext.restype = C.c_size_t
        lib.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
        lib.LZ4F_decompress.restype = C.c_size_t
        lib.LZ4_versionString.restype = C.c_char_p
        _lz4 = lib
    return _lz4


def liblz4_version() -> str:
    return liblz4().LZ4_versionString().decode()


def lz4f_compress(data, store_size: bool = True, block_linked: bool = True, block_size_id: int = 0) -> bytes:
    """== lz4.frame.compress(data) as called at gateway_operator.py:359 (python-lz4 defaults).  The keyword
    arguments produce the other frame flavours a receiver may meet (python-lz4's store_size / block_linked /
    block_size knobs): no content size, independent blocks, 256 KiB..4 MiB blocks (ids 5..7)."""
    a = _buf(data)
    lib = liblz4()
    prefs = _Prefs()
    prefs.frameInfo.contentSize = a.size if store_size else 0  # python-lz4 store_size=True
    prefs.frameInfo.blockMode = 0 if block_linked else 1
    prefs.frameInfo.blockSizeID = block_size_id
    bound = lib.LZ4F_compressFrameBound(a.size, C.byref(prefs))
    out = np.empty(bound, np.uint8)
    n = lib.LZ4F_compressFrame(out.ctypes.data, bound, a.ctypes.data, a.size, C.byref(prefs))
    if lib.LZ4F_isError(n):
        raise OracleError(lib.LZ4F_getErrorName(n).decode())
    return out[:n].tobytes()


def lz4f_compress_stream(pieces, store_size: bool = True, block_linked: bool = True) -> bytes:
    """A frame made the streaming way -- LZ4F_compressBegin, then LZ4F_compressUpdate + LZ4F_flush per piece, LZ4F_compressEnd -- as a producer
    that flushes (python-lz4's LZ4FrameCompressor.flush()) makes it: every flush closes a block early, so the frame holds blocks SHORTER than the
    block maximum in its middle.  A receiver must take such frames (lz4.frame.decompress does)."""
    lib = liblz4()
    lib.LZ4F_createCompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    lib.LZ4F_createCompressionContext.restype = C.c_size_t
    lib.LZ4F_freeCompressionContext.argtypes = [C.c_void_p]
    lib.LZ4F_compressBegin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_Prefs)]
    lib.LZ4F_compressBegin.restype = C.c_size_t
    lib.LZ4F_compressBound.argtypes = [C.c_size_t, C.POINTER(_Prefs)]
    lib.LZ4F_compressBound.restype = C.c_size_t
    lib.LZ4F_compressUpdate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.LZ4F_compressUpdate.restype = C.c_size_t
    lib.LZ4F_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.LZ4F_flush.restype = C.c_size_t
    lib.LZ4F_compressEnd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.LZ4F_compressEnd.restype = C.c_size_t
    pieces = [_buf(p) for p in pieces]
    prefs = _Prefs()
    prefs.frameInfo.contentSize = sum(p.size for p in pieces) if store_size else 0
    prefs.frameInfo.blockMode = 0 if block_linked else 1
    cctx = C.c_void_p()
    r = lib.LZ4F_createCompressionContext(C.byref(cctx), 100)
    if lib.LZ4F_isError(r):
        raise OracleError(lib.LZ4F_getErrorName(r).decode())
    try:
        cap = 64 + sum(int(lib.LZ4F_compressBound(p.size, C.byref(prefs))) + 64 for p in pieces) + int(lib.LZ4F_compressBound(0, C.byref(prefs)))
        out = np.empty(cap, np.uint8)
        pos = 0

        def chk(n):
            if lib.LZ4F_isError(n):
                raise OracleError(lib.LZ4F_getErrorName(n).decode())
            return int(n)

        pos += chk(lib.LZ4F_compressBegin(cctx, out.ctypes.data, cap, C.byref(prefs)))
        for p in pieces:
            pos += chk(lib.LZ4F_compressUpdate(cctx, out.ctypes.data + pos, cap - pos, p.ctypes.data if p.size else None, p.size, None))
            pos += chk(lib.LZ4F_flush(cctx, out.ctypes.data + pos, cap - pos, None))
        pos += chk(lib.LZ4F_compressEnd(cctx, out.ctypes.data + pos, cap - pos, None))
        return out[:pos].tobytes()
    finally:
        lib.LZ4F_freeCompressionContext(cctx)


def lz4f_compress_into(a: np.ndarray, out: np.ndarray) -> int:
    """Allocation-free variant for timing loops."""
    lib = liblz4()
    prefs = _Prefs()
    prefs.frameInfo.contentSize = a.size
    n = lib.LZ4F_compressFrame(out.ctypes.data, out.size, a.ctypes.data, a.size, C.byref(prefs))
    if lib.LZ4F_isError(n):
        raise OracleError(lib.LZ4F_getErrorName(n).decode())
    return int(n)


def lz4f_frame_bound(n: int) -> int:
    prefs = _Prefs()
    prefs.frameInfo.contentSize = n
    return int(liblz4().LZ4F_compressFrameBound(n, C.byref(prefs)))


def lz4f_decompress(frame, expected_len: int) -> bytes:
    """== lz4.frame.decompress(frame) (gateway_receiver.py:196): must consume the whole frame."""
    f = _buf(frame)
    lib = liblz4()
    ctx = C.c_void_p()
    rc = lib.LZ4F_createDecompressionContext(C.byref(ctx), 100)
    if lib.LZ4F_isError(rc):
        raise OracleError(lib.LZ4F_getErrorName(rc).decode())
    try:
        out = np.empty(max(expected_len, 1) + 64, np.uint8)
        ip, op = 0, 0
        hint = 1
        while ip < f.size and hint != 0:
            src_sz = C.c_size_t(f.size - ip)
            dst_sz = C.c_size_t(out.size - op)
            hint = lib.LZ4F_decompress(ctx, out.ctypes.data + op, C.byref(dst_sz), f.ctypes.data + ip, C.byref(src_sz), None)
            if lib.LZ4F_isError(hint):
                raise OracleError("liblz4: " + lib.LZ4F_getErrorName(hint).decode())
            ip += src_sz.value
            op += dst_sz.value
            if src_sz.value == 0 and dst_sz.value == 0:
                raise OracleError("liblz4: no progress (output larger than expected_len?)")
        if hint != 0:
            raise OracleError("liblz4: frame incomplete")
        if ip != f.size:
            raise OracleError("liblz4: trailing bytes after frame")
        return out[:op].tobytes()
    finally:
        lib.LZ4F_freeDecompressionContext(ctx)


def hashlib_md5(data) -> bytes:
    """== hashlib.md5() loop of s3_interface.py:181-192 (64 KiB updates), .digest() (bytes)."""
    a = _buf(data)
    m = hashlib.md5()
    mv = memoryview(a)
    for i in range(0, a.size, 1 << 16):
        m.update(mv[i:i + (1 << 16)])
    return m.digest()
