"""Loader for libskyhost.so (csrc/skyhost.c): the fingerprint -> (address, length) map behind the destination's device-resident segment store
(gateway/dedup_wire.py::DeviceSegmentStore).  Plain C, built next to libskyhip.so by the same Makefile; like it, there is no pure-Python substitute."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PATH = Path(__file__).resolve().parent / "csrc" / "libskyhost.so"
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _PATH.exists():
            raise ImportError(f"{_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(str(_PATH))
        vp = C.c_void_p
        lib.skyhost_map_new.argtypes = [C.c_uint32]
        lib.skyhost_map_new.restype = vp
        lib.skyhost_map_free.argtypes = [vp]
        lib.skyhost_map_free.restype = None
        lib.skyhost_map_count.argtypes = [vp]
        lib.skyhost_map_count.restype = C.c_uint64
        lib.skyhost_map_put.argtypes = [vp, C.c_int64, vp, vp, vp, C.POINTER(C.c_uint64)]
        lib.skyhost_map_put.restype = C.c_int64
        lib.skyhost_map_get.argtypes = [vp, C.c_int64, vp, vp, vp]
        lib.skyhost_map_get.restype = C.c_int64
        _lib = lib
    return _lib
