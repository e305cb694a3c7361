"""Loader for libskyhip.so (the gfx950 HIP extension).  There is no CPU fallback: if the library is missing
or cannot be loaded this raises, and every caller above it fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(os.environ.get("SKYHIP_LIB_PATH", _CSRC / "libskyhip.so"))   # override only for dev builds (profiling)

EXPORTS = (
    "skyhip_abi_version", "skyhip_create", "skyhip_destroy", "skyhip_frame_bound", "skyhip_process_batch", "skyhip_process_device",
    "skyhip_cdc_results", "skyhip_dedup_reset", "skyhip_get_timing", "skyhip_reset_timing", "skyhip_selftest", "skyhip_strerror",
    "skyhip_last_hip_error", "skyhip_debug_prof", "skyhip_decompress_device", "skyhip_decompress_batch", "skyhip_decompress_ms",
    "skyhip_host_alloc", "skyhip_host_free", "skyhip_decompress_batch_md5", "skyhip_debug_fault", "skyhip_host_register", "skyhip_host_unregister",
    "skyhip_debug_guard_alloc", "skyhip_debug_guard_free", "skyhip_debug_guard_probe", "skyhip_dedup_literals",
    "skyhip_dev_alloc", "skyhip_dev_free", "skyhip_decompress_to_device", "skyhip_gather_md5", "skyhip_segment_md5_device",
)


class SkyHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"skyhip error {code}: {msg}")
        self.code = code


class Timing(C.Structure):
    _fields_ = [("lz4_ms", C.c_double), ("layout_ms", C.c_double), ("gather_ms", C.c_double), ("md5_ms", C.c_double), ("cdc_ms", C.c_double),
                ("lz4_launches", C.c_uint64), ("lz4_in_bytes", C.c_uint64), ("lz4_out_bytes", C.c_uint64), ("md5_launches", C.c_uint64),
                ("md5_in_bytes", C.c_uint64)]


_lib = None


def build(verbose: bool = False) -> Path:
    """Compile the extension in-tree with hipcc for gfx950 (cross-compiles on a GPU-less box)."""
    import subprocess

    subprocess.run(["make", "-C", str(_CSRC)] + ([] if verbose else ["-s"]), check=True)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"libskyhip.so does not export {name}")
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    lib.skyhip_abi_version.restype = C.c_int
    lib.skyhip_create.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.skyhip_create.restype = C.c_int
    lib.skyhip_destroy.argtypes = [vp]
    lib.skyhip_destroy.restype = None
    lib.skyhip_frame_bound.argtypes = [C.c_size_t]
    lib.skyhip_frame_bound.restype = C.c_size_t
    lib.skyhip_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.skyhip_host_alloc.restype = C.c_int
    lib.skyhip_host_free.argtypes = [vp, vp]
    lib.skyhip_host_free.restype = C.c_int
    lib.skyhip_host_register.argtypes = [vp, vp, C.c_size_t]
    lib.skyhip_host_register.restype = C.c_int
    lib.skyhip_host_unregister.argtypes = [vp, vp]
    lib.skyhip_host_unregister.restype = C.c_int
    lib.skyhip_process_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32]
    lib.skyhip_process_batch.restype = C.c_int
    lib.skyhip_process_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32]
    lib.skyhip_process_device.restype = C.c_int
    lib.skyhip_cdc_results.argtypes = [vp, C.c_int, vp, vp, C.c_size_t, vp, vp, vp]
    lib.skyhip_cdc_results.restype = C.c_int
    lib.skyhip_dedup_reset.argtypes = [vp]
    lib.skyhip_dedup_reset.restype = C.c_int
    lib.skyhip_dedup_literals.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.skyhip_dedup_literals.restype = C.c_int
    lib.skyhip_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.skyhip_dev_alloc.restype = C.c_int
    lib.skyhip_dev_free.argtypes = [vp, vp]
    lib.skyhip_dev_free.restype = C.c_int
    lib.skyhip_decompress_to_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.skyhip_decompress_to_device.restype = C.c_int
    lib.skyhip_gather_md5.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.skyhip_gather_md5.restype = C.c_int
    lib.skyhip_segment_md5_device.argtypes = [vp, C.c_size_t, vp, vp, vp]
    lib.skyhip_segment_md5_device.restype = C.c_int
    lib.skyhip_get_timing.argtypes = [vp, C.POINTER(Timing)]
    lib.skyhip_get_timing.restype = None
    lib.skyhip_reset_timing.argtypes = [vp]
    lib.skyhip_reset_timing.restype = None
    lib.skyhip_debug_fault.argtypes = [vp, C.c_long]
    lib.skyhip_debug_fault.restype = C.c_int
    lib.skyhip_debug_guard_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.skyhip_debug_guard_alloc.restype = C.c_int
    lib.skyhip_debug_guard_free.argtypes = [vp]
    lib.skyhip_debug_guard_free.restype = C.c_int
    lib.skyhip_debug_guard_probe.argtypes = [vp, vp]
    lib.skyhip_debug_guard_probe.restype = C.c_int
    lib.skyhip_selftest.argtypes = [vp]
    lib.skyhip_selftest.restype = C.c_int
    lib.skyhip_decompress_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.skyhip_decompress_device.restype = C.c_int
    lib.skyhip_decompress_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.skyhip_decompress_batch.restype = C.c_int
    lib.skyhip_decompress_batch_md5.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.skyhip_decompress_batch_md5.restype = C.c_int
    lib.skyhip_decompress_ms.argtypes = [vp, C.c_int]
    lib.skyhip_decompress_ms.restype = C.c_double
    lib.skyhip_debug_prof.argtypes = [vp, vp]
    lib.skyhip_debug_prof.restype = C.c_int
    lib.skyhip_strerror.argtypes = [C.c_int]
    lib.skyhip_strerror.restype = C.c_char_p
    lib.skyhip_last_hip_error.argtypes = [vp]
    lib.skyhip_last_hip_error.restype = C.c_char_p
    del u64p
    _lib = lib
    return lib
