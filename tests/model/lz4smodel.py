"""ctypes binding of tests/model/lz4s_model.c -- the sequential restatement of the slice-parallel LZ4 parse.

TEST INFRASTRUCTURE ONLY.  The GPU kernel (skyplane_amd/csrc/lz4s_kernel.inc) is deterministic, so its compressed blocks
must equal the model's byte for byte; `frame_blocks` splits one of this library's frames back into its blocks so that a
test can make that comparison through the C ABI."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BLOCK = 65536


def _lib():
    so = HERE / "_build" / "liblz4smodel.so"
    src = HERE / "lz4s_model.c"
    spec = HERE.parents[1] / "skyplane_amd" / "csrc" / "lz4s_spec.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, spec.stat().st_mtime):
        subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    lib = C.CDLL(str(so))
    lib.lz4s_model_block.restype = C.c_uint32
    lib.lz4s_model_block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return lib


_L = None


def block(data) -> bytes:
    """Compressed form of one block (<= 64 KiB) as the kernel must produce it (may be longer than the input)."""
    global _L
    _L = _L or _lib()
    a = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    assert a.size <= BLOCK
    dst = np.empty(a.size + a.size // 255 + 64, np.uint8)
    cs = _L.lz4s_model_block(a.ctypes.data if a.size else None, a.size, dst.ctypes.data, None)
    return dst[:cs].tobytes()


def frame_blocks(frame):
    """Split an LZ4 frame (15-byte header with content size, no checksums) into (stored_raw, payload) per block."""
    f = bytes(frame)
    assert f[:4] == b"\x04\x22\x4d\x18" and f[4] & 0x08, "not one of our frames"
    pos, out = 15, []
    while True:
        w = int.from_bytes(f[pos:pos + 4], "little")
        pos += 4
        if w == 0:
            break
        size = w & 0x7FFFFFFF
        out.append((bool(w >> 31), f[pos:pos + size]))
        pos += size
    assert pos == len(f)
    return out


def check_frame(raw, frame):
    """Every block of `frame` is what the model produces for the corresponding 64 KiB of `raw` (stored raw exactly when
    the model's output does not shrink it)."""
    raw = bytes(raw)
    blocks = frame_blocks(frame)
    assert len(blocks) == (len(raw) + BLOCK - 1) // BLOCK
    for i, (is_raw, payload) in enumerate(blocks):
        src = raw[i * BLOCK:(i + 1) * BLOCK]
        want = block(src)
        if len(want) < len(src):
            assert not is_raw and payload == want, f"block {i}: compressed bytes differ from the model ({len(payload)} vs {len(want)} bytes)"
        else:
            assert is_raw and payload == src, f"block {i}: expected a stored block"
