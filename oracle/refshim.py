"""Import shim that lets the REAL reference gateway modules run in the build container.  TEST INFRASTRUCTURE ONLY.

`import skyplane` fails offline (skyplane/__init__.py pulls cloud SDKs: cryptography, boto3, paramiko ...).  The
gateway modules on the hot path do not need them, so a scratch directory is put on sys.path that holds
  skyplane/__init__.py   -- only `__path__ = ["/root/reference/skyplane"]`: sub-modules load from the read-only tree
  lz4/frame.py           -- compress()/decompress() over the system liblz4 with python-lz4's default preferences
                            (oracle/ref.py; python-lz4 4.3.2 itself is not installed)
  nacl/secret.py, OpenSSL/__init__.py -- empty stand-ins (TLS and e2ee stay off)
(SURVEY.md 8c describes the same shim.)  Nothing from /root/reference is copied; the directory lives in a tmp dir.
The GPU box has no /root/reference: there the tree staged by oracle/stage_reference.py is used; callers still check `available()`.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

_REPO = Path(__file__).resolve().parents[1]
# the read-only reference tree (build container) or, on the GPU box, the copy oracle/stage_reference.py staged under oracle/_ref/
# (git-ignored, travels with gpurun)
REFERENCE = (Path(os.environ["SKY_REFERENCE_ROOT"]) if os.environ.get("SKY_REFERENCE_ROOT") else
             Path("/root/reference") if Path("/root/reference/skyplane/chunk.py").is_file() else _REPO / "oracle" / "_ref")


def available() -> bool:
    return (REFERENCE / "skyplane" / "chunk.py").is_file()


_LZ4_FRAME = '''
import sys
sys.path.insert(0, {repo!r})
from oracle import ref as _ref


def compress(data, **kw):
    """lz4.frame.compress(data) with python-lz4 defaults (block-linked 64 KiB blocks, store_size=True)."""
    return _ref.lz4f_compress(bytes(data), store_size=kw.get("store_size", True), block_linked=kw.get("block_linked", True))


def decompress(data, **kw):
    """lz4.frame.decompress(data): LZ4F_decompress over the whole frame; raises on malformed input."""
    data = bytes(data)
    cap = None
    if len(data) >= 15 and data[4] & 0x08:
        cap = int.from_bytes(data[6:14], "little")
    if cap is None:
        cap = 255 * len(data) + 65536
    return _ref.lz4f_decompress(data, cap)
'''


def install(scratch: Path) -> Path:
    """Create the shim under `scratch`, put it first on sys.path, point the reference's config at scratch."""
    if not available():
        raise RuntimeError("reference tree not present")
    scratch = Path(scratch)
    (scratch / "skyplane").mkdir(parents=True, exist_ok=True)
    (scratch / "skyplane" / "__init__.py").write_text(f'__path__ = ["{REFERENCE / "skyplane"}"]\n')
    (scratch / "lz4").mkdir(exist_ok=True)
    (scratch / "lz4" / "__init__.py").write_text("")
    (scratch / "lz4" / "frame.py").write_text(_LZ4_FRAME.format(repo=str(_REPO)))
    (scratch / "nacl").mkdir(exist_ok=True)
    (scratch / "nacl" / "__init__.py").write_text("")
    (scratch / "nacl" / "secret.py").write_text("class SecretBox:\n    def __init__(self, *a, **k):\n        raise RuntimeError('e2ee is off in this harness')\n")
    (scratch / "OpenSSL").mkdir(exist_ok=True)
    (scratch / "OpenSSL" / "__init__.py").write_text("crypto = None\n")
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    os.environ["SKYPLANE_CONFIG"] = str(scratch / "config")
    os.environ["HOME"] = str(scratch)
    if str(scratch) not in sys.path:
        sys.path.insert(0, str(scratch))
    return scratch
