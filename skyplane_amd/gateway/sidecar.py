"""Where the GPU stage keeps its by-products, derived from the one path the reference's ChunkStore knows
(``get_chunk_file_path``, skyplane/gateway/chunk_store.py:108-109) so that the operators, the sender patch and the
deferred receiver work with an UNMODIFIED reference ChunkStore:

    <chunk_dir>/<chunk_id>.chunk        raw bytes (the reference's file, untouched)
    <chunk_dir>/<chunk_id>.chunk.lz4f   LZ4 frame of the chunk (what goes on the wire when is_compressed=True)
    <chunk_dir>/<chunk_id>.chunk.md5    hex MD5 of the raw bytes
"""
from pathlib import Path

SIDECAR_SUFFIX = ".lz4f"
DIGEST_SUFFIX = ".md5"


def compressed_path(chunk_store, chunk_id: str) -> Path:
    p = Path(chunk_store.get_chunk_file_path(chunk_id))
    return p.with_name(p.name + SIDECAR_SUFFIX)


def digest_path(chunk_store, chunk_id: str) -> Path:
    p = Path(chunk_store.get_chunk_file_path(chunk_id))
    return p.with_name(p.name + DIGEST_SUFFIX)
