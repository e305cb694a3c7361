"""Regenerates tests/golden/recipe_v1.bin: one dedup-on-the-wire recipe (skyplane_amd/gateway/dedup_wire.py, version 1) with fixed contents, so that a
change of the payload sub-format cannot go unnoticed.  The format is this repo's own (the reference has no dedup): pinned against itself.
Run from the repo root: python tests/golden/make_recipe_golden.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import ref  # noqa: E402
from skyplane_amd.gateway import dedup_wire  # noqa: E402


def build():
    rng = np.random.default_rng(20260921)
    lens = np.array([1024, 4096, 1500, 16384, 2048], np.uint32)
    kinds = np.array([0, 1, 0, 0, 1], np.uint8)
    fps = rng.integers(0, 256, (5, 16), dtype=np.uint8)
    lit = (b"skyplane dedup on the wire " * 800)[: int(lens[kinds == 0].sum())]
    frame = ref.lz4f_compress_port(lit)          # the C restatement's greedy compressor: deterministic, independent of the liblz4 version
    return dedup_wire.encode_recipe(0x0123456789ABCDEF, 3, lens, kinds, fps, frame, len(lit)), lit


if __name__ == "__main__":
    blob, _ = build()
    (Path(__file__).parent / "recipe_v1.bin").write_bytes(blob)
    print("wrote recipe_v1.bin:", len(blob), "bytes")
