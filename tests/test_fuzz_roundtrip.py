"""Randomised parity: many structured-random inputs (copies at random distances, runs, literals, lengths around every
format threshold) through the compressor and both decoders.  CPU variant runs the shipping kernels under the emulator;
the gpu-marked variant runs the same generator through libskyhip.so."""
import hashlib

import numpy as np
import pytest

from oracle import ref
from skyplane_amd import synth


def make_case(rng: np.random.Generator, n: int) -> bytes:
    """Stream of ops: literal burst / copy from history (any distance, incl. overlapping) / run of one byte."""
    out = bytearray()
    alpha = int(rng.choice([2, 4, 16, 256]))
    while len(out) < n:
        k = rng.random()
        if k < 0.35 or len(out) < 8:
            m = int(rng.choice([1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 17, 40, 269, 270, 271, 600]))
            out += rng.integers(0, alpha, m, dtype=np.uint8).tobytes()
        elif k < 0.85:
            dist = int(min(len(out), rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 63, 64, 65, 255, 256, 4095, 4096, 65534, 65535, 65536, 70000])))
            ln = int(rng.choice([3, 4, 5, 7, 14, 15, 16, 17, 18, 19, 20, 31, 32, 33, 34, 35, 36, 63, 64, 65, 273, 274, 275, 1000, 5000]))
            start = len(out) - dist
            for i in range(ln):                    # byte-wise so that overlapping copies replicate
                out.append(out[start + i])
        else:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.choice([4, 5, 18, 19, 20, 300, 70000]))
    return bytes(out[:n])


SIZES = [13, 14, 17, 64, 65, 200, 4096, 65535, 65536, 65537, 65536 + 12, 65536 + 13, 131072, 150_001, 262_144 + 5]


def _cases(seed, count):
    rng = np.random.Generator(np.random.PCG64([synth.SEED_BASE, seed]))
    return [make_case(rng, int(rng.choice(SIZES))) for _ in range(count)]


def test_fuzz_emulated_kernels():
    from tests.emu import emulib

    chunks = _cases(1, 60)
    frames, md5s, _ = emulib.process(chunks)
    for d, f, m in zip(chunks, frames, md5s):
        assert m == hashlib.md5(d).digest()
        assert ref.lz4f_decompress(f, len(d)) == d
        dec, _ = ref.lz4f_decode(f, len(d), strict=True)
        assert dec == d
    rc, outs, status = emulib.decompress(frames, [len(d) for d in chunks])
    assert rc == 0 and outs == chunks
    rc, outs, status = emulib.decompress([ref.lz4f_compress(d) for d in chunks], [len(d) for d in chunks])   # linked frames
    assert rc == 0 and outs == chunks


@pytest.mark.gpu
def test_fuzz_gpu():
    import torch

    torch.cuda.init()
    from skyplane_amd import hip_ops

    with hip_ops.SkyHipContext(device_id=0, max_chunk_bytes=1 << 20, max_batch=16) as ctx:
        for seed in (2, 3, 4):
            chunks = _cases(seed, 150)
            res = ctx.process_batch(chunks, flags=hip_ops.F_LZ4 | hip_ops.F_MD5 | hip_ops.F_CDC)
            for d, r in zip(chunks, res):
                assert r.md5 == hashlib.md5(d).digest()
                assert ref.lz4f_decompress(r.frame, len(d)) == d
                assert (r.cuts == ref.gear_cdc(d)).all()
            assert ctx.decompress_batch([r.frame for r in res], [len(d) for d in chunks]) == chunks
            assert ctx.decompress_batch([ref.lz4f_compress(d) for d in chunks], [len(d) for d in chunks]) == chunks
