"""LZ4 frames built sequence by sequence (test infrastructure): what no compressor emits on its own but every decoder must take.

The batch decoder of skyplane_amd/csrc/lz4d_kernel.inc has paths that depend on the SHAPE of the sequence stream -- length fields with one or more
extension bytes, 64 three-byte sequences in one 192-byte window, batches above the LDS staging size, matches that begin before a batch and end inside it,
sequences that do not fit a window -- and the frames of liblz4 and of this library's compressor reach some of them rarely or never.  These builders make
blocks out of explicit (literals, offset, match length) lists; the expected output comes from a ten-line Python decoder and is cross-checked with liblz4
(what lz4.frame.decompress at skyplane/gateway/operators/gateway_receiver.py:195-201 calls)."""
import numpy as np

from oracle import ref

BLOCK = 65536


def _ext(x):
    out = bytearray()
    x -= 15
    while x >= 255:
        out.append(255); x -= 255
    out.append(x)
    return bytes(out)


def encode_block(seqs, last_literals, history=b""):
    """seqs: [(literals: bytes, offset: int, match_len >= 4)], then the block's final literal-only sequence.  Returns (compressed, decoded).
    history: what a block-linked frame's earlier blocks decoded to (a match may reach up to 65535 bytes back into it)."""
    comp, out = bytearray(), bytearray(history[-65535:])
    h = len(out)
    for lit, off, ml in seqs:
        assert ml >= 4 and 1 <= off <= len(out) + len(lit) and off <= 65535
        comp.append((min(len(lit), 15) << 4) | min(ml - 4, 15))
        if len(lit) >= 15:
            comp += _ext(len(lit))
        comp += lit
        out += lit
        comp += bytes((off & 255, off >> 8))
        if ml - 4 >= 15:
            comp += _ext(ml - 4)
        for _ in range(ml):
            out.append(out[-off])
    comp.append(min(len(last_literals), 15) << 4)
    if len(last_literals) >= 15:
        comp += _ext(len(last_literals))
    comp += last_literals
    out += last_literals
    return bytes(comp), bytes(out[h:])


def frame_of(blocks, sized=True, linked=False):
    """blocks: [(compressed, decoded)] with every block but the last decoding to exactly 64 KiB.  No checksums; block-independent unless `linked`."""
    total = sum(len(d) for _, d in blocks)
    flg = 0x40 | (0 if linked else 0x20) | (0x08 if sized else 0)
    desc = bytes((flg, 0x40)) + (total.to_bytes(8, "little") if sized else b"")
    hdr = b"\x04\x22\x4d\x18" + desc + bytes(((ref.xxh32(desc) >> 8) & 0xFF,))
    body = b"".join(len(c).to_bytes(4, "little") + c for c, _ in blocks)
    return hdr + body + bytes(4), b"".join(d for _, d in blocks)


def _fill_block(rng, gen, history=b""):
    """Run `gen(state)` -> (literals, offset, match_len) until the block holds exactly 64 KiB; the last 5+ bytes are literals (the format's end rule)."""
    seqs, n = [], 0
    while True:
        lit, off, ml = gen(n)
        if n + len(lit) + ml > BLOCK - 16:
            break
        seqs.append((lit, off, ml)); n += len(lit) + ml
    tail = rng.integers(0, 256, BLOCK - n, dtype=np.uint8).tobytes()
    return encode_block(seqs, tail, history)


def crafted_frames(seed=0):
    """{name: (frame, decoded)}"""
    rng = np.random.default_rng(seed)

    def rnd(k):
        return rng.integers(0, 256, k, dtype=np.uint8).tobytes()

    cases = {}
    # 1. nothing but three-byte sequences: 64 of them start in every 192-byte window (the batch holds 63)
    cases["three_byte_sequences"] = frame_of([_fill_block(rng, lambda n: (rnd(40) if n == 0 else b"", int(rng.integers(1, 41)), 4))])
    # 2. every length around the nibble and extension-byte boundaries, literals and matches
    edge = [0, 1, 14, 15, 16, 17, 254 + 15, 255 + 15, 256 + 15, 300, 509 + 15, 510 + 15, 600]
    medge = [4, 5, 18, 19, 20, 21, 254 + 19, 255 + 19, 256 + 19, 400, 510 + 19, 1000]
    it = iter([(rnd(a), None, b) for a in edge for b in medge])

    def g2(n):
        try:
            lit, _, ml = next(it)
        except StopIteration:
            lit, ml = rnd(int(rng.integers(0, 4))), 4
        return lit, int(rng.integers(1, n + len(lit) + 1)) if n + len(lit) else 1, ml
    cases["length_field_edges"] = frame_of([_fill_block(rng, lambda n: (rnd(20), 1, 4) if n == 0 else g2(n))])
    # 3. batches above the LDS staging size: many matches of ~270 bytes in one window
    cases["fat_batches"] = frame_of([_fill_block(rng, lambda n: (rnd(30) if n == 0 else b"", int(rng.integers(1, max(2, min(n, 5000)))) if n else 1, int(rng.integers(200, 274))))])
    # 4. run-length style: short offsets right at the start of batches (a match that begins before the batch and ends in it), overlapping copies of every period
    cases["short_offsets"] = frame_of([_fill_block(rng, lambda n: (rnd(int(rng.integers(1, 4))) if n == 0 or rng.random() < 0.3 else b"", int(rng.integers(1, min(n, 12) + 1)) if n else 1,
                                                              int(rng.choice([4, 7, 16, 33, 64, 65, 100, 300])))) for _ in range(2)])
    # 5. everything at random, two full blocks and a short one
    def g5(n):
        r = rng.random()
        lit = rnd(0 if r < 0.5 else int(rng.integers(1, 15)) if r < 0.85 else int(rng.choice([15, 16, 40, 269, 270, 271, 700])))
        avail = n + len(lit)
        if avail == 0:
            lit = rnd(8); avail = 8
        off = int(rng.integers(1, min(avail, 65535) + 1)) if rng.random() < 0.7 else int(rng.integers(1, min(avail, 24) + 1))
        ml = int(rng.choice([4, 5, 8, 12, 18, 19, 20, 33, 70, 273, 274, 280, 2000], p=[.25, .15, .15, .1, .05, .05, .05, .05, .05, .03, .03, .02, .02]))
        return lit, off, ml
    blocks = [_fill_block(rng, g5) for _ in range(2)]
    seqs, n = [], 0
    for _ in range(200):
        lit, off, ml = g5(n); seqs.append((lit, off, ml)); n += len(lit) + ml
    blocks.append(encode_block(seqs, rnd(9)))
    cases["random_sequences"] = frame_of(blocks)
    cases["random_sequences_unsized"] = frame_of(blocks, sized=False)
    # 5b. the same in a block-LINKED frame (what lz4.frame.compress makes by default): offsets reach into the previous block
    def g5l(hist):
        def g(n):
            lit, off, ml = g5(n)
            if hist and rng.random() < 0.5:
                off = int(rng.integers(1, min(n + len(lit) + hist, 65535) + 1))
            return lit, off, ml
        return g
    lblocks, hist = [], b""
    for _ in range(3):
        c, d = _fill_block(rng, g5l(len(hist)), hist)
        lblocks.append((c, d)); hist = (hist + d)[-65535:]
    seqs, n = [], 0
    g = g5l(len(hist))
    for _ in range(100):
        lit, off, ml = g(n); seqs.append((lit, off, ml)); n += len(lit) + ml
    lblocks.append(encode_block(seqs, rnd(11), hist))
    cases["random_sequences_linked"] = frame_of(lblocks, linked=True)
    # 6. one long literal run and one long match that do not fit a window, and a block of literals only
    cases["long_fields"] = frame_of([encode_block([(rnd(5000), 4000, 30000), (b"", 1, 20000), (rnd(3), 25000, 4)], rnd(BLOCK - 5000 - 30000 - 20000 - 3 - 4)),
                                     encode_block([], rnd(777))])
    for name, (f, d) in cases.items():
        assert ref.lz4f_decompress(f, len(d)) == d, name        # liblz4 agrees with the Python decoder above
    return cases
