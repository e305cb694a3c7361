"""Deterministic synthetic byte streams for tests and bench.

There is no Silesia corpus and no network on the build or GPU boxes, so config 2's
"Silesia-corpus-replicated" stream is replaced by a frozen synthetic stand-in: segments of
64 KiB-1 MiB drawn from six content classes whose mix lands near Silesia's LZ4-fast ratio (~2.1).
Everything is generated from ``numpy.random.Generator(PCG64(seed))`` with
``seed = 0x5EED0000 + config_id`` so CPU oracle and GPU runs see identical bytes.
"""
from __future__ import annotations

import numpy as np

MB = 1024 * 1024
CHUNK_BYTES = 8 * MB  # BASELINE.json: 8 MiB chunks
SEED_BASE = 0x5EED0000

CLASSES = ("text", "records", "numeric", "binary", "sparse", "random")
# SURVEY §8d item 2's six classes; weights tuned (measured with the system liblz4 1.9.3, python-lz4
# default preferences) so the mix lands near Silesia's LZ4-fast ratio of ~2.1
SILESIA_LIKE_WEIGHTS = {"text": 0.33, "records": 0.22, "numeric": 0.08, "binary": 0.15, "sparse": 0.12, "random": 0.10}


def rng_for(config_id: int, stream: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([SEED_BASE + config_id, stream]))


# --------------------------------------------------------------------------------------------
# dictionary-gather generator: the workhorse for text / records / binary classes
# --------------------------------------------------------------------------------------------
def _make_dict(rng: np.random.Generator, n_words: int, min_len: int, max_len: int, alphabet: np.ndarray, suffix: bytes = b""):
    lens = rng.integers(min_len, max_len + 1, n_words)
    total = int(lens.sum())
    letters = alphabet[rng.integers(0, alphabet.size, total)]
    sfx = np.frombuffer(suffix, np.uint8)
    if sfx.size:
        # append suffix to every word
        out = np.empty(total + sfx.size * n_words, np.uint8)
        offs = np.zeros(n_words + 1, np.int64)
        np.cumsum(lens + sfx.size, out=offs[1:])
        src_off = np.zeros(n_words + 1, np.int64)
        np.cumsum(lens, out=src_off[1:])
        idx = np.arange(out.size, dtype=np.int64)
        w = np.searchsorted(offs, idx, side="right") - 1
        pos = idx - offs[w]
        is_sfx = pos >= lens[w]
        out[~is_sfx] = letters[(src_off[w] + pos)[~is_sfx]]
        out[is_sfx] = sfx[(pos - lens[w])[is_sfx]]
        return out, offs[:-1], (lens + sfx.size)
    offs = np.zeros(n_words + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    return letters, offs[:-1], lens


def _gather_words(blob: np.ndarray, woff: np.ndarray, wlen: np.ndarray, idx: np.ndarray, nbytes: int) -> np.ndarray:
    """Concatenate dictionary words idx[0], idx[1], ... and cut to nbytes (vectorised)."""
    lens = wlen[idx]
    ends = np.cumsum(lens)
    k = int(np.searchsorted(ends, nbytes, side="left")) + 1
    idx, lens, ends = idx[:k], lens[:k], ends[:k]
    total = int(ends[-1])
    starts = ends - lens
    src = np.repeat(woff[idx] - starts, lens) + np.arange(total, dtype=np.int64)
    return blob[src][:nbytes]


def _zipf_idx(rng: np.random.Generator, n_words: int, count: int, a: float = 1.15) -> np.ndarray:
    p = 1.0 / np.arange(1, n_words + 1) ** a
    p /= p.sum()
    cdf = np.cumsum(p)
    return np.searchsorted(cdf, rng.random(count), side="left").clip(0, n_words - 1)


_LOWER = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", np.uint8)


def gen_text(rng: np.random.Generator, nbytes: int) -> np.ndarray:
    """English-like word text: Zipf over a 4k-word dictionary, space separated."""
    blob, woff, wlen = _make_dict(rng, 4096, 2, 10, _LOWER[:20], suffix=b" ")
    idx = _zipf_idx(rng, 4096, nbytes // 3 + 16)
    return _gather_words(blob, woff, wlen, idx, nbytes)


def gen_records(rng: np.random.Generator, nbytes: int) -> np.ndarray:
    """Structured XML-like records: repeated tags around short variable fields."""
    tags = [b"<row id=\"", b"\" ts=\"2026-09-", b"\"><name>", b"</name><value>", b"</value><status>", b"</status></row>\n"]
    fields_blob, foff, flen = _make_dict(rng, 2048, 1, 9, np.frombuffer(b"0123456789abcdef", np.uint8))
    tag_blob = np.frombuffer(b"".join(tags), np.uint8)
    toff = np.cumsum([0] + [len(t) for t in tags[:-1]])
    blob = np.concatenate([tag_blob, fields_blob])
    woff = np.concatenate([toff, foff + tag_blob.size]).astype(np.int64)
    wlen = np.concatenate([[len(t) for t in tags], flen]).astype(np.int64)
    nrec = nbytes // 60 + 8
    fidx = _zipf_idx(rng, 2048, nrec * len(tags), a=0.9) + len(tags)
    idx = np.empty(nrec * len(tags) * 2, np.int64)
    idx[0::2] = np.tile(np.arange(len(tags)), nrec)
    idx[1::2] = fidx
    return _gather_words(blob, woff, wlen, idx, nbytes)


def gen_numeric(rng: np.random.Generator, nbytes: int) -> np.ndarray:
    """Small-alphabet data: half DNA-like (ACGT), half ASCII decimal columns."""
    half = nbytes // 2
    dna = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, half)]
    vals = rng.integers(0, 100000, (nbytes - half) // 6 + 2)
    digits = np.empty((vals.size, 6), np.uint8)
    v = vals.copy()
    for c in range(4, -1, -1):
        digits[:, c] = 48 + v % 10
        v //= 10
    digits[:, 5] = 44  # ','
    return np.concatenate([dna, digits.reshape(-1)[: nbytes - half]])


def gen_binary(rng: np.random.Generator, nbytes: int) -> np.ndarray:
    """x86-like binary: Zipf over a dictionary of short opcode strings with random operands."""
    blob, woff, wlen = _make_dict(rng, 1500, 3, 10, np.arange(256, dtype=np.uint8))
    idx = _zipf_idx(rng, 1500, nbytes // 3 + 16, a=1.1)
    out = _gather_words(blob, woff, wlen, idx, nbytes).copy()
    # sprinkle random immediates (1 byte in 25)
    m = rng.random(nbytes) < 0.04
    out[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
    return out


def gen_sparse(rng: np.random.Generator, nbytes: int) -> np.ndarray:
    """Runs / sparse zeros: zero background, ~1 % random bytes, some constant runs."""
    out = np.zeros(nbytes, np.uint8)
    m = rng.random(nbytes) < 0.01
    out[m] = rng.integers(1, 256, int(m.sum()), dtype=np.uint8)
    nruns = max(1, nbytes // 8192)
    starts = rng.integers(0, max(1, nbytes - 512), nruns)
    vals = rng.integers(0, 256, nruns, dtype=np.uint8)
    for s, v in zip(starts[:256], vals[:256]):
        out[s:s + 300] = v
    return out


def gen_random(rng: np.random.Generator, nbytes: int) -> np.ndarray:
    return rng.integers(0, 256, nbytes, dtype=np.uint8)


_GEN = {"text": gen_text, "records": gen_records, "numeric": gen_numeric, "binary": gen_binary, "sparse": gen_sparse, "random": gen_random}


def gen_class(name: str, nbytes: int, rng: np.random.Generator) -> np.ndarray:
    out = _GEN[name](rng, nbytes)
    assert out.dtype == np.uint8 and out.size == nbytes, (name, out.size, nbytes)
    return out


def silesia_like(nbytes: int, config_id: int = 2, weights=None, seg_min: int = 64 * 1024, seg_max: int = MB) -> np.ndarray:
    """Config-2 stand-in: mixed-class segments of 64 KiB-1 MiB."""
    weights = weights or SILESIA_LIKE_WEIGHTS
    rng = rng_for(config_id)
    names = list(weights)
    p = np.array([weights[k] for k in names], float)
    p /= p.sum()
    out = np.empty(nbytes, np.uint8)
    pos = 0
    while pos < nbytes:
        seg = int(rng.integers(seg_min, seg_max + 1))
        seg = min(seg, nbytes - pos)
        cls = names[int(rng.choice(len(names), p=p))]
        out[pos:pos + seg] = gen_class(cls, seg, rng)
        pos += seg
    return out


def mixed_chunks(n_chunks: int, chunk_bytes: int = CHUNK_BYTES, config_id: int = 4) -> np.ndarray:
    """Config-4 stand-in: per-chunk class drawn uniformly from {random, text, records+binary mix, sparse}."""
    rng = rng_for(config_id)
    out = np.empty((n_chunks, chunk_bytes), np.uint8)
    for i in range(n_chunks):
        k = int(rng.integers(0, 4))
        if k == 0:
            out[i] = gen_random(rng, chunk_bytes)
        elif k == 1:
            out[i] = gen_text(rng, chunk_bytes)
        elif k == 2:
            h = chunk_bytes // 2
            out[i, :h] = gen_records(rng, h)
            out[i, h:] = gen_binary(rng, chunk_bytes - h)
        else:
            out[i] = gen_sparse(rng, chunk_bytes)
    return out


def dedup_stream(nbytes: int, dup_fraction: float = 0.5, config_id: int = 3, span_min: int = 8 * 1024, span_max: int = 64 * 1024) -> np.ndarray:
    """Config-3 stand-in: a stream where ~dup_fraction of the spans are copies of EARLIER spans
    pasted at non-aligned offsets (content-defined cuts must find them; fixed cuts would not)."""
    rng = rng_for(config_id)
    out = np.empty(nbytes, np.uint8)
    pos = 0
    history = []  # (start, length) of fresh spans
    while pos < nbytes:
        span = int(rng.integers(span_min, span_max + 1))
        span = min(span, nbytes - pos)
        if history and rng.random() < dup_fraction:
            s, l = history[int(rng.integers(0, len(history)))]      # the WHOLE earlier span is pasted (up to the end of the stream)
            l = min(l, nbytes - pos)
            out[pos:pos + l] = out[s:s + l]
            pos += l
        else:
            out[pos:pos + span] = gen_random(rng, span) if rng.random() < 0.5 else gen_text(rng, span)
            history.append((pos, span))
            pos += span
    return out


def small_cases(config_id: int = 0):
    """Edge-case inputs the parity tests sweep (name -> bytes): empty, tiny, ragged, block-boundary."""
    rng = rng_for(config_id, 99)
    cases = {
        "empty": b"",
        "one": b"x",
        "twelve": b"abcdabcdabcd",
        "thirteen": b"aaaaaaaaaaaaa",
        "zeros_100": bytes(100),
        "zeros_64k": bytes(65536),
        "zeros_64k_p1": bytes(65537),
        "abc_run": (b"abc" * 30000)[:70001],
        "text_5000": gen_text(rng, 5000).tobytes(),
        "rand_4096": gen_random(rng, 4096).tobytes(),
        "rand_65536": gen_random(rng, 65536).tobytes(),
        "rand_65535": gen_random(rng, 65535).tobytes(),
        "mixed_200k": silesia_like(200_000, config_id=7, seg_min=3000, seg_max=40000).tobytes(),
        "records_131077": gen_records(rng, 131077).tobytes(),
        "long_match_tail": (gen_random(rng, 300).tobytes() * 400)[:100_003],
        "period_1_2_3": bytes(70) + b"ab" * 50 + b"xyz" * 40 + bytes([7]) * 19,
    }
    return cases
