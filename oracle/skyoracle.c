/*
 * skyoracle.c -- CPU ORACLE for the Skyplane gateway compress+hash stage.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the
 * product path (skyplane_amd/, libskyhip.so) never links, imports or calls it.
 *
 * What it restates (reference = skyplane-project/skyplane v0.3.2, 100 % Python;
 * the arithmetic of the hot path lives in third-party wheels that are NOT
 * under /root/reference):
 *   - lz4 4.3.2 (python-lz4, poetry.lock:1540-1541) -> liblz4 LZ4F_*:
 *       call sites skyplane/gateway/operators/gateway_operator.py:358-361
 *       (lz4.frame.compress) and gateway_receiver.py:195-201
 *       (lz4.frame.decompress).  Restated here from the published LZ4 Frame
 *       format (v1.6.x) and LZ4 Block format documents: frame parser + block
 *       decoder (the acceptance side) and a greedy single-probe block
 *       compressor (a "port" used only as a timed CPU baseline fallback).
 *   - CPython hashlib.md5 (OpenSSL): call sites
 *       skyplane/obj_store/s3_interface.py:181-192 (+ gcs/azure/cos/scp
 *       siblings), requested at gateway_operator.py:555-565.  Restated from
 *       RFC 1321.
 *   - skyplane/chunk.py:95-167 WireProtocolHeader.to_bytes/from_bytes.
 *   - Gear CDC / fingerprints / dedup: NOT in the reference;
 *     the functions below ARE the frozen specification ("parity unpinned").
 *
 * Pinning: tests/test_oracle.py checks every function against the golden
 * vectors in tests/golden/ (RFC 1321 A.5 suite, hashlib digests, frames made
 * by the system liblz4 1.9.3 through the reference's own call pattern, the
 * reference chunk.py's header bytes) -- see tests/golden/make_golden.py.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define SKO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* MD5 (RFC 1321).  Mirrors hashlib.md5().update(b)...digest() as used */
/* by s3_interface.py:181-192; update granularity does not matter.     */
/* ------------------------------------------------------------------ */
static const uint32_t MD5_K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t MD5_S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                  5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

static inline uint32_t rotl32(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
static inline uint32_t rd32le(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline void wr32le(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

static void md5_block(uint32_t st[4], const uint8_t* blk) {
    uint32_t M[16];
    for (int i = 0; i < 16; i++) M[i] = rd32le(blk + 4 * i);
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
    for (int i = 0; i < 64; i++) {
        uint32_t f; int g;
        if (i < 16)      { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d;          g = (3 * i + 5) & 15; }
        else             { f = c ^ (b | ~d);       g = (7 * i) & 15; }
        uint32_t t = d; d = c; c = b;
        b = b + rotl32(a + f + MD5_K[i] + M[g], MD5_S[i]);
        a = t;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

SKO_API void sko_md5(const uint8_t* data, size_t len, uint8_t out[16]) {
    uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    size_t nfull = len / 64;
    for (size_t i = 0; i < nfull; i++) md5_block(st, data + 64 * i);
    uint8_t tail[128];
    size_t rem = len - 64 * nfull;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, data + 64 * nfull, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8u;
    for (int i = 0; i < 8; i++) tail[tl - 8 + i] = (uint8_t)(bits >> (8 * i));
    md5_block(st, tail);
    if (tl == 128) md5_block(st, tail + 64);
    for (int i = 0; i < 4; i++) wr32le(out + 4 * i, st[i]);
}

/* ------------------------------------------------------------------ */
/* XXH32 (needed for the LZ4 frame header checksum byte HC).           */
/* ------------------------------------------------------------------ */
#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u
SKO_API uint32_t sko_xxh32(const uint8_t* p, size_t len, uint32_t seed) {
    const uint8_t* end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do {
            v1 = rotl32(v1 + rd32le(p) * XP2, 13) * XP1; p += 4;
            v2 = rotl32(v2 + rd32le(p) * XP2, 13) * XP1; p += 4;
            v3 = rotl32(v3 + rd32le(p) * XP2, 13) * XP1; p += 4;
            v4 = rotl32(v4 + rd32le(p) * XP2, 13) * XP1; p += 4;
        } while (p + 16 <= end);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + XP5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32le(p) * XP3, 17) * XP4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * XP5, 11) * XP1; p++; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

/* ------------------------------------------------------------------ */
/* LZ4 block decoder (LZ4 Block format).  strict != 0 additionally     */
/* enforces the encoder-side end-of-block rules a conformant block     */
/* must satisfy (last sequence literals-only, last 5 bytes literal,    */
/* last match starts >= 12 bytes before the end).                      */
/* Returns decoded size, or a negative error code.                     */
/* ------------------------------------------------------------------ */
enum {
    SKO_E_TRUNC = -1, SKO_E_OFFSET = -2, SKO_E_OVERRUN = -3, SKO_E_ENDRULE = -4, SKO_E_MAGIC = -5,
    SKO_E_FLG = -6, SKO_E_HC = -7, SKO_E_BLOCKSIZE = -8, SKO_E_CONTENTSIZE = -9, SKO_E_TRAILING = -10,
    SKO_E_CAP = -11
};

SKO_API long sko_lz4_block_decode(const uint8_t* src, size_t slen, uint8_t* dst, size_t cap, int strict,
                                  const uint8_t* dict, size_t dict_len) {
    size_t ip = 0, op = 0;
    size_t last_match_start = (size_t)-1, last_match_end = 0;
    if (slen == 0) return SKO_E_TRUNC;
    for (;;) {
        if (ip >= slen) return SKO_E_TRUNC;
        unsigned tok = src[ip++];
        size_t lit = tok >> 4;
        if (lit == 15) {
            unsigned b;
            do { if (ip >= slen) return SKO_E_TRUNC; b = src[ip++]; lit += b; } while (b == 255);
        }
        if (ip + lit > slen) return SKO_E_TRUNC;
        if (op + lit > cap) return SKO_E_OVERRUN;
        memcpy(dst + op, src + ip, lit);
        ip += lit; op += lit;
        if (ip == slen) {                      /* last sequence: literals only */
            if (strict) {
                if ((tok & 15) != 0) return SKO_E_ENDRULE;
                if (last_match_start != (size_t)-1) {
                    if (op - last_match_end < 5) return SKO_E_ENDRULE;
                    if (op - last_match_start < 12) return SKO_E_ENDRULE;
                }
            }
            return (long)op;
        }
        if (ip + 2 > slen) return SKO_E_TRUNC;
        size_t off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        if (off == 0) return SKO_E_OFFSET;
        size_t ml = tok & 15;
        if (ml == 15) {
            unsigned b;
            do { if (ip >= slen) return SKO_E_TRUNC; b = src[ip++]; ml += b; } while (b == 255);
        }
        ml += 4;
        if (off > op + dict_len) return SKO_E_OFFSET;
        if (op + ml > cap) return SKO_E_OVERRUN;
        last_match_start = op;
        for (size_t k = 0; k < ml; k++) {      /* byte-wise: overlap (off < ml) is legal */
            size_t from = op + k;
            dst[op + k] = (from >= off) ? dst[from - off] : dict[dict_len - (off - from)];
        }
        op += ml;
        last_match_end = op;
    }
}

/* ------------------------------------------------------------------ */
/* LZ4 frame decoder == what lz4.frame.decompress does at              */
/* gateway_receiver.py:195-201.  Handles both linked and independent   */
/* blocks, stored blocks, content size, optional checksums.            */
/* info[0]=FLG info[1]=BD info[2]=#blocks info[3]=#stored(raw) blocks  */
/* Returns decoded size or negative error.                             */
/* ------------------------------------------------------------------ */
SKO_API long sko_lz4f_decompress(const uint8_t* f, size_t flen, uint8_t* dst, size_t cap, int strict,
                                 uint32_t info[4]) {
    if (flen < 7) return SKO_E_TRUNC;
    if (rd32le(f) != 0x184D2204u) return SKO_E_MAGIC;
    unsigned flg = f[4], bd = f[5];
    if ((flg >> 6) != 1) return SKO_E_FLG;          /* version must be 01 */
    if (flg & 0x02) return SKO_E_FLG;               /* reserved bit */
    if (bd & 0x8F) return SKO_E_FLG;                /* reserved bits */
    int indep = (flg >> 5) & 1, bsum = (flg >> 4) & 1, csize = (flg >> 3) & 1, csum = (flg >> 2) & 1, dictid = flg & 1;
    unsigned bid = (bd >> 4) & 7;
    if (bid < 4) return SKO_E_FLG;
    size_t bmax = (size_t)1 << (8 + 2 * bid);       /* 4:64K 5:256K 6:1M 7:4M */
    size_t hp = 6;
    uint64_t content = 0;
    if (csize) {
        if (flen < hp + 8) return SKO_E_TRUNC;
        for (int i = 0; i < 8; i++) content |= (uint64_t)f[hp + i] << (8 * i);
        hp += 8;
    }
    if (dictid) hp += 4;
    if (flen < hp + 1) return SKO_E_TRUNC;
    unsigned hc = (sko_xxh32(f + 4, hp - 4, 0) >> 8) & 0xFF;
    if (f[hp] != hc) return SKO_E_HC;
    size_t ip = hp + 1, op = 0;
    uint32_t nblk = 0, nraw = 0;
    for (;;) {
        if (ip + 4 > flen) return SKO_E_TRUNC;
        uint32_t bs = rd32le(f + ip); ip += 4;
        if (bs == 0) break;                          /* EndMark */
        int raw = (bs >> 31) & 1;
        size_t sz = bs & 0x7FFFFFFFu;
        if (sz > bmax) return SKO_E_BLOCKSIZE;
        if (ip + sz > flen) return SKO_E_TRUNC;
        if (raw) {
            if (op + sz > cap) return SKO_E_CAP;
            memcpy(dst + op, f + ip, sz);
            op += sz; nraw++;
        } else {
            size_t room = cap - op; if (room > bmax) room = bmax;
            size_t dl = indep ? 0 : (op > 65536 ? 65536 : op);
            long r = sko_lz4_block_decode(f + ip, sz, dst + op, room, strict, dst + op - dl, dl);
            if (r < 0) return r;
            op += (size_t)r;
        }
        ip += sz;
        if (bsum) {
            if (ip + 4 > flen) return SKO_E_TRUNC;
            if (rd32le(f + ip) != sko_xxh32(f + ip - sz, sz, 0)) return SKO_E_HC;
            ip += 4;
        }
        nblk++;
    }
    if (csum) {
        if (ip + 4 > flen) return SKO_E_TRUNC;
        if (rd32le(f + ip) != sko_xxh32(dst, op, 0)) return SKO_E_HC;
        ip += 4;
    }
    if (csize && content != op) return SKO_E_CONTENTSIZE;
    if (ip != flen) return SKO_E_TRAILING;
    if (info) { info[0] = flg; info[1] = bd; info[2] = nblk; info[3] = nraw; }
    return (long)op;
}

/* ------------------------------------------------------------------ */
/* Greedy single-probe LZ4 block compressor ("port": the published     */
/* LZ4 fast algorithm, 4096-entry table, hash (v*2654435761)>>20,      */
/* no skip acceleration).  Only used as a timed CPU baseline when the  */
/* system liblz4 cannot be loaded; never by the product.               */
/* ------------------------------------------------------------------ */
SKO_API size_t sko_lz4_block_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    uint32_t tab[4096];
    memset(tab, 0, sizeof tab);
    size_t ip = 0, anchor = 0, op = 0;
    if (n >= 13) {
        size_t mflimit = n - 12, matchlimit = n - 5;
        while (ip <= mflimit) {
            uint32_t v = rd32le(src + ip);
            uint32_t h = (v * 2654435761u) >> 20;
            size_t c = tab[h];
            tab[h] = (uint32_t)ip;
            if (c < ip && ip - c <= 65535 && rd32le(src + c) == v) {
                size_t ml = 4;
                while (ip + ml < matchlimit && src[c + ml] == src[ip + ml]) ml++;
                size_t lit = ip - anchor;
                if (op + 1 + lit / 255 + 1 + lit + 2 + ml / 255 + 1 > cap) return 0;
                uint8_t* tokp = dst + op++;
                if (lit >= 15) { *tokp = 0xF0; size_t r = lit - 15; while (r >= 255) { dst[op++] = 255; r -= 255; } dst[op++] = (uint8_t)r; }
                else *tokp = (uint8_t)(lit << 4);
                memcpy(dst + op, src + anchor, lit); op += lit;
                dst[op++] = (uint8_t)(ip - c); dst[op++] = (uint8_t)((ip - c) >> 8);
                size_t mc = ml - 4;
                if (mc >= 15) { *tokp |= 15; size_t r = mc - 15; while (r >= 255) { dst[op++] = 255; r -= 255; } dst[op++] = (uint8_t)r; }
                else *tokp |= (uint8_t)mc;
                ip += ml; anchor = ip;
            } else ip++;
        }
    }
    size_t lit = n - anchor;
    if (op + 1 + lit / 255 + 1 + lit > cap) return 0;
    uint8_t* tokp = dst + op++;
    if (lit >= 15) { *tokp = 0xF0; size_t r = lit - 15; while (r >= 255) { dst[op++] = 255; r -= 255; } dst[op++] = (uint8_t)r; }
    else *tokp = (uint8_t)(lit << 4);
    memcpy(dst + op, src + anchor, lit); op += lit;
    return op;
}

/* Frame writer around the port above: independent 64 KiB blocks, content size, no checksums. */
SKO_API size_t sko_lz4f_compress_port(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (cap < 15 + n + 4 * ((n + 65535) / 65536) + 4) return 0;
    size_t op = 0;
    wr32le(dst, 0x184D2204u); dst[4] = 0x68; dst[5] = 0x40;
    for (int i = 0; i < 8; i++) dst[6 + i] = (uint8_t)((uint64_t)n >> (8 * i));
    dst[14] = (uint8_t)((sko_xxh32(dst + 4, 10, 0) >> 8) & 0xFF);
    op = 15;
    for (size_t b = 0; b < n; b += 65536) {
        size_t bn = n - b < 65536 ? n - b : 65536;
        size_t c = sko_lz4_block_compress(src + b, bn, dst + op + 4, bn - 1);
        if (c == 0 || c >= bn) { wr32le(dst + op, (uint32_t)bn | 0x80000000u); memcpy(dst + op + 4, src + b, bn); op += 4 + bn; }
        else { wr32le(dst + op, (uint32_t)c); op += 4 + c; }
    }
    wr32le(dst + op, 0); op += 4;
    return op;
}

/* ------------------------------------------------------------------ */
/* WireProtocolHeader.to_bytes  (skyplane/chunk.py:141-155): 53 bytes, */
/* big-endian: magic u64, version u32=3, chunk_id 16 B, data_len u64,  */
/* raw_data_len u64, is_compressed u8, n_chunks_left_on_socket u64.    */
/* ------------------------------------------------------------------ */
static void wr64be(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i)); }
SKO_API void sko_wire_header(const uint8_t chunk_id[16], uint64_t data_len, uint64_t raw_len, int is_compressed,
                             uint64_t n_left, uint8_t out[53]) {
    wr64be(out, 0x534B595F4C41524BULL);           /* chunk.py:104-105 "SKY_LARK" */
    out[8] = 0; out[9] = 0; out[10] = 0; out[11] = 3; /* chunk.py:108-113 version 3 */
    memcpy(out + 12, chunk_id, 16);
    wr64be(out + 28, data_len);
    wr64be(out + 36, raw_len);
    out[44] = is_compressed ? 1 : 0;
    wr64be(out + 45, n_left);
}

/* ------------------------------------------------------------------ */
/* Gear CDC -- frozen specification (NOT in the reference).            */
/*   GEAR[i] = splitmix64 stream, seed 0x534B595F47454152 ("SKY_GEAR") */
/*   H(i)    = sum_{k=0..63, i-k>=0} GEAR[b[i-k]] << k   (mod 2^64)    */
/*           = the value of h after  h=(h<<1)+GEAR[b[i]]  run from     */
/*             the chunk start; only the last 64 bytes matter.         */
/*   A cut point is an END offset e (segment = [prev_cut, e)).  With   */
/*   i = e-1 the last byte of the segment and len = e - prev_cut:      */
/*     len <  min_size            : never cut                          */
/*     min_size <= len < avg_size : cut iff (H(i) & mask_s) == 0       */
/*     avg_size <= len < max_size : cut iff (H(i) & mask_l) == 0       */
/*     len == max_size            : cut                                */
/*   The chunk end is always the last cut.  mask_l's bits are a subset */
/*   of mask_s's bits, so every mask_s hit is also a mask_l hit.       */
/* ------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
SKO_API void sko_gear_table(uint64_t out[256]) {
    uint64_t s = 0x534B595F47454152ULL;
    for (int i = 0; i < 256; i++) out[i] = splitmix64(&s);
}
SKO_API size_t sko_gear_cdc(const uint8_t* data, size_t n, uint32_t min_size, uint32_t avg_size, uint32_t max_size,
                            uint64_t mask_s, uint64_t mask_l, uint32_t* cuts, size_t cuts_cap) {
    uint64_t G[256];
    sko_gear_table(G);
    size_t nc = 0, prev = 0;
    uint64_t h = 0;
    for (size_t i = 0; i < n; i++) {
        h = (h << 1) + G[data[i]];
        size_t len = i + 1 - prev;
        int cut = 0;
        if (len >= max_size) cut = 1;
        else if (len >= avg_size) cut = (h & mask_l) == 0;
        else if (len >= min_size) cut = (h & mask_s) == 0;
        if (i + 1 == n) cut = 1;
        if (cut) {
            if (nc < cuts_cap) cuts[nc] = (uint32_t)(i + 1);
            nc++;
            prev = i + 1;
        }
    }
    return nc;
}

/* ------------------------------------------------------------------ */
/* Dedup -- frozen specification.  Segments are numbered in stream     */
/* order (global index).  first[i] = smallest global index j <= i with */
/* the same 16-byte fingerprint; duplicate iff first[i] != i.          */
/* O(n log n) via sort; fingerprints fp[i*16 .. i*16+16).              */
/* ------------------------------------------------------------------ */
typedef struct { const uint8_t* fp; uint64_t idx; } sko_ent;
static int sko_cmp(const void* a, const void* b) {
    const sko_ent* x = a; const sko_ent* y = b;
    int c = memcmp(x->fp, y->fp, 16);
    if (c) return c;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
SKO_API int sko_dedup(const uint8_t* fp, size_t n, uint64_t base_index, uint64_t* first) {
    sko_ent* e = malloc(sizeof(sko_ent) * (n ? n : 1));
    if (!e) return -1;
    for (size_t i = 0; i < n; i++) { e[i].fp = fp + 16 * i; e[i].idx = i; }
    qsort(e, n, sizeof *e, sko_cmp);
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && memcmp(e[j].fp, e[i].fp, 16) == 0) { first[e[j].idx] = base_index + e[i].idx; j++; }
        i = j;
    }
    free(e);
    return 0;
}

/* Batched helpers so Python-side timing loops do not pay per-call ctypes overhead. */
SKO_API void sko_md5_batch(const uint8_t* data, size_t stride, size_t len, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++) sko_md5(data + i * stride, len, out + 16 * i);
}
