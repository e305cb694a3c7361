// lz4s_model.c -- sequential C restatement of the slice-parallel LZ4 parse (skyplane_amd/csrc/lz4s_kernel.inc).
//
// TEST INFRASTRUCTURE ONLY: nothing in the product links this file.  The GPU kernel's parse is deterministic (commutative
// table updates, slices parsed from read-only tables), so this plain loop nest is its specification: the tests require
// the kernel's compressed block to be byte-identical to lz4s_model_block()'s, on top of the reference's own acceptance
// criterion (liblz4's LZ4F_decompress, what skyplane/gateway/gateway_receiver.py:195-201 calls, reproduces the input).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lz4s_spec.h"
#ifndef LZ4S_PROBE_PAT
#define LZ4S_PROBE_PAT 0xFu
#endif
#ifndef LZ4S_VISITS_MODEL
#define LZ4S_VISITS_MODEL 12u
#endif

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t ext_bytes(uint32_t x) { return x < 15u ? 0u : 1u + (x - 15u) / 255u; }
static uint32_t put_ext(uint8_t* o, uint32_t x) {
    uint32_t r = x - 15u, k = 0;
    while (r >= 255u) { o[k++] = 255; r -= 255u; }
    o[k++] = (uint8_t)r;
    return k;
}
static uint32_t common(const uint8_t* a, const uint8_t* b, uint32_t lim) {
    uint32_t k = 0;
    while (k < lim && a[k] == b[k]) k++;
    return k;
}

typedef struct { uint32_t n_rec, n_kept, n_trimmed, n_dropped, n_probe_fail; } lz4s_stats;

// Compress one block of n <= 65536 bytes.  Returns the compressed size; writes the block when dst != NULL
// (dst must hold n + n/255 + 16 bytes).  recs_out (optional, 16 * LZ4S_LANES words) receives the raw per-slice records.
uint32_t* lz4s_dbg_visits = 0;   /* optional: LZ4S_LANES words, visits made per slice (load statistics for kernel tuning) */
/* optional trace of every visit, for the LDS cost model of the kernel's parse (scripts/dev/lz4s_cost_model.py): one word per visit,
   slice | visit number << 10 | valid table candidates (bit per region) << 14 | 16-byte extension steps << 18; lz4s_dbg_ntrace counts them */
uint32_t* lz4s_dbg_trace = 0;
uint32_t lz4s_dbg_trace_cap = 0, lz4s_dbg_ntrace = 0;
uint32_t lz4s_model_block(const uint8_t* s, uint32_t n, uint8_t* dst, lz4s_stats* st) {
    const uint32_t NB = 1u << LZ4S_LOGB;
    uint32_t* T = (uint32_t*)malloc((size_t)NB * LZ4S_Q * 4);
    uint32_t* rec = (uint32_t*)malloc((size_t)LZ4S_LANES * 16 * 4);
    uint32_t* nrec = (uint32_t*)calloc(LZ4S_LANES, 4);
    lz4s_stats z = {0, 0, 0, 0, 0};
    uint32_t op = 0, anchor = 0;
    memset(T, 0xFF, (size_t)NB * LZ4S_Q * 4);
    if (n >= 13u) {
        const uint32_t mflimit = n - 12u, matchlimit = n - 5u;
        // pre-pass: earliest position per (bucket, region) among equal tags; every LZ4S_INS_STEP-th position from LZ4S_FIRST_INS on is entered
        for (uint32_t p = LZ4S_FIRST_INS; p <= mflimit; p++) {
            if (((p & (LZ4S_SLICE - 1u)) % LZ4S_INS_STEP) != 0u) continue;      // positions 0, STEP, 2 STEP, ... of every 64-byte slice
            const uint32_t x = LZ4S_HASH(rd32(s + p), s[p + 4]);
            uint32_t* e = &T[LZ4S_BUCKET(x) * LZ4S_Q + (p >> LZ4S_RLOG)];
            const uint32_t v = LZ4S_ENTRY(LZ4S_TAG(x), p & ((1u << LZ4S_RLOG) - 1u));
            if (v < *e) *e = v;
        }
        // slice parse
        const uint32_t nsl = (n + LZ4S_SLICE - 1u) / LZ4S_SLICE;
        for (uint32_t j = 0; j < nsl; j++) {
            const uint32_t s0 = j * LZ4S_SLICE, s1 = s0 + LZ4S_SLICE < n ? s0 + LZ4S_SLICE : n;
            const uint32_t lim = s1 + LZ4S_EXT < matchlimit ? s1 + LZ4S_EXT : matchlimit;
            uint32_t pos = s0, lanchor = s0, visits = 0;
            for (uint32_t p = s0; p < s1 && p <= mflimit && visits < LZ4S_VISITS_MODEL; p++) {
                if (p < pos) continue;
                const uint32_t x = LZ4S_HASH(rd32(s + p), s[p + 4]);
                const uint32_t q = p >> LZ4S_RLOG, tb = LZ4S_TAG(x) << 16, rel = p & ((1u << LZ4S_RLOG) - 1u);
                const uint32_t* e = &T[LZ4S_BUCKET(x) * LZ4S_Q];
                uint32_t best = 0, bc = 0, validmask = 0, extsteps = 0;
                int cand = 0;
                const uint32_t cap8 = lim - p < 8u ? lim - p : 8u;
                // short-period candidate (runs, "abab", 32-bit patterns): the 4 bytes before p repeat at p.  Overlapping
                // copies run as long as the period holds, which no table entry (the EARLIEST occurrence) can offer.
                if (p >= 4u && rd32(s + p - 4u) == rd32(s + p)) { best = common(s + p, s + p - 4u, cap8); bc = p - 4u; cand = 1; }
#if LZ4S_PER1
                // distance 1 (a run of one byte value begins at p - 1): found at the run's second byte, where the distance-4 test needs its fifth
                if (p >= 1u && rd32(s + p - 1u) == rd32(s + p)) { const uint32_t l = common(s + p, s + p - 1u, cap8); if (l >= best) { best = l; bc = p - 1u; } cand = 1; }
#endif
                // best = the longest of the candidates, ties to the nearest (largest position)
                for (int k = (int)q; k >= 0; k--) {                   // nearest region first; a farther one must be strictly longer
                    if (!((LZ4S_PROBE_PAT >> (p & 3u)) & 1u)) break;  // positions whose residue mod 4 is not in the pattern are not looked up (as shipped: all are)
                    const uint32_t d = e[k] - tb;
                    if (!((uint32_t)k < q ? d < 0x10000u : d < rel)) continue;
                    cand = 1;
                    validmask |= 1u << k;
                    const uint32_t c = ((uint32_t)k << LZ4S_RLOG) + d;
                    const uint32_t l = common(s + p, s + c, cap8);
                    if (l >= 4u && (l > best || (l == best && c > bc))) { best = l; bc = c; }
                }
                const uint32_t this_visit = visits;
                if (cand) visits++;                                   // a position with a candidate by tag (or a period hit) is a visit of the lane
                if (!best) {                                          // tag hit that does not verify (or no candidate): a literal
                    if (cand && lz4s_dbg_trace && lz4s_dbg_ntrace < lz4s_dbg_trace_cap) lz4s_dbg_trace[lz4s_dbg_ntrace++] = j | this_visit << 10 | validmask << 14;
                    z.n_probe_fail++;
                    continue;
                }
                uint32_t len = best;
                if (len == 8u) {
                    while (p + len < lim) {
                        const uint32_t room = lim - (p + len) < 8u ? lim - (p + len) : 8u;
                        const uint32_t t = common(s + p + len, s + bc + len, room);
                        len += t;
                        if (t < 8u) break;
                    }
                    extsteps = (len - 8u + 15u) / 16u + (p + len < lim && ((len - 8u) & 15u) == 0u ? 1u : 0u);   /* 16-byte steps the kernel takes: the last one finds the mismatch (or the limit) */
                }
                if (lz4s_dbg_trace && lz4s_dbg_ntrace < lz4s_dbg_trace_cap) lz4s_dbg_trace[lz4s_dbg_ntrace++] = j | this_visit << 10 | validmask << 14 | (extsteps > 16383u ? 16383u : extsteps) << 18;
                // move the start back over pending literals: at most LZ4S_BACK bytes (4 as shipped; with a larger value the distance-4 and distance-1
                // candidates still get 4: the kernel of rounds 2-3 had only 4 of the 8 bytes before position p - 4 in registers)
                const uint32_t backmax = LZ4S_BACK < 4u ? LZ4S_BACK : ((bc + 4u == p || (LZ4S_PER1 && bc + 1u == p)) ? 4u : LZ4S_BACK);
                uint32_t nb = 0;
                while (nb < backmax && p - nb > lanchor && bc - nb > 0u && s[p - nb - 1u] == s[bc - nb - 1u]) nb++;
                const uint32_t mp = p - nb, c0 = bc - nb;
                len += nb;
                rec[j * 16u + nrec[j]++] = LZ4S_REC(mp - c0, mp - lanchor, len);
                z.n_rec++;
                pos = mp + len; lanchor = pos;
            }
            if (lz4s_dbg_visits) lz4s_dbg_visits[j] = visits;
        }
        // stitch: trim overlaps in slice order; a slice's first surviving match that continues the previous surviving match
        // (no literals in between, same offset) is merged into it; then emit
        typedef struct { uint32_t mp, len, off; } seq_t;
        seq_t* q = (seq_t*)malloc(sizeof(seq_t) * (LZ4S_LANES * 16 + 1));
        uint32_t nq = 0;
        uint32_t cover = 0;        // prefix max of RAW match ends (what the trimming sees)
        for (uint32_t j = 0; j < nsl; j++) {
            uint32_t a = j * LZ4S_SLICE;   // running position inside the slice's record list
            int first = 1;
            for (uint32_t k = 0; k < nrec[j]; k++) {
                const uint32_t r = rec[j * 16u + k];
                uint32_t mp = a + LZ4S_REC_LIT(r), len = LZ4S_REC_LEN(r);
                const uint32_t off = LZ4S_REC_OFF(r), end = mp + len;
                a = end;
                if (end <= cover) { z.n_dropped++; continue; }
                if (mp < cover) {
                    if (end - cover < 4u) { z.n_dropped++; cover = end; continue; }   // NOTE: the raw end still counts as covered-for-trimming
                    len = end - cover; mp = cover; z.n_trimmed++;
                }
                cover = end;
                if (mp > mflimit) { z.n_dropped++; continue; }
                if (first && nq && q[nq - 1].mp + q[nq - 1].len == mp && q[nq - 1].off == off) q[nq - 1].len += len;
                else { q[nq].mp = mp; q[nq].len = len; q[nq].off = off; nq++; }
                first = 0;
                z.n_kept++;
            }
        }
        for (uint32_t t = 0; t < nq; t++) {
            const uint32_t mp = q[t].mp, len = q[t].len, off = q[t].off;
            const uint32_t lit = mp - anchor, ml = len - 4u;
            if (dst) {
                dst[op] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
                uint32_t o = op + 1u;
                if (lit >= 15u) o += put_ext(dst + o, lit);
                memcpy(dst + o, s + anchor, lit); o += lit;
                dst[o] = (uint8_t)off; dst[o + 1] = (uint8_t)(off >> 8); o += 2u;
                if (ml >= 15u) o += put_ext(dst + o, ml);
                op = o;
            } else op += 1u + ext_bytes(lit) + lit + 2u + ext_bytes(ml);
            anchor = mp + len;
        }
        free(q);
    }
    {   // last sequence: literals only
        const uint32_t lit = n - anchor;
        if (dst) {
            dst[op] = (uint8_t)((lit < 15u ? lit : 15u) << 4);
            uint32_t o = op + 1u;
            if (lit >= 15u) o += put_ext(dst + o, lit);
            memcpy(dst + o, s + anchor, lit);
            op = o + lit;
        } else op += 1u + ext_bytes(lit) + lit;
    }
    if (st) *st = z;
    free(T); free(rec); free(nrec);
    return op;
}
