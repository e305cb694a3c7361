/* skyhost.c -- host-side helper of the destination's segment store (dedup on the wire, skyplane_amd/gateway/dedup_wire.py): an open-addressing map
 * from a segment's 128-bit fingerprint to (address, length) of its bytes.  No reference counterpart (the reference has no dedup: SURVEY fact 0.3); plain C,
 * no HIP -- it exists because a recipe names ~1800 segments per 8 MiB chunk and a Python dictionary look-up per segment, under the interpreter lock that
 * the destination's lanes share, was what bounded gpu_decompress on the dedup path (GPU call r5r: ~2.5 ms of Python per chunk against 2 ms of device time).
 * The caller (one Python lock per store) serialises calls on one map.  Built by csrc/Makefile into libskyhost.so; loaded by skyplane_amd/_hostlib.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "skyhost.h"      /* the ABI declared in include/ */

typedef struct { uint64_t lo, hi, addr; uint32_t len, used; } sky_slot;      /* 32 bytes */
struct skyhost_map { sky_slot* s; uint64_t mask, count; };

static uint64_t mix(uint64_t lo, uint64_t hi) { uint64_t x = lo ^ (hi * 0x9E3779B97F4A7C15ull); x ^= x >> 32; return x * 0xD6E8FEB86659FD93ull; }

skyhost_map* skyhost_map_new(uint32_t log2_slots) {
    if (log2_slots < 10) log2_slots = 10;
    if (log2_slots > 34) return NULL;
    skyhost_map* m = (skyhost_map*)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->mask = ((uint64_t)1 << log2_slots) - 1;
    m->s = (sky_slot*)calloc(m->mask + 1, sizeof(sky_slot));
    if (!m->s) { free(m); return NULL; }
    return m;
}
void skyhost_map_free(skyhost_map* m) { if (m) { free(m->s); free(m); } }
uint64_t skyhost_map_count(const skyhost_map* m) { return m ? m->count : 0; }

static sky_slot* find(const skyhost_map* m, uint64_t lo, uint64_t hi) {
    uint64_t i = (mix(lo, hi) >> 20) & m->mask;
    for (;;) {
        sky_slot* e = &m->s[i];
        if (!e->used || (e->lo == lo && e->hi == hi)) return e;
        i = (i + 1) & m->mask;
    }
}
static int grow(skyhost_map* m) {
    const uint64_t old_n = m->mask + 1;
    sky_slot* old = m->s;
    sky_slot* ns = (sky_slot*)calloc(old_n * 2, sizeof(sky_slot));
    if (!ns) return -1;
    m->s = ns; m->mask = old_n * 2 - 1;
    for (uint64_t i = 0; i < old_n; i++) if (old[i].used) *find(m, old[i].lo, old[i].hi) = old[i];
    free(old);
    return 0;
}

/* n entries; a fingerprint that is there already keeps its first value.  Returns the number of NEW entries (their bytes in *new_bytes), -1 on allocation failure. */
int64_t skyhost_map_put(skyhost_map* m, int64_t n, const uint8_t* fps, const uint64_t* addr, const uint32_t* len, uint64_t* new_bytes) {
    int64_t added = 0;
    uint64_t bytes = 0;
    for (int64_t k = 0; k < n; k++) {
        if ((m->count + 1) * 2 > m->mask + 1 && grow(m)) return -1;
        uint64_t lo, hi;
        memcpy(&lo, fps + 16 * k, 8); memcpy(&hi, fps + 16 * k + 8, 8);
        sky_slot* e = find(m, lo, hi);
        if (e->used) continue;
        e->lo = lo; e->hi = hi; e->addr = addr[k]; e->len = len[k]; e->used = 1;
        m->count++; added++; bytes += len[k];
    }
    if (new_bytes) *new_bytes = bytes;
    return added;
}

/* n look-ups: out_addr[k] / out_len[k], or 0 / 0 for a fingerprint that is not there.  Returns the number of misses. */
int64_t skyhost_map_get(const skyhost_map* m, int64_t n, const uint8_t* fps, uint64_t* out_addr, uint32_t* out_len) {
    int64_t miss = 0;
    for (int64_t k = 0; k < n; k++) {
        uint64_t lo, hi;
        memcpy(&lo, fps + 16 * k, 8); memcpy(&hi, fps + 16 * k + 8, 8);
        const sky_slot* e = find(m, lo, hi);
        if (e->used) { out_addr[k] = e->addr; out_len[k] = e->len; }
        else { out_addr[k] = 0; out_len[k] = 0; miss++; }
    }
    return miss;
}
