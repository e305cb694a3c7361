/* skyhost.h -- C ABI of libskyhost.so (skyplane_amd/csrc/skyhost.c): the host-side fingerprint -> (address, length) map behind the destination's
 * device-resident segment store (skyplane_amd/gateway/dedup_wire.py::DeviceSegmentStore, dedup on the wire: SURVEY 8f item 4).
 *
 * No reference counterpart: skyplane has no chunking or dedup.  The stand-in it replaces is this repo's own Python dictionary
 * (SegmentStore._segs: one dictionary operation per segment, ~1800 per 8 MiB chunk, under the interpreter lock the destination's lanes share).  Plain C,
 * no HIP; the caller serialises calls on one map (one lock per store). */
#ifndef SKYHOST_H
#define SKYHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct skyhost_map skyhost_map;

skyhost_map* skyhost_map_new(uint32_t log2_slots);      /* open addressing, grows by doubling before it is half full; NULL on allocation failure */
void         skyhost_map_free(skyhost_map* m);
uint64_t     skyhost_map_count(const skyhost_map* m);

/* n entries: fingerprint fps[16 k .. 16 k + 16) -> (addr[k], len[k]).  A fingerprint that is there already keeps its FIRST value.  Returns the number of new
 * entries (the sum of their lengths in *new_bytes, may be NULL), -1 on allocation failure. */
int64_t      skyhost_map_put(skyhost_map* m, int64_t n, const uint8_t* fps, const uint64_t* addr, const uint32_t* len, uint64_t* new_bytes);

/* n look-ups: out_addr[k] / out_len[k], or 0 / 0 for a fingerprint that is not there.  Returns the number of misses. */
int64_t      skyhost_map_get(const skyhost_map* m, int64_t n, const uint8_t* fps, uint64_t* out_addr, uint32_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
