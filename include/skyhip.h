/*
 * skyhip.h -- C ABI of libskyhip.so, the MI355X (gfx950) chunk-processing stage for the Skyplane gateway.
 *
 * The reference (skyplane-project/skyplane v0.3.2) is pure Python and has NO FFI for this path; the calls
 * below are what a ctypes binding inside the reference would replace (see INTEGRATION.md for that stub):
 *
 *   skyhip_process_batch / skyhip_process_device
 *       -> `data = lz4.frame.compress(data)`           skyplane/gateway/operators/gateway_operator.py:358-361
 *       -> `hashlib.md5()` / `.update()` / `.digest()`  skyplane/obj_store/s3_interface.py:181-192 (and the
 *          gcs/azure/cos/scp siblings), requested by `generate_md5=True` at gateway_operator.py:555-565;
 *          the digest is what Chunk.md5_hash (skyplane/chunk.py:21) is declared to carry.
 *       -> (new, not in the reference) Gear content-defined cut points, per-segment fingerprints and the
 *          on-GPU dedup table named by BASELINE.json's north_star.
 *   skyhip_frame_bound
 *       -> the size a receiver must be prepared to read: WireProtocolHeader.data_len (skyplane/chunk.py:99).
 *
 * Contract: plain pointers and sizes only; one context per worker process, created AFTER
 * fork, used from one thread at a time; calls are synchronous; every entry point returns 0 on success or
 * a negative SKYHIP_E_* code -- no exceptions, no aborts, output buffers are not to be trusted on failure.
 * The frames produced decode with lz4.frame.decompress (gateway_receiver.py:195-201) to the input bytes;
 * digests equal hashlib.md5(raw).digest().  There is NO CPU fallback inside this library.
 */
#ifndef SKYHIP_H
#define SKYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKYHIP_ABI_VERSION 1

/* flags for process_batch / process_device */
#define SKYHIP_F_LZ4   0x1u   /* emit an LZ4 frame per chunk */
#define SKYHIP_F_MD5   0x2u   /* emit the RFC 1321 digest of each chunk's raw bytes */
#define SKYHIP_F_CDC   0x4u   /* emit Gear content-defined cut points (+ per-segment MD5 fingerprints) */
#define SKYHIP_F_DEDUP 0x8u   /* look the fingerprints up in / insert them into the context's dedup table */

/* error codes (all negative) */
#define SKYHIP_OK            0
#define SKYHIP_E_INVAL      -1   /* bad argument */
#define SKYHIP_E_NOMEM      -2   /* hipMalloc / hipHostMalloc failed */
#define SKYHIP_E_HIP        -3   /* a HIP runtime call or kernel launch failed (see skyhip_last_hip_error) */
#define SKYHIP_E_TOOBIG     -4   /* chunk larger than max_chunk_bytes / batch larger than supported */
#define SKYHIP_E_CAP        -5   /* an output buffer is smaller than skyhip_frame_bound(len) / cut capacity */
#define SKYHIP_E_NODEVICE   -6   /* no usable gfx950 device */
#define SKYHIP_E_TABLEFULL  -7   /* (no longer returned: a full dedup table is aged, segments that find no slot are reported as new) */
#define SKYHIP_E_FORMAT     -8   /* a frame handed to skyhip_decompress_* is malformed or uses an unsupported option */

typedef struct skyhip_ctx skyhip_ctx;   /* opaque: owns device scratch, streams, events, dedup table */

/* Per-kernel device time (HIP events on the library's own streams) accumulated since the last reset. */
typedef struct skyhip_timing {
    double lz4_ms;        /* sky_lz4s_frames (large device-resident batches) / sky_lz4s_compress (block queue): the dominant kernel */
    double layout_ms;     /* sky_frame_layout */
    double gather_ms;     /* sky_frame_gather */
    double md5_ms;        /* sky_md5_chunks */
    double cdc_ms;        /* gear candidate + cut selection + segment fingerprints + dedup */
    uint64_t lz4_launches;
    uint64_t lz4_in_bytes;   /* raw bytes the LZ4 kernel consumed */
    uint64_t lz4_out_bytes;  /* frame bytes produced (headers, size words and EndMark included) */
    uint64_t md5_launches;
    uint64_t md5_in_bytes;
} skyhip_timing;

int  skyhip_abi_version(void);

/* device_id: HIP ordinal.  max_chunk_bytes: largest chunk a call may contain (<= 1 GiB).
 * max_batch: number of chunks whose LZ4 scratch is resident at once; larger batches are processed in
 * sub-batches of this size (MD5 always runs over the whole batch in one launch). */
int  skyhip_create(int device_id, size_t max_chunk_bytes, int max_batch, skyhip_ctx** out);
void skyhip_destroy(skyhip_ctx* ctx);

/* Worst-case LZ4 frame size for raw_len input bytes: 15-byte header + 4 bytes per 64 KiB block + raw + 4. */
size_t skyhip_frame_bound(size_t raw_len);

/* Host-buffer batch (what the gateway operator calls).  Caller owns every buffer; pinned memory is
 * faster but not required.  in[i]/in_len[i]: raw chunk bytes.  out[i]/out_cap[i]: where chunk i's LZ4
 * frame goes (cap >= skyhip_frame_bound(in_len[i])); out_len[i] receives the frame size.
 * md5[i]: 16-byte digest.  cuts[i]/cuts_cap[i]/n_cuts[i]: optional CDC END offsets (NULL to skip). */
int  skyhip_process_batch(skyhip_ctx* ctx, int n,
                          const uint8_t* const* in, const size_t* in_len,
                          uint8_t* const* out, const size_t* out_cap, size_t* out_len,
                          uint8_t (*md5)[16],
                          uint32_t* const* cuts, const size_t* cuts_cap, size_t* n_cuts,
                          uint32_t flags);

/* Pinned (DMA-able) host memory for the buffers handed to skyhip_process_batch / skyhip_decompress_batch: with it the
 * H2D/D2H copies are asynchronous and overlap the kernels of the neighbouring sub-batch; pageable buffers are accepted
 * but staged synchronously by the runtime.  This is the zero-copy hand-off of SURVEY.md 8f item 2: the operator reads
 * `<chunk_id>.chunk` straight into such a buffer instead of `f.read()` at gateway_operator.py:350-351.
 * Blocks still alive at skyhip_destroy are freed there. */
int  skyhip_host_alloc(skyhip_ctx* ctx, size_t bytes, void** out);
int  skyhip_host_free(skyhip_ctx* ctx, void* p);
/* Page-lock memory the CALLER owns -- in the gateway: a MAP_SHARED mapping of an arena file in the chunk directory (tmpfs), the shared-memory
 * hand-off between gpu_compress and the sender ("pinned shared memory instead of tmpfs files"; replaces the per-chunk
 * `<id>.chunk.lz4f` file the sender read back with `f.read()`, gateway_operator.py:350-351).  Frames then travel device -> arena by DMA and
 * arena -> socket by sendfile: no CPU copy in between.  The range must stay mapped until skyhip_host_unregister (skyhip_destroy unregisters what
 * is left).  Registering is expensive (pages are pinned): do it once per arena, not per chunk. */
int  skyhip_host_register(skyhip_ctx* ctx, void* p, size_t bytes);
int  skyhip_host_unregister(skyhip_ctx* ctx, void* p);

/* Device-resident batch (kernel-only path: inputs already in HBM, PCIe excluded).
 * d_in / d_out are DEVICE pointers on ctx's device; in_off/in_len/out_off/out_cap are HOST arrays of
 * byte offsets into d_in / d_out.  out_len (host, may be NULL) and md5 (host, may be NULL) are filled
 * after the batch completes.  Frames are written at d_out + out_off[i].
 * A batch of at least two chunks per CU of the device (environment SKYHIP_FRAMES_MIN at skyhip_create: another threshold, 0 = never) is compressed by ONE
 * launch in which a workgroup takes a whole chunk at a time and writes its frame in place -- header, block size words, blocks, EndMark --; smaller
 * batches go block by block through scratch slots and a gather pass.  Both produce the same bytes.  Frame regions [out_off[i], out_off[i] + out_cap[i])
 * must NOT overlap: the in-place launch has no ordering between workgroups (chunks are STARTED in index order, nothing says when one is finished), so two
 * chunks that share a region may leave a mixture of both frames.  Two contexts of one process may
 * call this concurrently (one host thread each): their whole-chip compressor launches are queued back to back on the device, their digest kernels overlap. */
int  skyhip_process_device(skyhip_ctx* ctx, int n,
                           const void* d_in, const uint64_t* in_off, const uint64_t* in_len,
                           void* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                           uint64_t* out_len, uint8_t (*md5)[16], uint32_t flags);

/* LZ4 frame DEcompression -- the destination gateway's mirrored hot loop:
 *   `data_batch_decompressed = lz4.frame.decompress(to_write)`   skyplane/gateway/operators/gateway_receiver.py:195-201
 * Accepts frames with 64 KiB..4 MiB blocks, block-independent (this library's) or block-linked (python-lz4's
 * default), stored blocks allowed, content size present or absent, no checksums/dictionary.  out_len[i] = decoded
 * bytes (== the frame's content size when it has one), which the caller compares with
 * WireProtocolHeader.raw_data_len as gateway_receiver.py:213-218 does.
 * status[i] (may be NULL) = 0 or a positive decoder code; a rejected frame yields out_len[i] = 0 and the call
 * returns SKYHIP_E_FORMAT after decoding every good frame. */
int  skyhip_decompress_device(skyhip_ctx* ctx, int n,
                              const void* d_in, const uint64_t* in_off, const uint64_t* in_len,
                              void* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                              uint64_t* out_len, int32_t* status);
int  skyhip_decompress_batch(skyhip_ctx* ctx, int n,
                             const uint8_t* const* in, const size_t* in_len,
                             uint8_t* const* out, const size_t* out_cap, size_t* out_len, int32_t* status);
/* Same, plus md5[i] = RFC 1321 digest of chunk i's DECODED bytes, computed on the device right after the decode --
 * the receiver-side check the reference leaves as `# todo check hash` (gateway_receiver.py:231); md5 may be NULL.
 * Needs out_cap[i] <= max_chunk_bytes of the context. */
int  skyhip_decompress_batch_md5(skyhip_ctx* ctx, int n,
                                 const uint8_t* const* in, const size_t* in_len,
                                 uint8_t* const* out, const size_t* out_cap, size_t* out_len, int32_t* status,
                                 uint8_t (*md5)[16]);
double skyhip_decompress_ms(skyhip_ctx* ctx, int reset);   /* accumulated device time of scan + decode kernels */

/* CDC results of the LAST device launch that had SKYHIP_F_CDC set (host copies).  skyhip_process_batch splits a batch
 * larger than max_batch into sub-batches; it fills its own cuts[]/n_cuts[] for every chunk, but this call then
 * describes the last sub-batch only -- use batches <= max_batch (or skyhip_process_device) when fingerprints are needed.
 * n_cuts[i] cut END offsets for chunk i are written to cuts + cut_prefix[i] ... ; fps holds 16 bytes per
 * segment in the same order; first_seen[k] = global index of the first segment with that fingerprint
 * (== its own global index when it is not a duplicate).  Any pointer may be NULL. */
int  skyhip_cdc_results(skyhip_ctx* ctx, int n, uint64_t* cut_prefix /* n+1 */, uint32_t* cuts, size_t cuts_cap,
                        uint8_t* fps, uint64_t* first_seen, uint64_t* seg_base_index);

/* Dedup on the wire, source side (skyplane_amd/gateway/dedup_wire.py; no reference counterpart: SURVEY 8f item 4).  Valid right after a
 * skyhip_process_batch call with SKYHIP_F_CDC | SKYHIP_F_DEDUP over the same n <= max_batch chunks: for every chunk i the concatenation of its NEW
 * segments (first_seen == own index) -- the chunk's literal stream -- is put together ON THE DEVICE from the chunks still resident in the context's
 * staging area and compressed into one LZ4 frame in out[i] (out_cap[i] >= skyhip_frame_bound(in_len[i]); pinned memory makes the copy asynchronous).
 * lit_len[i] = raw length of the stream.  out_len[i] = 0 for a chunk without duplicates (lit_len == in_len: the frame the first call made IS its
 * literal stream) and for one without new segments. */
int  skyhip_dedup_literals(skyhip_ctx* ctx, int n, uint8_t* const* out, const size_t* out_cap, size_t* out_len, size_t* lit_len);

/* Dedup on the wire, destination side (skyplane_amd/gateway/dedup_wire.py): a recipe's chunk is put together ON THE DEVICE from literal streams that stay
 * there.  skyhip_dev_alloc / skyhip_dev_free: device memory the caller owns (the segment store's literal streams); it belongs to the PROCESS, not to the
 * context that allocated it -- any context of the process may read it, it outlives skyhip_destroy, and skyhip_dev_free(NULL, p) frees it when that context
 * is gone (the lanes of a worker share one store and leave at different times).  skyhip_decompress_to_device: like
 * skyhip_decompress_batch, but frame i is decoded into DEVICE memory dst[i] (>= out_cap[i] bytes) and nothing comes back to the host but lengths and
 * status.  skyhip_gather_md5: chunk i := the runs run_prefix[i] .. run_prefix[i+1] back to back, run k being run_len[k] bytes at DEVICE address
 * run_src[k]; the chunk is copied to host out[i] (pinned memory: asynchronously, beside the digest) and md5[i] (may be NULL) is its digest --
 * what lz4.frame.decompress + the "# todo check hash" of gateway_receiver.py:195-231 amount to for a chunk that travelled as a recipe. */
int  skyhip_dev_alloc(skyhip_ctx* ctx, size_t bytes, void** out);
int  skyhip_dev_free(skyhip_ctx* ctx, void* p);
int  skyhip_decompress_to_device(skyhip_ctx* ctx, int n, const uint8_t* const* in, const size_t* in_len, void* const* dst, const size_t* out_cap,
                                 size_t* out_len, int32_t* status);
int  skyhip_gather_md5(skyhip_ctx* ctx, int n, const uint64_t* run_prefix /* n+1 */, const uint64_t* run_src, const uint32_t* run_len,
                       uint8_t* const* out, const size_t* out_cap, size_t* out_len, uint8_t (*md5)[16]);
/* MD5 of nseg byte ranges that are ALREADY in device memory (range k = len[k] < 32768 bytes at DEVICE address dev_addr[k]): fps[k] = its RFC 1321 digest.
 * What the destination of a deduplicated transfer checks its newly arrived literal segments with -- every segment of a recipe carries its fingerprint,
 * the segments are independent messages (one lane each, thousands at once), and a chunk whose literals all match and whose references name segments that
 * matched when THEY arrived needs no serial whole-chunk chain on the batch's critical path (the reference's "# todo check hash",
 * skyplane/gateway/operators/gateway_receiver.py:231; the whole-chunk digest still travels with the chunk for the object store's ContentMD5 check,
 * gateway_operator.py:633-643). */
int  skyhip_segment_md5_device(skyhip_ctx* ctx, size_t nseg, const uint64_t* dev_addr, const uint32_t* len, uint8_t (*fps)[16]);

/* Forget every fingerprint in the dedup table. */
int  skyhip_dedup_reset(skyhip_ctx* ctx);

void skyhip_get_timing(skyhip_ctx* ctx, skyhip_timing* out);
void skyhip_reset_timing(skyhip_ctx* ctx);

/* On-device self test of the wavefront primitives (DPP scan vs ds_bpermute scan, ballot, readlane).
 * Returns 0 when every check passes, otherwise the index (>0) of the first failing check. */
int  skyhip_selftest(skyhip_ctx* ctx);

/* Development aid: accumulated in-kernel phase timers of a -DSKY_PROF=1 build (all zeros in the shipping build). */
int  skyhip_debug_prof(skyhip_ctx* ctx, uint64_t out[16]);
/* test hook: make the (n+1)-th checked HIP call of this context fail (n < 0: off); the failing call returns SKYHIP_E_HIP and
   must leave the context usable */
int  skyhip_debug_fault(skyhip_ctx* ctx, long n);

/* test instrumentation (tests/test_gpu_guard.py): device buffers placed with the HIP virtual-memory API so that they END on the last mapped byte before an
   unmapped address range (at_end != 0) or START right behind one -- a kernel that touches one byte outside dies with a memory access fault.  With
   SKYHIP_GUARD_ALLOC=1 in the environment the library places every device buffer it allocates for itself the same way, with no slack.  _probe reads one
   byte at p on the device and returns it. */
int  skyhip_debug_guard_alloc(size_t bytes, int at_end, void** out);
int  skyhip_debug_guard_free(void* p);
int  skyhip_debug_guard_probe(skyhip_ctx* ctx, const void* p);

const char* skyhip_strerror(int code);
const char* skyhip_last_hip_error(skyhip_ctx* ctx);   /* hipGetErrorString of the last failing HIP call; ctx == NULL: of this thread's last failed skyhip_create */

#ifdef __cplusplus
}
#endif
#endif /* SKYHIP_H */
