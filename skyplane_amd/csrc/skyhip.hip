// skyhip.hip -- device entry points + host side of the C ABI declared in include/skyhip.h.
//
// gfx950 only.  No CPU fallback lives here: if the HIP runtime or the device is missing every call fails
// with a negative code and the Python binding raises.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <unistd.h>

#include "skyhip.h"
#include "wave.h"
#include "skyhip_kernels.h"
#include "lz4_common.inc"
#include "lz4s_kernel.inc"
#include "md5_kernel.inc"
#include "frame_kernel.inc"
#include "lz4d_kernel.inc"
#ifdef SKY_WITH_CDC
#include "gear_kernel.inc"
#endif

// ------------------------------------------------------------------------------------------------
// __global__ wrappers around the portable kernel bodies
// ------------------------------------------------------------------------------------------------
#ifndef LZ4S_KERNEL_ATTR
// 5 waves per SIMD = 96 VGPRs: a compressor workgroup puts 4 waves on every SIMD of its CU, so one more wave of another kernel (MD5, frame
// gather) fits beside it.  At the 128 VGPRs the launch bound alone would allow, the compressor runs 3 % faster alone but nothing else can
// enter a CU it holds: the step of the bench went from 265 to 239 ms with this attribute (profiles/r2_coexist_ab.txt).
#define LZ4S_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
extern "C" __global__ void __launch_bounds__(LZ4S_LANES) LZ4S_KERNEL_ATTR sky_lz4s_compress(SkyLz4Args a) {
    // The kernel's LDS (141 KiB, dynamic: a static array of that size would tell the register allocator that only four waves per SIMD fit and it would
    // take 128 VGPRs) is addressed ABSOLUTELY, from LDS byte 16: the kernel has no other LDS, so the launch's dynamic segment starts at 0.  Through an
    // `extern __shared__` array every one of the kernel's ~200 LDS address computations carried a `v_add_u32 v, <base>, v` for a base that is only
    // known at link time.  (16, not 0: a literal 0 is the null pointer, which is -1 in the LDS address space.)
#ifndef LZ4S_ABS_LDS
#define LZ4S_ABS_LDS 1
#endif
#if LZ4S_ABS_LDS
    typedef __attribute__((address_space(3))) uint8_t sky_lds_u8;
    uint8_t* smem = (uint8_t*)(sky_lds_u8*)(uintptr_t)LZ4S_LDS_ORIGIN;
#else
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#endif
    sky_lz4s_compress_body(a, smem);
}
// the same compressor writing whole frames in place (one workgroup per chunk at a time; lz4s_kernel.inc): large device-resident batches
extern "C" __global__ void __launch_bounds__(LZ4S_LANES) LZ4S_KERNEL_ATTR sky_lz4s_frames(SkyLz4FArgs a) {
#if LZ4S_ABS_LDS
    typedef __attribute__((address_space(3))) uint8_t sky_lds_u8;
    uint8_t* smem = (uint8_t*)(sky_lds_u8*)(uintptr_t)LZ4S_LDS_ORIGIN;
#else
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#endif
    sky_lz4s_frames_body(a, smem);
}
#ifndef SKY_MD5_KERNEL_ATTR
#define SKY_MD5_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))      // <= 128 VGPRs: fits in what four compressor waves leave of a SIMD
#endif
extern "C" __global__ void __launch_bounds__(256) SKY_MD5_KERNEL_ATTR sky_md5_chunks(SkyMd5Args a) { sky_md5_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_frame_layout(SkyFrameArgs a) { sky_frame_layout_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_frame_gather(SkyFrameArgs a) { sky_frame_gather_body(a); }
extern "C" __global__ void __launch_bounds__(64) sky_lz4f_scan(SkyLz4dArgs a) { sky_lz4f_scan_body(a); }
#ifndef SKY_D_WAVES
#define SKY_D_WAVES 8      // wavefronts per SIMD the decoder is compiled for: the kernel waits on memory (72 % of its wave-cycles), 64 registers with 14 spilled beat 79 by 16 %
#endif
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SKY_D_WAVES, 8))) sky_lz4_decode(SkyLz4dRun r) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * SKY_D_STAGE_LDS];      // a batch and a window per wavefront
    sky_lz4_decode_body(r, smem);
}
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SKY_D_WAVES, 8))) sky_lz4_parse(SkyLz4dLink r) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * SKY_D_STAGE_LDS];
    sky_lz4_parse_body(r, smem);
}
extern "C" __global__ void __launch_bounds__(SKY_LZ4D_LINK_LANES) sky_lz4_link(SkyLz4dLink r) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    sky_lz4_link_body(r, smem);
}
extern "C" __global__ void __launch_bounds__(SKY_LZ4R_LANES) sky_lz4_resolve(SkyLz4dResolve r) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    sky_lz4_resolve_body(r, smem);
}
extern "C" __global__ void __launch_bounds__(SKY_LZ4R_LANES) sky_lz4_chain(SkyLz4dResolve r) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    sky_lz4_chain_body(r, smem);
}
extern "C" __global__ void __launch_bounds__(64) sky_lz4_decode_seq(SkyLz4dRun r) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    sky_lz4_decode_seq_body(r, smem);
}
#ifdef SKY_WITH_CDC
#define SKY_GEAR_LDS_ORIGIN 16u      // the kernel's only LDS, addressed absolutely like the compressor's: no `v_add_u32 v, <link-time base>, v` per table look-up
extern "C" __global__ void __launch_bounds__(SKY_GEAR_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) sky_gear_candidates(SkyGearArgs a) {      // <= 64 VGPRs
    typedef __attribute__((address_space(3))) uint8_t sky_lds_u8;
    sky_gear_candidates_body(a, (uint8_t*)(sky_lds_u8*)(uintptr_t)SKY_GEAR_LDS_ORIGIN);
}
extern "C" __global__ void __launch_bounds__(64) sky_gear_select(SkyGearArgs a) { sky_gear_select_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_seg_prefix(SkySegPrefixArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    sky_seg_prefix_body(a, smem);
}
extern "C" __global__ void __launch_bounds__(256) sky_seg_desc(SkySegDescArgs a) { sky_seg_desc_body(a); }
extern "C" __global__ void __launch_bounds__(64) SKY_MD5_KERNEL_ATTR sky_segment_md5x(SkySegMd5Args a) {      // the same digests, rows staged through LDS (gear_kernel.inc)
    __shared__ __attribute__((aligned(16))) uint8_t smem[SKY_SEGX_LDS];
    sky_segment_md5x_body(a, smem);
}
extern "C" __global__ void __launch_bounds__(64) SKY_MD5_KERNEL_ATTR sky_segment_md5(SkySegMd5Args a) { sky_segment_md5_body(a); }      // <= 128 VGPRs like sky_md5_chunks: runs beside the compressor
extern "C" __global__ void __launch_bounds__(256) sky_gather_runs(SkyRunArgs a) { sky_gather_runs_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_lit_plan(SkyLitArgs a) { sky_lit_plan_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_lit_gather(SkyLitArgs a) { sky_lit_gather_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_dedup_insert(SkyDedupArgs a) { sky_dedup_insert_body(a); }
extern "C" __global__ void __launch_bounds__(256) sky_dedup_resolve(SkyDedupArgs a) { sky_dedup_resolve_body(a); }
#endif

// wave-primitive self test: one wave, results[k] != 0 marks a failing check
extern "C" __global__ void __launch_bounds__(64) sky_selftest_kernel(const uint8_t* buf, uint8_t* wbuf, uint32_t* results) {
    const int lane = sky_lane();
    // 1: DPP scan == ds_bpermute scan on a non-trivial pattern
    uint32_t x = (uint32_t)(lane * 2654435761u) >> 20;
    uint32_t a = sky_scan_incl_add(x), b = sky_scan_incl_add_shfl(x);
    uint32_t ref = 0;
    for (int i = 0; i <= lane; i++) ref += ((uint32_t)(i * 2654435761u) >> 20);
    sky_u64 bad1 = sky_ballot(a != b || a != ref);
    // 2: ballot / ctz / readlane
    sky_u64 m = sky_ballot((lane % 3) == 1);
    uint32_t r7 = sky_readlane((uint32_t)lane * 5u + 1u, 7);
    bool bad2 = (sky_ctz64(m) != 1) || (sky_popc64(m) != 21) || (r7 != 36u);
    // 3: unaligned loads of every width at odd offsets
    const uint8_t* p = buf + 1 + lane * 3;
    uint32_t v32 = sky_ld32u(p);
    sky_u64 v64 = sky_ld64u(p);
    sky_u128 v128 = sky_ld128u(p);
    uint32_t e32 = 0; sky_u64 e64 = 0;
    for (int k = 0; k < 4; k++) e32 |= (uint32_t)p[k] << (8 * k);
    for (int k = 0; k < 8; k++) e64 |= (sky_u64)p[k] << (8 * k);
    uint32_t e3 = 0; for (int k = 0; k < 4; k++) e3 |= (uint32_t)p[12 + k] << (8 * k);
    sky_u64 bad3 = sky_ballot(v32 != e32 || v64 != e64 || v128.x != e32 || v128.w != e3);
    // 4: unaligned stores
    uint8_t* q = wbuf + 1 + lane * 19;
    sky_st128u(q, v128); sky_st16u(q + 16, 0xBEEFu);
    __threadfence_block();
    bool okst = true;
    for (int k = 0; k < 16; k++) okst &= (q[k] == p[k]);
    okst &= (q[16] == 0xEF && q[17] == 0xBE);
    sky_u64 bad4 = sky_ballot(!okst);
    // 5: shfl
    uint32_t s = sky_shfl((uint32_t)lane + 100u, (lane * 7 + 3) & 63);
    sky_u64 bad5 = sky_ballot(s != (uint32_t)((lane * 7 + 3) & 63) + 100u);
    if (lane == 0) {
        results[0] = bad1 != 0; results[1] = bad2; results[2] = bad3 != 0; results[3] = bad4 != 0; results[4] = bad5 != 0;
    }
}

extern "C" __global__ void __launch_bounds__(64) sky_probe_kernel(const uint8_t* p, uint32_t* out) {
    if (sky_lane() == 0) *out = *(const volatile uint8_t*)p;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct EvPair { hipEvent_t a, b; int kind; };
enum { K_LZ4 = 0, K_LAYOUT, K_GATHER, K_MD5, K_CDC, K_N };

// ---- guard-band device allocations (test instrumentation: tests/test_gpu_guard.py) ----------------------------------------------------------------
// The device twin of the emulator's PROT_NONE fences (tests/test_emu_guard.py): a buffer is placed with the HIP virtual-memory API so that it ENDS on the
// last mapped byte before an unmapped address range (or starts right behind one); a kernel that reads or writes one byte too far then dies with a memory
// access fault instead of quietly reading its neighbour -- which is what a hipMalloc'ed buffer lets it do.  With SKYHIP_GUARD_ALLOC=1 in the environment
// every device buffer the library allocates for itself (staging areas, block scratch, metadata, decoder tables) is placed that way, with NO slack behind
// what was asked for; skyhip_debug_guard_alloc hands the same kind of buffer to a test for the caller-owned side (d_in / d_out).  Never set in production:
// every ensure() that changes a size re-maps.
struct SkyGuardBlock { void* user; void* va; size_t va_size; void* map_at; size_t map_size; hipMemGenericAllocationHandle_t h; };
static std::mutex g_guard_mu;
static std::vector<SkyGuardBlock> g_guard_blocks;
static bool sky_guard_on() {
    static const bool on = [] { const char* e = getenv("SKYHIP_GUARD_ALLOC"); return e && atoi(e) > 0; }();
    return on;
}
// at_end: the buffer's last byte is the last mapped byte (its start is then only as aligned as `bytes` is, rounded down to `align`: up to align - 1 bytes of
// slack behind a buffer whose size is not a multiple of it); otherwise its first byte is the first mapped byte
static hipError_t sky_guard_malloc(void** out, size_t bytes, bool at_end, size_t align) {
    *out = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended)) != hipSuccess) return e;
    if (gran < 4096) gran = 4096;
    SkyGuardBlock b = {};
    b.map_size = ((bytes ? bytes : 1) + gran - 1) / gran * gran;
    b.va_size = b.map_size + 2 * gran;
    if ((e = hipMemAddressReserve(&b.va, b.va_size, gran, nullptr, 0)) != hipSuccess) return e;
    b.map_at = (char*)b.va + gran;
    if ((e = hipMemCreate(&b.h, b.map_size, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(b.va, b.va_size); return e; }
    if ((e = hipMemMap(b.map_at, b.map_size, 0, b.h, 0)) != hipSuccess) { (void)hipMemRelease(b.h); (void)hipMemAddressFree(b.va, b.va_size); return e; }
    hipMemAccessDesc ad = {};
    ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(b.map_at, b.map_size, &ad, 1)) != hipSuccess) {
        (void)hipMemUnmap(b.map_at, b.map_size); (void)hipMemRelease(b.h); (void)hipMemAddressFree(b.va, b.va_size); return e;
    }
    b.user = at_end ? (void*)(((uintptr_t)b.map_at + b.map_size - bytes) & ~(uintptr_t)(align ? align - 1 : 0)) : b.map_at;
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard_blocks.push_back(b);
    *out = b.user;
    return hipSuccess;
}
static hipError_t sky_guard_free(void* p) {
    SkyGuardBlock b = {};
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        size_t i = 0;
        for (; i < g_guard_blocks.size(); i++) if (g_guard_blocks[i].user == p) break;
        if (i == g_guard_blocks.size()) return hipErrorInvalidValue;
        b = g_guard_blocks[i];
        g_guard_blocks.erase(g_guard_blocks.begin() + (long)i);
    }
    (void)hipDeviceSynchronize();
    hipError_t e = hipMemUnmap(b.map_at, b.map_size);
    (void)hipMemRelease(b.h);
    // The address range is NOT given back: the next reservation would get the same addresses, and buffers that were unmapped and mapped again at one
    // address within milliseconds read and wrote each other's bytes (GPU call r5c: wrong frames and digests at a different case in every run, no fault) --
    // a translation that outlives its mapping.  A test process reserves a few hundred ranges of a 47-bit space; nothing is lost.
    return e;
}

template <typename T> struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (sky_guard_on()) {      // exactly n elements, the last one against an unmapped page; a new size = a new mapping
            if (n == cap && p) return hipSuccess;
            if (p) (void)sky_guard_free(p);
            p = nullptr; cap = 0;
            hipError_t e = sky_guard_malloc((void**)&p, n * sizeof(T), true, 16);
            if (e == hipSuccess) cap = n;
            return e;
        }
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) { if (sky_guard_on()) (void)sky_guard_free(p); else (void)hipFree(p); } p = nullptr; cap = 0; }
};
template <typename T> struct PinBuf {
    T* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

#ifdef SKY_WITH_CDC
#include "cdc_host.inc"
#endif
#include "lz4d_host.inc"

struct skyhip_ctx {
    int dev = 0;
    size_t max_chunk = 0;
    int max_batch = 0;
    uint32_t blocks_per_chunk = 0;
    hipStream_t s_lz4 = nullptr, s_md5 = nullptr, s_cdc = nullptr, s_fr = nullptr;     // s_fr: frame layout + gather of sub-batch k beside the compressor on k+1
    // per-chunk metadata (whole batch)
    DevBuf<sky_u64> d_in_off, d_out_off, d_frame_len;
    DevBuf<uint32_t> d_in_len, d_blk_prefix;
    DevBuf<uint8_t> d_md5;
    PinBuf<sky_u64> h_in_off, h_out_off, h_frame_len;
    PinBuf<uint32_t> h_in_len, h_blk_prefix;
    PinBuf<uint8_t> h_md5;
    // per-block data for one sub-batch
    // block scratch, double buffered: sub-batch k compresses into buffer k & 1 while the frames of k-1 are assembled out of the other one
    DevBuf<uint8_t> d_scratch[2];
    DevBuf<uint32_t> d_csize[2], d_blk_word[2];
    hipEvent_t ev_lz4_done[2] = {nullptr, nullptr}, ev_fr_done[2] = {nullptr, nullptr};
    DevBuf<uint32_t> d_queue;     // slice-parallel compressor: block queue head
    int md5_wg = 64;              // lanes per MD5 workgroup.  One wave per CU is as fast as MD5 gets: every lane streams its own chunk (64 cache lines per
                                  // load instruction), and a CU's memory path serves one such wave at full chain speed -- 2048 chunks take 98 ms as 64-lane
                                  // workgroups, 187 ms as 256-lane, 374 ms as 512-lane ones (profiles/r2_md5_workgroup.txt).  SKYHIP_MD5_WG overrides.
    bool md5_wg_env = false;      // SKYHIP_MD5_WG given: no automatic choice
    int frames_min = 0;           // a device-resident call with at least this many chunks uses sky_lz4s_frames (one workgroup per chunk at a time)
    int lz4s_grid = 0;            // workgroups of the slice-parallel compressor = CUs of the device (141 KiB of LDS each: one per CU)
    DevBuf<sky_u64> d_blk_dst[2];
    // host-batch staging (skyhip_process_batch): a whole group of chunks resident, copies on their own streams
    DevBuf<uint8_t> d_stage_in, d_stage_out, d_lit;
    DevBuf<sky_u64> d_run_src, d_run_dst; DevBuf<uint32_t> d_run_len;      // skyhip_gather_md5
    DevBuf<sky_u64> d_sv_desc; DevBuf<uint8_t> d_sv_fps; DevBuf<uint32_t> d_sv_total;      // skyhip_segment_md5_device
    int stage_n = 0; size_t stage_in_stride = 0; std::vector<uint64_t> stage_len;      // the chunks skyhip_process_batch left in d_stage_in (skyhip_dedup_literals)
    hipStream_t s_up = nullptr, s_down = nullptr;
    std::vector<hipEvent_t> ev_up;    // upload-complete event per LZ4 sub-batch of a group (grow-only)
    std::vector<void*> host_allocs;   // skyhip_host_alloc'ed blocks still alive (freed by skyhip_destroy at the latest)
    std::vector<void*> host_regs;     // skyhip_host_register'ed ranges of the caller (unregistered by skyhip_destroy at the latest)
#ifdef SKY_WITH_CDC
    SkyCdcState cdc;   // CDC / dedup state
#endif
    SkyLz4dState dec;   // frame decompressor state
    double dec_ms = 0;
    // timing
    std::vector<EvPair> ev_busy, ev_free, ev_open;     // open = begun, not ended yet (returned to ev_free if a call bails out in between)
    skyhip_timing tm;
    char hip_err[256];
    bool counted = false;         // this context is in g_live_contexts
    long fault_after = -1;        // skyhip_debug_fault: the (n+1)-th checked HIP call from now fails artificially (error-path tests); -1 = off
    uint32_t* d_self = nullptr;
    sky_u64* d_prof = nullptr;   // SKY_PROF builds only
};

#define HIPCHK(ctx, expr)                                                                             \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e == hipSuccess && (ctx)->fault_after >= 0 && (ctx)->fault_after-- == 0) _e = hipErrorUnknown;   /* skyhip_debug_fault (tests) */ \
        if (_e != hipSuccess) {                                                                       \
            snprintf((ctx)->hip_err, sizeof((ctx)->hip_err), "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? SKYHIP_E_NOMEM : SKYHIP_E_HIP;                         \
        }                                                                                             \
    } while (0)

static int ev_begin(skyhip_ctx* c, hipStream_t s, int kind, EvPair* out) {
    EvPair p;
    if (!c->ev_free.empty()) { p = c->ev_free.back(); c->ev_free.pop_back(); }
    else { HIPCHK(c, hipEventCreate(&p.a)); HIPCHK(c, hipEventCreate(&p.b)); }
    p.kind = kind;
    c->ev_open.push_back(p);
    HIPCHK(c, hipEventRecord(p.a, s));
    *out = p;
    return 0;
}
static int ev_end(skyhip_ctx* c, hipStream_t s, EvPair& p) {
    HIPCHK(c, hipEventRecord(p.b, s));
    for (size_t i = 0; i < c->ev_open.size(); i++)
        if (c->ev_open[i].a == p.a) { c->ev_open.erase(c->ev_open.begin() + (long)i); break; }
    c->ev_busy.push_back(p);
    return 0;
}
static void ev_collect_free(skyhip_ctx* c) {      // error path: recorded pairs are recycled without being accounted
    for (auto& p : c->ev_busy) c->ev_free.push_back(p);
    c->ev_busy.clear();
}
// scope guard of a call that uses ev_begin/ev_end: whatever is still open when the call returns (an error path between
// the two) is recycled instead of stranded
struct EvOpenGuard {
    skyhip_ctx* c;
    bool ok = false;      // set by the call's successful end
    ~EvOpenGuard() {
        if (!ok) {        // bailing out with work possibly in flight: nothing may still touch the caller's or the context's buffers
            for (hipStream_t st : {c->s_lz4, c->s_fr, c->s_md5, c->s_cdc, c->s_up, c->s_down}) if (st) (void)hipStreamSynchronize(st);
            ev_collect_free(c);
        }
        for (auto& p : c->ev_open) c->ev_free.push_back(p);
        c->ev_open.clear();
    }
};
struct EventOwner {       // a one-off event that must not outlive the call, whichever way the call ends
    hipEvent_t e = nullptr;
    ~EventOwner() { if (e) (void)hipEventDestroy(e); }
};
static void ev_collect(skyhip_ctx* c) {
    for (auto& p : c->ev_busy) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            switch (p.kind) {
                case K_LZ4: c->tm.lz4_ms += ms; break;
                case K_LAYOUT: c->tm.layout_ms += ms; break;
                case K_GATHER: c->tm.gather_ms += ms; break;
                case K_MD5: c->tm.md5_ms += ms; break;
                case K_CDC: c->tm.cdc_ms += ms; break;
            }
        }
        c->ev_free.push_back(p);
    }
    c->ev_busy.clear();
}

extern "C" {

int skyhip_abi_version(void) { return SKYHIP_ABI_VERSION; }

size_t skyhip_frame_bound(size_t raw_len) {
    size_t nblk = (raw_len + SKY_LZ4_BLOCK - 1) / SKY_LZ4_BLOCK;
    return SKY_FRAME_HDR + raw_len + 4 * nblk + 4;
}

const char* skyhip_strerror(int code) {
    switch (code) {
        case SKYHIP_OK: return "ok";
        case SKYHIP_E_INVAL: return "invalid argument";
        case SKYHIP_E_NOMEM: return "out of device or pinned host memory";
        case SKYHIP_E_HIP: return "HIP runtime call or kernel launch failed";
        case SKYHIP_E_TOOBIG: return "chunk or batch larger than the context was created for";
        case SKYHIP_E_CAP: return "output buffer smaller than skyhip_frame_bound / cut capacity";
        case SKYHIP_E_NODEVICE: return "no usable gfx950 device";
        case SKYHIP_E_TABLEFULL: return "dedup table full";
        case SKYHIP_E_FORMAT: return "malformed or unsupported LZ4 frame";
        default: return "unknown skyhip error";
    }
}

static thread_local char g_create_err[256] = "";     // what went wrong in the last failed skyhip_create of this thread (there is no context to ask)
const char* skyhip_last_hip_error(skyhip_ctx* ctx) { return ctx ? ctx->hip_err : g_create_err; }

// Contexts alive in this process.  Every context brings six HIP streams, a process's streams share GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of
// streams that share a queue run one after the other: measured fine for two contexts per process at the default (profiles/r5_operator_lanes.txt: 33.6 / 44.0 /
// 38.1 Gbit/s through the loopback with 1 / 2 / 3 lanes per worker), bimodal at 8-32 queues (profiles/r5_cdc.txt: 148 / 163 / 252 / 257 ms per step in one call).
// Nothing can pick a stream's queue from here, but a deployment that lands in a configuration that measured badly is told so, once.
static std::atomic<int> g_live_contexts{0};
static std::atomic<int> g_queue_note_said{0};
static void sky_queue_note(int live) {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const int q = (e && atoi(e) > 0) ? atoi(e) : 4;
    const bool odd_queues = q != 4 && live >= 2, many = live >= 3;
    if (!(odd_queues || many) || getenv("SKYHIP_QUIET")) return;
    int said = 0;
    if (!g_queue_note_said.compare_exchange_strong(said, 1)) return;
    fprintf(stderr, "[skyhip] pid %d: %d contexts = %d streams on %d hardware queues (GPU_MAX_HW_QUEUES%s): kernels of streams that share a queue run one after the "
                    "other.  Measured: two contexts per process at the default 4 queues is the stable configuration; %s (profiles/r5_operator_lanes.txt, r5_cdc.txt).  "
                    "SKYHIP_QUIET=1 silences this note.\n",
            (int)getpid(), live, 6 * live, q, e ? " from the environment" : " default",
            odd_queues ? "8-32 queues showed a second, 1.7 x slower mode for compressor + CDC launches" : "a third context mostly queues behind the other two's digest launches");
}

int skyhip_create(int device_id, size_t max_chunk_bytes, int max_batch, skyhip_ctx** out) {
    if (!out || max_batch <= 0 || max_chunk_bytes == 0 || max_chunk_bytes > ((size_t)1 << 30)) return SKYHIP_E_INVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SKYHIP_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return SKYHIP_E_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SKYHIP_E_NODEVICE;   // this library carries gfx950 code only
    skyhip_ctx* c = new (std::nothrow) skyhip_ctx();
    if (!c) return SKYHIP_E_NOMEM;
    c->hip_err[0] = 0;
    memset(&c->tm, 0, sizeof c->tm);
    c->dev = device_id; c->max_chunk = max_chunk_bytes; c->max_batch = max_batch;
    c->blocks_per_chunk = (uint32_t)((max_chunk_bytes + SKY_LZ4_BLOCK - 1) / SKY_LZ4_BLOCK);
    int rc = [&]() -> int {
        HIPCHK(c, hipSetDevice(device_id));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_lz4, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_fr, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) { HIPCHK(c, hipEventCreateWithFlags(&c->ev_lz4_done[k], hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_fr_done[k], hipEventDisableTiming)); }
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_md5, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_cdc, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_down, hipStreamNonBlocking));
        // (the block scratch of the block-queue path -- 8.06 MiB per chunk of max_batch, twice -- is allocated by the first call that takes that path:
        // a context that only ever sees large device-resident batches writes its frames in place and never needs it)
        HIPCHK(c, hipFuncSetAttribute((const void*)sky_lz4s_compress, hipFuncAttributeMaxDynamicSharedMemorySize, LZ4S_LDS_ORIGIN + LZ4S_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute((const void*)sky_lz4_link, hipFuncAttributeMaxDynamicSharedMemorySize, SKY_LZ4D_LINK_LDS));
        HIPCHK(c, hipFuncSetAttribute((const void*)sky_lz4_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, SKY_LZ4R_LDS));
        HIPCHK(c, hipFuncSetAttribute((const void*)sky_lz4_chain, hipFuncAttributeMaxDynamicSharedMemorySize, SKY_LZ4C_LDS));
        c->lz4s_grid = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char* e = getenv("SKYHIP_LZ4S_GRID")) { const int v = atoi(e); if (v > 0) c->lz4s_grid = v; }
        // frames in place need a chunk per CU (and then some, for balance) to fill the chip; below that the block queue + gather keeps every CU busy
        c->frames_min = 2 * c->lz4s_grid;
        if (const char* e = getenv("SKYHIP_FRAMES_MIN")) { const int v = atoi(e); if (v >= 0) c->frames_min = v; }
        HIPCHK(c, hipFuncSetAttribute((const void*)sky_lz4s_frames, hipFuncAttributeMaxDynamicSharedMemorySize, LZ4S_LDS_ORIGIN + LZ4S_LDS_BYTES));
        HIPCHK(c, c->d_queue.ensure(16));
        { const char* e = getenv("SKYHIP_MD5_WG"); const int v = e ? atoi(e) : 0; if (v >= 64 && v <= 256 && v % 64 == 0) { c->md5_wg = v; c->md5_wg_env = true; } }
#ifdef SKY_WITH_CDC
        c->cdc.segmd5_grid = (uint32_t)c->lz4s_grid * 32u;
        c->cdc.gear_grid_beside = c->cdc.gear_grid; c->cdc.segmd5_grid_beside = c->cdc.segmd5_grid;      // (see sky_cdc_run: bounded grids were measured and are NOT the default)
        if (const char* e = getenv("SKYHIP_GEAR_BESIDE")) { const int v = atoi(e); if (v > 0) c->cdc.gear_grid_beside = (uint32_t)c->lz4s_grid * (uint32_t)v; }        // (tuning: wavefronts per CU)
        if (const char* e = getenv("SKYHIP_SEGMD5_BESIDE")) { const int v = atoi(e); if (v > 0) c->cdc.segmd5_grid_beside = (uint32_t)c->lz4s_grid * (uint32_t)v; }
        c->cdc.gear_grid = (uint32_t)c->lz4s_grid * 32u;      // wavefronts of sky_gear_candidates: what the chip holds when the kernel has it to itself
#endif
        if (getenv("SKYHIP_DEBUG")) {
            int nbs = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbs, (const void*)sky_lz4s_compress, LZ4S_LANES, LZ4S_LDS_ORIGIN + LZ4S_LDS_BYTES);
            hipFuncAttributes fs;
            (void)hipFuncGetAttributes(&fs, (const void*)sky_lz4s_compress);
            fprintf(stderr, "[skyhip] sky_lz4s_compress: %d workgroups/CU (x16 waves), %d VGPRs, %u B dynamic LDS, grid %d\n", nbs, fs.numRegs, (unsigned)LZ4S_LDS_BYTES, c->lz4s_grid);
        }
        return 0;
    }();
    if (rc) { snprintf(g_create_err, sizeof g_create_err, "%s", c->hip_err); skyhip_destroy(c); return rc; }
    g_create_err[0] = 0;
    *out = c;
    c->counted = true;
    sky_queue_note(++g_live_contexts);
    return SKYHIP_OK;
}

// ---- one whole-chip compressor launch at a time per device --------------------------------------------------------------------------------
// sky_lz4s_frames takes every CU's LDS: two of them (two contexts of one process, e.g. the lanes of an operator or the steps bench.py keeps in
// flight) cannot share the chip, the second one's workgroups only trickle in as the first one's exit.  So a context makes its stream wait for the
// previous such launch of ANY context on the device before it records its own start event: the launches run back to back, and the event pair around
// each one times the kernel, not the queueing.  (Streams of different contexts are ordered through an event; nothing blocks on the host.)
static std::mutex g_frames_mu;
static hipEvent_t g_frames_last[64] = {};        // per device: completion event of the most recent sky_lz4s_frames launch, owned by its context
static skyhip_ctx* g_frames_owner[64] = {};

void skyhip_destroy(skyhip_ctx* c) {
    if (!c) return;
    if (c->counted) { c->counted = false; --g_live_contexts; }
    (void)hipSetDevice(c->dev);
    {
        std::lock_guard<std::mutex> lk(g_frames_mu);
        if (c->dev >= 0 && c->dev < 64 && g_frames_owner[c->dev] == c) { g_frames_owner[c->dev] = nullptr; g_frames_last[c->dev] = nullptr; }
    }
    if (c->s_lz4) (void)hipStreamSynchronize(c->s_lz4);
    if (c->s_fr) (void)hipStreamSynchronize(c->s_fr);
    if (c->s_md5) (void)hipStreamSynchronize(c->s_md5);
    if (c->s_cdc) (void)hipStreamSynchronize(c->s_cdc);
    if (c->s_up) (void)hipStreamSynchronize(c->s_up);
    if (c->s_down) (void)hipStreamSynchronize(c->s_down);
    ev_collect(c);
    for (auto& p : c->ev_open) c->ev_free.push_back(p);
    c->ev_open.clear();
    for (auto& p : c->ev_free) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->d_in_off.release(); c->d_out_off.release(); c->d_frame_len.release(); c->d_in_len.release(); c->d_blk_prefix.release(); c->d_md5.release();
    c->h_in_off.release(); c->h_out_off.release(); c->h_frame_len.release(); c->h_in_len.release(); c->h_blk_prefix.release(); c->h_md5.release();
    for (int k = 0; k < 2; k++) {
        c->d_scratch[k].release(); c->d_csize[k].release(); c->d_blk_word[k].release(); c->d_blk_dst[k].release();
        if (c->ev_lz4_done[k]) (void)hipEventDestroy(c->ev_lz4_done[k]);
        if (c->ev_fr_done[k]) (void)hipEventDestroy(c->ev_fr_done[k]);
    }
    c->d_queue.release();
    c->d_stage_in.release(); c->d_stage_out.release(); c->d_lit.release(); c->d_run_src.release(); c->d_run_dst.release(); c->d_run_len.release();
    for (hipEvent_t e : c->ev_up) (void)hipEventDestroy(e);
    c->ev_up.clear();
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    c->host_allocs.clear();
    for (void* p : c->host_regs) (void)hipHostUnregister(p);
    c->host_regs.clear();
    c->dec.release();
#ifdef SKY_WITH_CDC
    sky_cdc_state_release(&c->cdc);
#endif
    if (c->d_self) (void)hipFree(c->d_self);
    if (c->d_prof) (void)hipFree(c->d_prof);
    if (c->s_lz4) (void)hipStreamDestroy(c->s_lz4);
    if (c->s_fr) (void)hipStreamDestroy(c->s_fr);
    if (c->s_md5) (void)hipStreamDestroy(c->s_md5);
    if (c->s_cdc) (void)hipStreamDestroy(c->s_cdc);
    if (c->s_up) (void)hipStreamDestroy(c->s_up);
    if (c->s_down) (void)hipStreamDestroy(c->s_down);
    delete c;
}

void skyhip_get_timing(skyhip_ctx* c, skyhip_timing* out) { if (c && out) *out = c->tm; }
void skyhip_reset_timing(skyhip_ctx* c) { if (c) memset(&c->tm, 0, sizeof c->tm); }

// Hooks that turn the device-resident call into the middle of the host-buffer pipeline (skyhip_process_batch):
// the chunks of LZ4 sub-batch s become resident when ev_up[s] fires (H2D copies queued on the context's upload stream),
// and as soon as a sub-batch's frames are laid out they start travelling to host_out[] on the download stream, while
// later sub-batches still upload / compress and the one whole-batch MD5 launch runs beside all of it.
struct SkyPipe {
    const hipEvent_t* ev_up;     // [number of LZ4 sub-batches]; recorded in order on one stream
    int n_sub;
    uint8_t* const* host_out;    // [n] destination of each frame (null when SKYHIP_F_LZ4 is not set)
    size_t* host_out_len;        // [n]
};

static int sky_process_impl(skyhip_ctx* c, int n, const void* d_in, const uint64_t* in_off, const uint64_t* in_len, void* d_out,
                            const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, uint8_t (*md5)[16], uint32_t flags,
                            const SkyPipe* pipe) {
    if (!c || n < 0 || (n > 0 && (!d_in || !in_off || !in_len))) return SKYHIP_E_INVAL;
    if ((flags & SKYHIP_F_LZ4) && n > 0 && (!d_out || !out_off || !out_cap)) return SKYHIP_E_INVAL;
    if ((flags & SKYHIP_F_DEDUP) && !(flags & SKYHIP_F_CDC)) return SKYHIP_E_INVAL;
    if (n == 0) return SKYHIP_OK;
    HIPCHK(c, hipSetDevice(c->dev));
    const size_t N = (size_t)n;
    HIPCHK(c, c->h_in_off.ensure(N)); HIPCHK(c, c->h_out_off.ensure(N)); HIPCHK(c, c->h_frame_len.ensure(N));
    HIPCHK(c, c->h_in_len.ensure(N)); HIPCHK(c, c->h_blk_prefix.ensure(N + 1)); HIPCHK(c, c->h_md5.ensure(16 * N));
    HIPCHK(c, c->d_in_off.ensure(N)); HIPCHK(c, c->d_out_off.ensure(N)); HIPCHK(c, c->d_frame_len.ensure(N));
    HIPCHK(c, c->d_in_len.ensure(N)); HIPCHK(c, c->d_blk_prefix.ensure(N + 1)); HIPCHK(c, c->d_md5.ensure(16 * N));
    uint64_t nblk_total = 0, bytes_total = 0;
    for (size_t i = 0; i < N; i++) {
        if (in_len[i] > c->max_chunk) return SKYHIP_E_TOOBIG;
        if ((flags & SKYHIP_F_LZ4) && out_cap[i] < skyhip_frame_bound((size_t)in_len[i])) return SKYHIP_E_CAP;
        c->h_in_off.p[i] = in_off[i];
        c->h_in_len.p[i] = (uint32_t)in_len[i];
        c->h_out_off.p[i] = (flags & SKYHIP_F_LZ4) ? out_off[i] : 0;
        c->h_blk_prefix.p[i] = (uint32_t)nblk_total;
        nblk_total += (in_len[i] + SKY_LZ4_BLOCK - 1) / SKY_LZ4_BLOCK;
        bytes_total += in_len[i];
        if (nblk_total > 0xFFFFFFF0ull) return SKYHIP_E_TOOBIG;
    }
    c->h_blk_prefix.p[N] = (uint32_t)nblk_total;
    // metadata upload on the lz4 stream; the other streams wait on an event
    HIPCHK(c, hipMemcpyAsync(c->d_in_off.p, c->h_in_off.p, N * 8, hipMemcpyHostToDevice, c->s_lz4));
    HIPCHK(c, hipMemcpyAsync(c->d_in_len.p, c->h_in_len.p, N * 4, hipMemcpyHostToDevice, c->s_lz4));
    HIPCHK(c, hipMemcpyAsync(c->d_out_off.p, c->h_out_off.p, N * 8, hipMemcpyHostToDevice, c->s_lz4));
    HIPCHK(c, hipMemcpyAsync(c->d_blk_prefix.p, c->h_blk_prefix.p, (N + 1) * 4, hipMemcpyHostToDevice, c->s_lz4));
    EvOpenGuard open_guard{c};
    EventOwner meta_owner;
    HIPCHK(c, hipEventCreateWithFlags(&meta_owner.e, hipEventDisableTiming));
    const hipEvent_t meta_ready = meta_owner.e;
    HIPCHK(c, hipEventRecord(meta_ready, c->s_lz4));

    int rc = 0;
    std::vector<hipEvent_t> ev_done;     // pipe mode: sub-batch s laid out and its frame lengths on the host
    struct EvGuard { std::vector<hipEvent_t>& v; ~EvGuard() { for (hipEvent_t e : v) (void)hipEventDestroy(e); } } ev_guard{ev_done};
    // ---- MD5 first: few, long-running waves; they become resident and the LZ4 grid fills the rest ----
    if (flags & SKYHIP_F_MD5) {
        HIPCHK(c, hipStreamWaitEvent(c->s_md5, meta_ready, 0));
        if (pipe) HIPCHK(c, hipStreamWaitEvent(c->s_md5, pipe->ev_up[pipe->n_sub - 1], 0));   // needs every chunk resident
        SkyMd5Args ma;
        ma.in = (const uint8_t*)d_in; ma.off = c->d_in_off.p; ma.len = c->d_in_len.p; ma.n = (uint32_t)N; ma.digest = c->d_md5.p;
        EvPair ep;
        if ((rc = ev_begin(c, c->s_md5, K_MD5, &ep))) return rc;
        // one wave per workgroup: a CU's memory path serves ONE such wave at full chain speed (82 ms per 8 MiB; 94 ms with two, profiles/r2_md5_workgroup.txt).
        // Round 2 switched to two waves per workgroup above 4096 chunks to leave more CUs free of MD5; measured again in round 3 (GPU call r3n) the step
        // is 1.9 % shorter with one (177.95 against 181.3 ms for 8192 chunks) and equal for 16384: the chain, not the co-residency, is what to protect.
        const int wg = c->md5_wg;
        hipLaunchKernelGGL(sky_md5_chunks, dim3((unsigned)((N + (size_t)wg - 1) / (size_t)wg)), dim3((unsigned)wg), 0, c->s_md5, ma);
        HIPCHK(c, hipGetLastError());
        if ((rc = ev_end(c, c->s_md5, ep))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_md5.p, c->d_md5.p, 16 * N, hipMemcpyDeviceToHost, c->s_md5));
        c->tm.md5_launches++; c->tm.md5_in_bytes += bytes_total;
    }
    const bool frames_in_place = (flags & SKYHIP_F_LZ4) && !pipe && c->frames_min > 0 && N >= (size_t)c->frames_min;
    // ---- CDC (+ fingerprints, dedup) on its own stream, beside the compressor (cdc_host.inc: the grids are sized for that) ----
    if (flags & SKYHIP_F_CDC) {
#ifndef SKY_WITH_CDC
        return SKYHIP_E_INVAL;
#else
        HIPCHK(c, hipStreamWaitEvent(c->s_cdc, meta_ready, 0));
        if (pipe) HIPCHK(c, hipStreamWaitEvent(c->s_cdc, pipe->ev_up[pipe->n_sub - 1], 0));
        EvPair ep;
        if ((rc = ev_begin(c, c->s_cdc, K_CDC, &ep))) return rc;
        uint64_t in_end = 0;      // one past the last input byte of the call: what the staged segment digests may read up to
        for (size_t i = 0; i < N; i++) if (c->h_in_off.p[i] + c->h_in_len.p[i] > in_end) in_end = c->h_in_off.p[i] + c->h_in_len.p[i];
        rc = sky_cdc_run(&c->cdc, c->s_cdc, (const uint8_t*)d_in, c->d_in_off.p, c->d_in_len.p, c->h_in_len.p, (uint32_t)N,
                         (flags & SKYHIP_F_DEDUP) != 0, c->hip_err, sizeof c->hip_err, frames_in_place, in_end);
        if (rc) return rc;
        if ((rc = ev_end(c, c->s_cdc, ep))) return rc;
#endif
    }
    // ---- LZ4: sub-batches of max_batch chunks; the compressor (s_lz4) writes block scratch k & 1 while the frames of sub-batch k-1 are laid out
    //      and gathered out of the other buffer on s_fr ----
    if (frames_in_place) {
        // ---- LZ4, large device-resident batch: ONE launch, a workgroup per chunk at a time, every frame written in place (no scratch, no gather) ----
        SkyLz4FArgs fa;
        fa.in = (const uint8_t*)d_in; fa.in_off = c->d_in_off.p; fa.in_len = c->d_in_len.p; fa.n_chunks = (uint32_t)N;
        fa.out = (uint8_t*)d_out; fa.out_off = c->d_out_off.p; fa.frame_len = c->d_frame_len.p; fa.prof = nullptr; fa.queue = c->d_queue.p;
#if SKY_PROF
        if (!c->d_prof) { HIPCHK(c, hipMalloc((void**)&c->d_prof, 64 * 16 * 8)); HIPCHK(c, hipMemset(c->d_prof, 0, 64 * 16 * 8)); }
        fa.prof = c->d_prof;
#endif
        HIPCHK(c, hipMemsetAsync(c->d_queue.p, 0, 4, c->s_lz4));     // chunk queue head
        EvPair ep;
        {
            std::lock_guard<std::mutex> lk(g_frames_mu);          // (see g_frames_last: launches of all contexts on this device run back to back)
            const int dv = c->dev >= 0 && c->dev < 64 ? c->dev : 0;
            if (g_frames_last[dv] && g_frames_owner[dv] != c) HIPCHK(c, hipStreamWaitEvent(c->s_lz4, g_frames_last[dv], 0));
            if ((rc = ev_begin(c, c->s_lz4, K_LZ4, &ep))) return rc;
            hipLaunchKernelGGL(sky_lz4s_frames, dim3(N < (size_t)c->lz4s_grid ? (unsigned)N : (unsigned)c->lz4s_grid), dim3(LZ4S_LANES), LZ4S_LDS_ORIGIN + LZ4S_LDS_BYTES, c->s_lz4, fa);
            HIPCHK(c, hipGetLastError());
            if ((rc = ev_end(c, c->s_lz4, ep))) return rc;
            HIPCHK(c, hipEventRecord(c->ev_lz4_done[0], c->s_lz4));
            g_frames_last[dv] = c->ev_lz4_done[0]; g_frames_owner[dv] = c;
        }
        c->tm.lz4_launches++; c->tm.lz4_in_bytes += bytes_total;
        HIPCHK(c, hipMemcpyAsync(c->h_frame_len.p, c->d_frame_len.p, N * 8, hipMemcpyDeviceToHost, c->s_lz4));
    } else if (flags & SKYHIP_F_LZ4) {
        {
            const size_t nbmax = (size_t)c->max_batch * c->blocks_per_chunk;
            for (int k = 0; k < 2; k++) {
                HIPCHK(c, c->d_scratch[k].ensure(nbmax * SKY_LZ4_SLOT));
                HIPCHK(c, c->d_csize[k].ensure(nbmax));
                HIPCHK(c, c->d_blk_word[k].ensure(nbmax));
                HIPCHK(c, c->d_blk_dst[k].ensure(nbmax));
            }
        }
        size_t sub = 0;
        for (size_t c0 = 0; c0 < N; c0 += (size_t)c->max_batch, sub++) {
            const int bf = (int)(sub & 1);
            const size_t nc = (N - c0 < (size_t)c->max_batch) ? N - c0 : (size_t)c->max_batch;
            const uint32_t nb = c->h_blk_prefix.p[c0 + nc] - c->h_blk_prefix.p[c0];
            if (pipe) HIPCHK(c, hipStreamWaitEvent(c->s_lz4, pipe->ev_up[c0 / (size_t)c->max_batch], 0));
            uint64_t sub_bytes = 0;
            for (size_t i = c0; i < c0 + nc; i++) sub_bytes += c->h_in_len.p[i];
            SkyLz4Args la;
            la.in = (const uint8_t*)d_in; la.in_off = c->d_in_off.p + c0; la.in_len = c->d_in_len.p + c0; la.blk_prefix = c->d_blk_prefix.p + c0;
            la.n_chunks = (uint32_t)nc; la.n_blocks = nb; la.scratch = c->d_scratch[bf].p; la.csize = c->d_csize[bf].p;
            la.ablate = 0;
#if SKY_ABL
            { const char* ab = getenv("SKYHIP_ABLATE"); la.ablate = ab ? (uint32_t)atoi(ab) : 0u; }   // timing-experiment builds only
#endif
            la.prof = nullptr;
            la.queue = c->d_queue.p;
#if SKY_PROF
            if (!c->d_prof) { HIPCHK(c, hipMalloc((void**)&c->d_prof, 64 * 16 * 8)); HIPCHK(c, hipMemset(c->d_prof, 0, 64 * 16 * 8)); }
            la.prof = c->d_prof;
#endif
            SkyFrameArgs fa;
            fa.in = la.in; fa.in_off = la.in_off; fa.in_len = la.in_len; fa.blk_prefix = la.blk_prefix; fa.n_chunks = la.n_chunks; fa.n_blocks = nb;
            fa.scratch = c->d_scratch[bf].p; fa.csize = c->d_csize[bf].p; fa.out = (uint8_t*)d_out; fa.out_off = c->d_out_off.p + c0;
            fa.frame_len = c->d_frame_len.p + c0; fa.blk_dst = c->d_blk_dst[bf].p; fa.blk_word = c->d_blk_word[bf].p;
            EvPair ep;
            if (nb) {
                if (sub >= 2) HIPCHK(c, hipStreamWaitEvent(c->s_lz4, c->ev_fr_done[bf], 0));     // the frames of sub-batch sub-2 have left this buffer
                HIPCHK(c, hipMemsetAsync(c->d_queue.p, 0, 4, c->s_lz4));     // block queue head
                if ((rc = ev_begin(c, c->s_lz4, K_LZ4, &ep))) return rc;
                hipLaunchKernelGGL(sky_lz4s_compress, dim3(nb < (uint32_t)c->lz4s_grid ? nb : (uint32_t)c->lz4s_grid), dim3(LZ4S_LANES), LZ4S_LDS_ORIGIN + LZ4S_LDS_BYTES, c->s_lz4, la);
                HIPCHK(c, hipGetLastError());
                if ((rc = ev_end(c, c->s_lz4, ep))) return rc;
                c->tm.lz4_launches++; c->tm.lz4_in_bytes += sub_bytes;
            }
            HIPCHK(c, hipEventRecord(c->ev_lz4_done[bf], c->s_lz4));
            HIPCHK(c, hipStreamWaitEvent(c->s_fr, c->ev_lz4_done[bf], 0));      // (also orders s_fr behind the metadata upload on s_lz4)
            if ((rc = ev_begin(c, c->s_fr, K_LAYOUT, &ep))) return rc;
            hipLaunchKernelGGL(sky_frame_layout, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, c->s_fr, fa);
            HIPCHK(c, hipGetLastError());
            if ((rc = ev_end(c, c->s_fr, ep))) return rc;
            if (nb) {
                if ((rc = ev_begin(c, c->s_fr, K_GATHER, &ep))) return rc;
                hipLaunchKernelGGL(sky_frame_gather, dim3(nb), dim3(256), 0, c->s_fr, fa);
                HIPCHK(c, hipGetLastError());
                if ((rc = ev_end(c, c->s_fr, ep))) return rc;
            }
            HIPCHK(c, hipEventRecord(c->ev_fr_done[bf], c->s_fr));
            if (pipe) {
                HIPCHK(c, hipMemcpyAsync(c->h_frame_len.p + c0, c->d_frame_len.p + c0, nc * 8, hipMemcpyDeviceToHost, c->s_fr));
                hipEvent_t e;
                HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ev_done.push_back(e);
                HIPCHK(c, hipEventRecord(e, c->s_fr));
            }
        }
        if (!pipe) HIPCHK(c, hipMemcpyAsync(c->h_frame_len.p, c->d_frame_len.p, N * 8, hipMemcpyDeviceToHost, c->s_fr));
        // pipe mode: ship each sub-batch's frames as soon as their lengths are known; later sub-batches keep computing
        for (size_t sidx = 0; sidx < ev_done.size(); sidx++) {
            HIPCHK(c, hipEventSynchronize(ev_done[sidx]));
            const size_t c0 = sidx * (size_t)c->max_batch, nc = (N - c0 < (size_t)c->max_batch) ? N - c0 : (size_t)c->max_batch;
            for (size_t i = c0; i < c0 + nc; i++) {
                const sky_u64 fl = c->h_frame_len.p[i];
                pipe->host_out_len[i] = (size_t)fl;
                HIPCHK(c, hipMemcpyAsync(pipe->host_out[i], (const uint8_t*)d_out + out_off[i], fl, hipMemcpyDeviceToHost, c->s_down));
            }
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->s_lz4));
    HIPCHK(c, hipStreamSynchronize(c->s_fr));
    HIPCHK(c, hipStreamSynchronize(c->s_md5));
    HIPCHK(c, hipStreamSynchronize(c->s_cdc));
    ev_collect(c);
#ifdef SKY_WITH_CDC
    if (flags & SKYHIP_F_CDC) { int frc = sky_cdc_finish(&c->cdc); if (frc) return frc; }
#endif
    if (flags & SKYHIP_F_LZ4) {
        for (size_t i = 0; i < N; i++) {
            c->tm.lz4_out_bytes += c->h_frame_len.p[i];
            if (out_len) out_len[i] = c->h_frame_len.p[i];
        }
    }
    if ((flags & SKYHIP_F_MD5) && md5) memcpy(md5, c->h_md5.p, 16 * N);
    open_guard.ok = true;
    return SKYHIP_OK;
}

int skyhip_process_device(skyhip_ctx* c, int n, const void* d_in, const uint64_t* in_off, const uint64_t* in_len, void* d_out,
                          const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, uint8_t (*md5)[16], uint32_t flags) {
    return sky_process_impl(c, n, d_in, in_off, in_len, d_out, out_off, out_cap, out_len, md5, flags, nullptr);
}

int skyhip_host_alloc(skyhip_ctx* c, size_t bytes, void** out) {
    if (!c || !out || bytes == 0) return SKYHIP_E_INVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->dev));
    void* p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    c->host_allocs.push_back(p);
    *out = p;
    return SKYHIP_OK;
}

int skyhip_host_free(skyhip_ctx* c, void* p) {
    if (!c) return SKYHIP_E_INVAL;
    if (!p) return SKYHIP_OK;
    for (size_t i = 0; i < c->host_allocs.size(); i++) {
        if (c->host_allocs[i] == p) {
            c->host_allocs.erase(c->host_allocs.begin() + (long)i);
            HIPCHK(c, hipSetDevice(c->dev));
            HIPCHK(c, hipStreamSynchronize(c->s_up));      // no copy of ours may still be reading / writing it
            HIPCHK(c, hipStreamSynchronize(c->s_down));
            HIPCHK(c, hipHostFree(p));
            return SKYHIP_OK;
        }
    }
    return SKYHIP_E_INVAL;   // not one of ours
}

int skyhip_host_register(skyhip_ctx* c, void* p, size_t bytes) {
    if (!c || !p || bytes == 0) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
    c->host_regs.push_back(p);
    return SKYHIP_OK;
}

int skyhip_host_unregister(skyhip_ctx* c, void* p) {
    if (!c) return SKYHIP_E_INVAL;
    if (!p) return SKYHIP_OK;
    for (size_t i = 0; i < c->host_regs.size(); i++) {
        if (c->host_regs[i] == p) {
            c->host_regs.erase(c->host_regs.begin() + (long)i);
            HIPCHK(c, hipSetDevice(c->dev));
            HIPCHK(c, hipStreamSynchronize(c->s_up));      // no copy of ours may still be reading / writing it
            HIPCHK(c, hipStreamSynchronize(c->s_down));
            HIPCHK(c, hipHostUnregister(p));
            return SKYHIP_OK;
        }
    }
    return SKYHIP_E_INVAL;   // not one of ours
}

// Host-buffer batch.  The chunks of a call are staged in HBM together (up to SKYHIP_STAGE_BYTES, default 32 GiB --
// a sliver of 288 GB; larger calls run as consecutive groups) so that whole-chunk MD5, a ~0.1 s serial chain per chunk
// whatever the batch size, is ONE launch for the group.  Inside a group everything is a pipeline: H2D copies are queued
// per LZ4 sub-batch on s_up, the compressor starts on sub-batch s when its upload event fires, and each sub-batch's
// frames start their D2H on s_down as soon as they are laid out (SkyPipe above).  The copies are truly asynchronous
// only for pinned host memory (skyhip_host_alloc, hipHostMalloc/hipHostRegister); pageable buffers work, but the
// runtime then stages them synchronously and the pipeline degenerates to upload-all, compute, download.
int skyhip_process_batch(skyhip_ctx* c, int n, const uint8_t* const* in, const size_t* in_len, uint8_t* const* out, const size_t* out_cap,
                         size_t* out_len, uint8_t (*md5)[16], uint32_t* const* cuts, const size_t* cuts_cap, size_t* n_cuts, uint32_t flags) {
    if (!c || n < 0) return SKYHIP_E_INVAL;
    if (n == 0) return SKYHIP_OK;
    if (!in || !in_len) return SKYHIP_E_INVAL;
    if ((flags & SKYHIP_F_LZ4) && (!out || !out_cap || !out_len)) return SKYHIP_E_INVAL;
    if ((flags & SKYHIP_F_MD5) && !md5) return SKYHIP_E_INVAL;
    if ((flags & SKYHIP_F_CDC) && (!cuts || !cuts_cap || !n_cuts)) return SKYHIP_E_INVAL;
    // validate everything before the first copy is queued: no early return may leave a copy in flight
    size_t longest = 0;
    for (int i = 0; i < n; i++) {
        if (in_len[i] > c->max_chunk) return SKYHIP_E_TOOBIG;
        if (in_len[i] && !in[i]) return SKYHIP_E_INVAL;
        if ((flags & SKYHIP_F_LZ4) && !out[i]) return SKYHIP_E_INVAL;
        if ((flags & SKYHIP_F_LZ4) && out_cap[i] < skyhip_frame_bound(in_len[i])) return SKYHIP_E_CAP;
        if (in_len[i] > longest) longest = in_len[i];
    }
    HIPCHK(c, hipSetDevice(c->dev));
    const size_t in_stride = (longest + 255) & ~(size_t)255;
    const size_t out_stride = (flags & SKYHIP_F_LZ4) ? ((skyhip_frame_bound(longest) + 255) & ~(size_t)255) : 0;
    size_t budget = (size_t)32 << 30;
    if (const char* e = getenv("SKYHIP_STAGE_BYTES")) { const unsigned long long v = strtoull(e, nullptr, 10); if (v) budget = (size_t)v; }
    size_t group = budget / (in_stride + out_stride + 1);
    // the CDC results of a call describe one device launch (see skyhip_cdc_results): keep that contract at max_batch
    if (group < (size_t)c->max_batch || (flags & SKYHIP_F_CDC)) group = (size_t)c->max_batch;
    if (group > (size_t)n) group = (size_t)n;
    HIPCHK(c, c->d_stage_in.ensure(in_stride * group + 256));
    if (flags & SKYHIP_F_LZ4) HIPCHK(c, c->d_stage_out.ensure(out_stride * group + 256));
    const size_t max_sub = (group + (size_t)c->max_batch - 1) / (size_t)c->max_batch;
    while (c->ev_up.size() < max_sub) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_up.push_back(e);
    }
    std::vector<uint64_t> off_in(group), len_in(group), off_out(group), cap_out(group), flen(group);
    for (size_t i = 0; i < group; i++) { off_in[i] = in_stride * i; off_out[i] = out_stride * i; cap_out[i] = out_stride; }

    int rc = SKYHIP_OK;
    hipError_t he = hipSuccess;
    const char* what = "";
    for (size_t c0 = 0; c0 < (size_t)n && rc == SKYHIP_OK && he == hipSuccess; c0 += group) {
        const size_t nc = ((size_t)n - c0 < group) ? (size_t)n - c0 : group;
        const int n_sub = (int)((nc + (size_t)c->max_batch - 1) / (size_t)c->max_batch);
        // generation reuse: the previous group's downloads must have left the staging area
        if (c0) { he = hipStreamSynchronize(c->s_down); if (he != hipSuccess) { what = "hipStreamSynchronize(s_down)"; break; } }
        for (int sb = 0; sb < n_sub && he == hipSuccess; sb++) {
            const size_t lo = (size_t)sb * (size_t)c->max_batch, hi = (lo + (size_t)c->max_batch < nc) ? lo + (size_t)c->max_batch : nc;
            for (size_t i = lo; i < hi && he == hipSuccess; i++) {
                len_in[i] = in_len[c0 + i];
                if (in_len[c0 + i]) {
                    he = hipMemcpyAsync(c->d_stage_in.p + off_in[i], in[c0 + i], in_len[c0 + i], hipMemcpyHostToDevice, c->s_up);
                    if (he != hipSuccess) what = "hipMemcpyAsync(H2D)";
                }
            }
            if (he == hipSuccess) { he = hipEventRecord(c->ev_up[sb], c->s_up); if (he != hipSuccess) what = "hipEventRecord(ev_up)"; }
        }
        if (he != hipSuccess) break;
        SkyPipe pipe;
        pipe.ev_up = c->ev_up.data(); pipe.n_sub = n_sub;
        pipe.host_out = (flags & SKYHIP_F_LZ4) ? out + c0 : nullptr;
        pipe.host_out_len = (flags & SKYHIP_F_LZ4) ? out_len + c0 : nullptr;
        rc = sky_process_impl(c, (int)nc, c->d_stage_in.p, off_in.data(), len_in.data(), (flags & SKYHIP_F_LZ4) ? c->d_stage_out.p : nullptr, off_out.data(),
                              cap_out.data(), flen.data(), md5 ? md5 + c0 : nullptr, flags, &pipe);
#ifdef SKY_WITH_CDC
        if (rc == SKYHIP_OK && (flags & SKYHIP_F_CDC)) rc = sky_cdc_copy_cuts(&c->cdc, (int)nc, cuts + c0, cuts_cap + c0, n_cuts + c0);
#endif
    }
    // the caller's buffers are ours until both copy streams are idle, on every path
    const hipError_t e1 = hipStreamSynchronize(c->s_up), e2 = hipStreamSynchronize(c->s_down);
    if (he == hipSuccess && e1 != hipSuccess) { he = e1; what = "hipStreamSynchronize(s_up)"; }
    if (he == hipSuccess && e2 != hipSuccess) { he = e2; what = "hipStreamSynchronize(s_down)"; }
    c->stage_n = 0;
    if (rc != SKYHIP_OK) return rc;
    if (he != hipSuccess) {
        snprintf(c->hip_err, sizeof(c->hip_err), "%s: %s (process_batch)", what, hipGetErrorString(he));
        return he == hipErrorOutOfMemory ? SKYHIP_E_NOMEM : SKYHIP_E_HIP;
    }
    if ((size_t)n <= group) {      // one group: every chunk of the call is still in the staging area, chunk i at i * in_stride
        c->stage_n = n; c->stage_in_stride = in_stride;
        c->stage_len.assign(in_len, in_len + n);
    }
    return SKYHIP_OK;
}

int skyhip_dedup_literals(skyhip_ctx* c, int n, uint8_t* const* out, const size_t* out_cap, size_t* out_len, size_t* lit_len) {
#ifndef SKY_WITH_CDC
    (void)c; (void)n; (void)out; (void)out_cap; (void)out_len; (void)lit_len;
    return SKYHIP_E_INVAL;
#else
    if (!c || n <= 0 || !out || !out_cap || !out_len || !lit_len) return SKYHIP_E_INVAL;
    // valid only right after the skyhip_process_batch call (CDC + DEDUP, one group) whose chunks these are
    if (c->stage_n != n || (uint32_t)n != c->cdc.last_n || !c->cdc.last_dedup || c->cdc.pending) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    for (int i = 0; i < n; i++) if (!out[i] || out_cap[i] < skyhip_frame_bound((size_t)c->stage_len[i])) return SKYHIP_E_CAP;
    HIPCHK(c, c->d_lit.ensure(c->stage_in_stride * (size_t)n + 256));
    std::vector<uint32_t> h_lit((size_t)n);
    int rc = sky_cdc_literals(&c->cdc, c->s_cdc, c->d_stage_in.p, (uint32_t)n, c->d_lit.p, c->stage_in_stride, h_lit.data(), c->hip_err, sizeof c->hip_err);
    if (rc) return rc;
    // the literal streams that are worth a frame of their own: a chunk without duplicates keeps the frame the first call made (its literal stream IS the chunk)
    std::vector<uint64_t> off, len, ooff, ocap, flen;
    std::vector<int> idx;
    const size_t out_stride = (skyhip_frame_bound(c->stage_in_stride) + 255) & ~(size_t)255;
    for (int i = 0; i < n; i++) {
        lit_len[i] = h_lit[i]; out_len[i] = 0;
        if (h_lit[i] == 0 || h_lit[i] == c->stage_len[i]) continue;
        idx.push_back(i);
        off.push_back(c->stage_in_stride * (uint64_t)i); len.push_back(h_lit[i]);
        ooff.push_back(out_stride * (uint64_t)(idx.size() - 1)); ocap.push_back(out_stride);
    }
    if (idx.empty()) return SKYHIP_OK;
    flen.resize(idx.size());
    HIPCHK(c, c->d_stage_out.ensure(out_stride * idx.size() + 256));
    rc = sky_process_impl(c, (int)idx.size(), c->d_lit.p, off.data(), len.data(), c->d_stage_out.p, ooff.data(), ocap.data(), flen.data(), nullptr, SKYHIP_F_LZ4, nullptr);
    c->stage_n = n;      // (sky_process_impl does not touch the staging area of the chunks)
    if (rc) return rc;
    for (size_t k = 0; k < idx.size(); k++) {
        out_len[idx[k]] = (size_t)flen[k];
        HIPCHK(c, hipMemcpyAsync(out[idx[k]], c->d_stage_out.p + ooff[k], flen[k], hipMemcpyDeviceToHost, c->s_down));
    }
    HIPCHK(c, hipStreamSynchronize(c->s_down));
    return SKYHIP_OK;
#endif
}

int skyhip_decompress_device(skyhip_ctx* c, int n, const void* d_in, const uint64_t* in_off, const uint64_t* in_len, void* d_out,
                             const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status) {
    if (!c || n < 0 || (n > 0 && (!d_in || !in_off || !in_len || !d_out || !out_off || !out_cap))) return SKYHIP_E_INVAL;
    if (n == 0) return SKYHIP_OK;
    HIPCHK(c, hipSetDevice(c->dev));
    return sky_lz4d_run(&c->dec, c->s_lz4, n, d_in, in_off, in_len, d_out, out_off, out_cap, out_len, status, &c->dec_ms, c->hip_err, sizeof c->hip_err);
}

int skyhip_decompress_batch_md5(skyhip_ctx* c, int n, const uint8_t* const* in, const size_t* in_len, uint8_t* const* out, const size_t* out_cap,
                                size_t* out_len, int32_t* status, uint8_t (*md5)[16]) {
    if (!c || n < 0) return SKYHIP_E_INVAL;
    if (n == 0) return SKYHIP_OK;
    if (!in || !in_len || !out || !out_cap || !out_len) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<uint64_t> ioff(n), ilen(n), ooff(n), ocap(n), olen(n);
    uint64_t itot = 0, otot = 0;
    for (int i = 0; i < n; i++) {
        if (in_len[i] && !in[i]) return SKYHIP_E_INVAL;
        if (out_cap[i] && !out[i]) return SKYHIP_E_INVAL;
        if (md5 && out_cap[i] > c->max_chunk) return SKYHIP_E_TOOBIG;        // the digest kernel works on chunks of this context
        ioff[i] = itot; ilen[i] = in_len[i]; itot += (in_len[i] + 255) & ~(uint64_t)255;
        ooff[i] = otot; ocap[i] = out_cap[i]; otot += (out_cap[i] + 255) & ~(uint64_t)255;
    }
    HIPCHK(c, c->dec.d_stage_in.ensure(itot + 256)); HIPCHK(c, c->dec.d_stage_out.ensure(otot + 256));
    // A pipeline over sub-batches of max_batch frames: every upload is queued on s_up at once (one event per sub-batch), the decoder starts on a
    // sub-batch when its frames are resident, and the decoded chunks leave on s_down while later sub-batches still upload and decode.  The digests
    // are one launch at the end: MD5 is a serial chain of ~80 ms per chunk however few chunks a launch holds, so per-sub-batch launches on one
    // stream would only queue behind each other.
    const size_t mb = (size_t)c->max_batch, n_sub = ((size_t)n + mb - 1) / mb;
    while (c->ev_up.size() < n_sub) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_up.push_back(e);
    }
    int rc = SKYHIP_OK;
    hipError_t he = hipSuccess;
    for (size_t sb = 0; sb < n_sub && he == hipSuccess; sb++) {
        const size_t lo = sb * mb, hi = lo + mb < (size_t)n ? lo + mb : (size_t)n;
        for (size_t i = lo; i < hi && he == hipSuccess; i++)
            if (in_len[i]) he = hipMemcpyAsync(c->dec.d_stage_in.p + ioff[i], in[i], in_len[i], hipMemcpyHostToDevice, c->s_up);
        if (he == hipSuccess) he = hipEventRecord(c->ev_up[sb], c->s_up);
    }
    for (size_t sb = 0; sb < n_sub && he == hipSuccess; sb++) {
        const size_t lo = sb * mb, hi = lo + mb < (size_t)n ? lo + mb : (size_t)n;
        he = hipStreamWaitEvent(c->s_lz4, c->ev_up[sb], 0);
        if (he != hipSuccess) break;
        const int src = sky_lz4d_run(&c->dec, c->s_lz4, (int)(hi - lo), c->dec.d_stage_in.p, ioff.data() + lo, ilen.data() + lo, c->dec.d_stage_out.p,
                                     ooff.data() + lo, ocap.data() + lo, olen.data() + lo, status ? status + lo : nullptr, &c->dec_ms, c->hip_err,
                                     sizeof c->hip_err);
        if (src != SKYHIP_OK && src != SKYHIP_E_FORMAT) { rc = src; break; }
        if (src == SKYHIP_E_FORMAT) rc = src;
        for (size_t i = lo; i < hi && he == hipSuccess; i++) {      // sky_lz4d_run returned: these bytes are final
            out_len[i] = (size_t)olen[i];
            if (olen[i]) he = hipMemcpyAsync(out[i], c->dec.d_stage_out.p + ooff[i], olen[i], hipMemcpyDeviceToHost, c->s_down);
        }
    }
    if (md5 && he == hipSuccess && (rc == SKYHIP_OK || rc == SKYHIP_E_FORMAT)) {
        const int mrc = sky_process_impl(c, n, c->dec.d_stage_out.p, ooff.data(), olen.data(), nullptr, nullptr, nullptr, nullptr, md5, SKYHIP_F_MD5, nullptr);
        if (mrc != SKYHIP_OK) rc = mrc;
    }
    const hipError_t e0 = hipStreamSynchronize(c->s_up);
    if (he == hipSuccess) he = e0;
    const hipError_t e1 = hipStreamSynchronize(c->s_lz4), e2 = hipStreamSynchronize(c->s_down);   // the caller's buffers are ours until here
    if (he == hipSuccess) he = e1 != hipSuccess ? e1 : e2;
    if (he != hipSuccess) {
        snprintf(c->hip_err, sizeof(c->hip_err), "%s (decompress_batch)", hipGetErrorString(he));
        return he == hipErrorOutOfMemory ? SKYHIP_E_NOMEM : SKYHIP_E_HIP;
    }
    return rc;
}

int skyhip_decompress_batch(skyhip_ctx* c, int n, const uint8_t* const* in, const size_t* in_len, uint8_t* const* out, const size_t* out_cap,
                            size_t* out_len, int32_t* status) {
    return skyhip_decompress_batch_md5(c, n, in, in_len, out, out_cap, out_len, status, nullptr);
}

int skyhip_dev_alloc(skyhip_ctx* c, size_t bytes, void** out) {
    if (!c || !out || bytes == 0) return SKYHIP_E_INVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->dev));
    if (sky_guard_on()) HIPCHK(c, sky_guard_malloc(out, bytes, true, 256));      // (tests/test_gpu_guard.py: the block ends at an unmapped page)
    else HIPCHK(c, hipMalloc(out, bytes));
    return SKYHIP_OK;
}
int skyhip_dev_free(skyhip_ctx* c, void* p) {
    if (!p) return SKYHIP_OK;
    if (sky_guard_on()) return sky_guard_free(p) == hipSuccess ? SKYHIP_OK : SKYHIP_E_INVAL;
    if (!c) return hipFree(p) == hipSuccess ? SKYHIP_OK : SKYHIP_E_HIP;      // (the context that allocated it is gone: device memory belongs to the process, see skyhip.h)
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, hipFree(p));
    return SKYHIP_OK;
}

int skyhip_decompress_to_device(skyhip_ctx* c, int n, const uint8_t* const* in, const size_t* in_len, void* const* dst, const size_t* out_cap, size_t* out_len,
                                int32_t* status) {
    if (!c || n < 0) return SKYHIP_E_INVAL;
    if (n == 0) return SKYHIP_OK;
    if (!in || !in_len || !dst || !out_cap || !out_len) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<uint64_t> ioff(n), ilen(n), ooff(n), ocap(n), olen(n);
    uint64_t itot = 0;
    for (int i = 0; i < n; i++) {
        if ((in_len[i] && !in[i]) || (out_cap[i] && !dst[i])) return SKYHIP_E_INVAL;
        ioff[i] = itot; ilen[i] = in_len[i]; itot += (in_len[i] + 255) & ~(uint64_t)255;
        ooff[i] = (uint64_t)(uintptr_t)dst[i]; ocap[i] = out_cap[i];      // the decoder adds these to a null base: absolute device addresses
    }
    HIPCHK(c, c->dec.d_stage_in.ensure(itot + 256));
    for (int i = 0; i < n; i++)
        if (in_len[i]) HIPCHK(c, hipMemcpyAsync(c->dec.d_stage_in.p + ioff[i], in[i], in_len[i], hipMemcpyHostToDevice, c->s_lz4));
    const int rc = sky_lz4d_run(&c->dec, c->s_lz4, n, c->dec.d_stage_in.p, ioff.data(), ilen.data(), nullptr, ooff.data(), ocap.data(), olen.data(), status,
                                &c->dec_ms, c->hip_err, sizeof c->hip_err);
    HIPCHK(c, hipStreamSynchronize(c->s_lz4));
    for (int i = 0; i < n; i++) out_len[i] = (size_t)olen[i];
    return rc;
}

int skyhip_gather_md5(skyhip_ctx* c, int n, const uint64_t* run_prefix, const uint64_t* run_src, const uint32_t* run_len, uint8_t* const* out,
                      const size_t* out_cap, size_t* out_len, uint8_t (*md5)[16]) {
#ifndef SKY_WITH_CDC
    (void)c; (void)n; (void)run_prefix; (void)run_src; (void)run_len; (void)out; (void)out_cap; (void)out_len; (void)md5;
    return SKYHIP_E_INVAL;
#else
    if (!c || n < 0) return SKYHIP_E_INVAL;
    if (n == 0) return SKYHIP_OK;
    if (!run_prefix || !out || !out_cap || !out_len || run_prefix[0] != 0) return SKYHIP_E_INVAL;
    const uint64_t nruns = run_prefix[n];
    if (nruns && (!run_src || !run_len)) return SKYHIP_E_INVAL;
    if (nruns > 0x7FFFFFFFull) return SKYHIP_E_TOOBIG;
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<uint64_t> coff(n), clen(n), dst(nruns ? nruns : 1);
    uint64_t tot = 0;
    for (int i = 0; i < n; i++) {
        if (run_prefix[i + 1] < run_prefix[i]) return SKYHIP_E_INVAL;
        uint64_t len = 0;
        for (uint64_t r = run_prefix[i]; r < run_prefix[i + 1]; r++) len += run_len[r];
        if (len > c->max_chunk) return SKYHIP_E_TOOBIG;
        if (len > out_cap[i] || (len && !out[i])) return SKYHIP_E_CAP;
        coff[i] = tot; clen[i] = len; tot += (len + 255) & ~(uint64_t)255;
    }
    HIPCHK(c, c->dec.d_stage_out.ensure(tot + 256));
    for (int i = 0; i < n; i++) {
        uint64_t at = (uint64_t)(uintptr_t)(c->dec.d_stage_out.p + coff[i]);
        for (uint64_t r = run_prefix[i]; r < run_prefix[i + 1]; r++) { dst[r] = at; at += run_len[r]; }
    }
    if (nruns) {
        HIPCHK(c, c->d_run_src.ensure(nruns)); HIPCHK(c, c->d_run_dst.ensure(nruns)); HIPCHK(c, c->d_run_len.ensure(nruns));
        HIPCHK(c, hipMemcpyAsync(c->d_run_src.p, run_src, nruns * 8, hipMemcpyHostToDevice, c->s_lz4));
        HIPCHK(c, hipMemcpyAsync(c->d_run_dst.p, dst.data(), nruns * 8, hipMemcpyHostToDevice, c->s_lz4));
        HIPCHK(c, hipMemcpyAsync(c->d_run_len.p, run_len, nruns * 4, hipMemcpyHostToDevice, c->s_lz4));
        SkyRunArgs ra; ra.src = c->d_run_src.p; ra.dst = c->d_run_dst.p; ra.len = c->d_run_len.p; ra.n_runs = (uint32_t)nruns;
        uint32_t wgs = (uint32_t)((nruns + 3) / 4);
        if (wgs > (uint32_t)c->lz4s_grid * 8u) wgs = (uint32_t)c->lz4s_grid * 8u;
        hipLaunchKernelGGL(sky_gather_runs, dim3(wgs), dim3(256), 0, c->s_lz4, ra);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->s_lz4));      // (dst lives on this function's stack; and the chunks are whole before anybody reads them)
    for (int i = 0; i < n; i++) {
        out_len[i] = (size_t)clen[i];
        if (clen[i]) HIPCHK(c, hipMemcpyAsync(out[i], c->dec.d_stage_out.p + coff[i], clen[i], hipMemcpyDeviceToHost, c->s_down));      // beside the digest chains
    }
    int rc = SKYHIP_OK;
    if (md5) rc = sky_process_impl(c, n, c->dec.d_stage_out.p, coff.data(), clen.data(), nullptr, nullptr, nullptr, nullptr, md5, SKYHIP_F_MD5, nullptr);
    HIPCHK(c, hipStreamSynchronize(c->s_down));
    return rc;
#endif
}

int skyhip_segment_md5_device(skyhip_ctx* c, size_t nseg, const uint64_t* dev_addr, const uint32_t* len, uint8_t (*fps)[16]) {
#ifndef SKY_WITH_CDC
    (void)c; (void)nseg; (void)dev_addr; (void)len; (void)fps;
    return SKYHIP_E_INVAL;
#else
    if (!c) return SKYHIP_E_INVAL;
    if (nseg == 0) return SKYHIP_OK;
    if (!dev_addr || !len || !fps) return SKYHIP_E_INVAL;
    if (nseg > 0x7FFFFFFFull) return SKYHIP_E_TOOBIG;
    HIPCHK(c, hipSetDevice(c->dev));
    // the digest kernel of the CDC path (sky_segment_md5: a persistent grid deals a segment list to its lanes) over a list the CALLER made: a descriptor is
    // (byte offset from `in`) << 15 | length, and with a null `in` the offset is the device address itself
    std::vector<uint64_t> desc(nseg);
    uint64_t bytes = 0;
    for (size_t i = 0; i < nseg; i++) {
        if (len[i] >= (1u << 15) || dev_addr[i] >= (1ull << 49) || (len[i] && !dev_addr[i])) return SKYHIP_E_INVAL;
        desc[i] = (dev_addr[i] << 15) | len[i];
        bytes += len[i];
    }
    HIPCHK(c, c->d_sv_desc.ensure(nseg)); HIPCHK(c, c->d_sv_fps.ensure(nseg * 16)); HIPCHK(c, c->d_sv_total.ensure(4));
    const uint32_t total = (uint32_t)nseg;
    HIPCHK(c, hipMemcpyAsync(c->d_sv_desc.p, desc.data(), nseg * 8, hipMemcpyHostToDevice, c->s_lz4));
    HIPCHK(c, hipMemcpyAsync(c->d_sv_total.p, &total, 4, hipMemcpyHostToDevice, c->s_lz4));
    SkySegMd5Args ma;
    ma.in = nullptr; ma.desc = c->d_sv_desc.p; ma.seg_total = c->d_sv_total.p; ma.max_segs = total; ma.fps = c->d_sv_fps.p;
    uint64_t end = 0;
    for (size_t i = 0; i < nseg; i++) if (len[i] && dev_addr[i] + len[i] > end) end = dev_addr[i] + len[i];
    ma.in_end = (const uint8_t*)(uintptr_t)end;                  // nothing past the last byte the caller named is read
    uint64_t waves = (nseg + 255u) / 256u;                       // (a range of at least 256 segments per wavefront, as in sky_cdc_run)
    if (waves > c->cdc.segmd5_grid) waves = c->cdc.segmd5_grid;
    if (waves < 1) waves = 1;
    if (c->cdc.segmd5_staged < 0) { const char* e = getenv("SKYHIP_SEGMD5_STAGED"); c->cdc.segmd5_staged = e ? (atoi(e) != 0) : SKY_SEGMD5_STAGED_DEFAULT; }
    if (c->cdc.segmd5_staged) hipLaunchKernelGGL(sky_segment_md5x, dim3((unsigned)waves), dim3(64), 0, c->s_lz4, ma);
    else hipLaunchKernelGGL(sky_segment_md5, dim3((unsigned)waves), dim3(64), 0, c->s_lz4, ma);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(fps, c->d_sv_fps.p, nseg * 16, hipMemcpyDeviceToHost, c->s_lz4));
    HIPCHK(c, hipStreamSynchronize(c->s_lz4));                   // (desc and total live on this function's stack)
    (void)bytes;
    return SKYHIP_OK;
#endif
}

double skyhip_decompress_ms(skyhip_ctx* c, int reset) {
    if (!c) return 0.0;
    const double v = c->dec_ms;
    if (reset) c->dec_ms = 0;
    return v;
}

int skyhip_cdc_results(skyhip_ctx* c, int n, uint64_t* cut_prefix, uint32_t* cuts, size_t cuts_cap, uint8_t* fps, uint64_t* first_seen,
                       uint64_t* seg_base_index) {
    if (!c) return SKYHIP_E_INVAL;
#ifdef SKY_WITH_CDC
    return sky_cdc_results(&c->cdc, n, cut_prefix, cuts, cuts_cap, fps, first_seen, seg_base_index);
#else
    (void)n; (void)cut_prefix; (void)cuts; (void)cuts_cap; (void)fps; (void)first_seen; (void)seg_base_index;
    return SKYHIP_E_INVAL;
#endif
}

int skyhip_dedup_reset(skyhip_ctx* c) {
    if (!c) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
#ifdef SKY_WITH_CDC
    return sky_dedup_reset(&c->cdc, c->s_cdc);
#else
    return SKYHIP_E_INVAL;
#endif
}

// SKY_PROF builds: read (and clear) the accumulated s_memtime phase counters; zeros in the shipping build.
int skyhip_debug_prof(skyhip_ctx* c, uint64_t out[16]) {
    if (!c || !out) return SKYHIP_E_INVAL;
    memset(out, 0, 16 * 8);
    if (c->d_prof) {
        uint64_t all[64 * 16];      // 64 copies (see the kernels): summed here
        HIPCHK(c, hipMemcpy(all, c->d_prof, sizeof all, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemset(c->d_prof, 0, sizeof all));
        // copy k was fed by the waves with (workgroup * 16 + wave) % 64 == k, i.e. by wave k % 16 of its workgroup: SKYHIP_PROF_WAVE=w keeps that wave's
        // counters only (which waves does a barrier wait for?)
        const char* e = getenv("SKYHIP_PROF_WAVE");
        const int only = e ? atoi(e) : -1;
        for (int k = 0; k < 64; k++) if (only < 0 || (k & 15) == only) for (int i = 0; i < 16; i++) out[i] += all[16 * k + i];
    }
    return SKYHIP_OK;
}

// test hook: the (n+1)-th checked HIP call of this context fails with hipErrorUnknown (n < 0 switches it off).  Exercises the
// early-return paths of the calls above, which must leave the context usable (tests/test_gpu_parity.py).
int skyhip_debug_fault(skyhip_ctx* c, long n) {
    if (!c) return SKYHIP_E_INVAL;
    c->fault_after = n;
    return SKYHIP_OK;
}

// guard-band buffers for the caller-owned side of the device-resident calls (tests/test_gpu_guard.py); a device must be current (after skyhip_create)
int skyhip_debug_guard_alloc(size_t bytes, int at_end, void** out) {
    if (!out) return SKYHIP_E_INVAL;
    const hipError_t e = sky_guard_malloc(out, bytes, at_end != 0, 1);
    return e == hipSuccess ? SKYHIP_OK : (e == hipErrorOutOfMemory ? SKYHIP_E_NOMEM : SKYHIP_E_HIP);
}
int skyhip_debug_guard_free(void* p) { return sky_guard_free(p) == hipSuccess ? SKYHIP_OK : SKYHIP_E_INVAL; }
// reads ONE byte at p on the device and returns it (>= 0): the harness's own proof that a byte past a guarded buffer kills the process
int skyhip_debug_guard_probe(skyhip_ctx* c, const void* p) {
    if (!c || !p) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    if (!c->d_self) HIPCHK(c, hipMalloc((void**)&c->d_self, 64));
    hipLaunchKernelGGL(sky_probe_kernel, dim3(1), dim3(64), 0, c->s_lz4, (const uint8_t*)p, c->d_self);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->s_lz4));
    uint32_t v = 0;
    HIPCHK(c, hipMemcpy(&v, c->d_self, 4, hipMemcpyDeviceToHost));
    return (int)(v & 0xFFu);
}

int skyhip_selftest(skyhip_ctx* c) {
    if (!c) return SKYHIP_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    uint8_t* d_buf = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_buf, 8192));
    uint8_t h[4096];
    for (int i = 0; i < 4096; i++) h[i] = (uint8_t)(i * 37 + (i >> 3));
    HIPCHK(c, hipMemcpy(d_buf, h, 4096, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(d_buf + 4096, 0, 4096));
    uint32_t* d_res = (uint32_t*)(d_buf + 4096 + 2048);
    hipLaunchKernelGGL(sky_selftest_kernel, dim3(1), dim3(64), 0, c->s_lz4, d_buf, d_buf + 4096, d_res);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->s_lz4));
    uint32_t res[8] = {0};
    HIPCHK(c, hipMemcpy(res, d_res, sizeof(uint32_t) * 5, hipMemcpyDeviceToHost));
    (void)hipFree(d_buf);
    for (int k = 0; k < 5; k++) if (res[k]) return k + 1;
    return 0;
}

}  // extern "C"
