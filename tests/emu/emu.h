// emu.h -- fiber-based SIMT emulator: TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Runs the shipping kernel source (skyplane_amd/csrc/*.inc, written against wave.h) on the CPU:
// every lane of a workgroup is a coroutine with its own stack; a collective parks the lane until all
// live lanes of its wave (or workgroup, for the barrier) have arrived, then the scheduler resolves it.
// Semantics mirrored from gfx950: 64-lane waves, ballot over live lanes only, ds_bpermute reads 0 from
// exited lanes.  Collectives must be called in wave-uniform control flow (asserted).
#pragma once
#include <stdint.h>
#include <string.h>

#define SKY_DEV static inline
#define SKY_WAVE 64
typedef unsigned long long sky_u64;

enum EmuOp { EMU_NONE = 0, EMU_BALLOT, EMU_READLANE, EMU_SHFL, EMU_SCAN, EMU_BARRIER, EMU_EXIT, EMU_WAVESYNC, EMU_SCANMAX, EMU_YIELD, EMU_PUSH };

struct EmuLaneState {
    void* sp;            // saved stack pointer of the parked coroutine
    void* stack;
    int tid;
    int op;              // collective the lane is parked at
    sky_u64 arg0;        // value
    sky_u64 arg1;        // lane index / predicate
    sky_u64 result;
    bool done;
};

struct EmuBlockCtx {
    int bid, bdim, gdim;
    EmuLaneState* lanes;
    void* sched_sp;
};

extern thread_local EmuBlockCtx* emu_blk;
extern thread_local EmuLaneState* emu_cur;
sky_u64 emu_collective(int op, sky_u64 a0, sky_u64 a1);

SKY_DEV int sky_tid() { return emu_cur->tid; }
SKY_DEV int sky_bid() { return emu_blk->bid; }
SKY_DEV int sky_bdim() { return emu_blk->bdim; }
SKY_DEV int sky_gdim() { return emu_blk->gdim; }
SKY_DEV int sky_lane() { return emu_cur->tid & 63; }
SKY_DEV int sky_wave_in_block() { return emu_cur->tid >> 6; }

SKY_DEV sky_u64 sky_ballot(bool p) { return emu_collective(EMU_BALLOT, p ? 1 : 0, 0); }
SKY_DEV uint32_t sky_readlane(uint32_t v, int lane) { return (uint32_t)emu_collective(EMU_READLANE, v, (sky_u64)lane); }
SKY_DEV uint32_t sky_readfirstlane(uint32_t v) { return (uint32_t)emu_collective(EMU_READLANE, v, (sky_u64)-1); }
SKY_DEV uint32_t sky_writelane(uint32_t old, uint32_t val, int lane) { return sky_lane() == lane ? val : old; }
SKY_DEV uint32_t sky_shfl(uint32_t v, int src) { return (uint32_t)emu_collective(EMU_SHFL, v, (sky_u64)(src & 63)); }
SKY_DEV uint32_t sky_push(uint32_t to, uint32_t v) { return (uint32_t)emu_collective(EMU_PUSH, v, (sky_u64)(to & 63u)); }
SKY_DEV sky_u64 sky_bitset64(sky_u64 m, uint32_t i) { return m | (1ull << (i & 63u)); }
SKY_DEV uint32_t sky_mbcnt64(sky_u64 m) { return (uint32_t)__builtin_popcountll(m & ((1ull << sky_lane()) - 1ull)); }
SKY_DEV int sky_clz64(sky_u64 x) { return __builtin_clzll(x); }
SKY_DEV uint32_t sky_scan_incl_add(uint32_t v) { return (uint32_t)emu_collective(EMU_SCAN, v, 0); }
SKY_DEV uint32_t sky_scan_incl_max(uint32_t v) { return (uint32_t)emu_collective(EMU_SCANMAX, v, 0); }
SKY_DEV uint32_t sky_wave_max_u32(uint32_t v) { return (uint32_t)emu_collective(EMU_READLANE, emu_collective(EMU_SCANMAX, v, 0), 63); }
SKY_DEV uint32_t sky_wave_shr1(uint32_t v) { const uint32_t t = (uint32_t)emu_collective(EMU_SHFL, v, (sky_u64)((sky_lane() - 1) & 63)); return sky_lane() ? t : 0u; }
SKY_DEV uint32_t sky_scan_incl_add_shfl(uint32_t v) { return (uint32_t)emu_collective(EMU_SCAN, v, 0); }
SKY_DEV void sky_sched_fence() {}
template <int P> SKY_DEV void sky_setprio() {}
SKY_DEV void sky_syncthreads() { emu_collective(EMU_BARRIER, 0, 0); }
SKY_DEV void sky_syncthreads_lds() { emu_collective(EMU_BARRIER, 0, 0); }
// lanes of the emulator run one after the other between collectives; on hardware they run in lock-step, and the
// kernels put a wave fence wherever a lane reads what another lane of the same wave just wrote: make it a rendezvous
SKY_DEV void sky_wave_fence() { emu_collective(EMU_WAVESYNC, 0, 0); }

// lanes never run concurrently, so plain read-modify-write is atomic here
SKY_DEV uint32_t sky_atomic_add_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
SKY_DEV void sky_lds_min_u32(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
SKY_DEV uint32_t sky_lds_add_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
SKY_DEV void sky_lds_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
SKY_DEV uint32_t sky_lds_poll_u32(const uint32_t* p) { return *(const volatile uint32_t*)p; }
SKY_DEV sky_u64 sky_lds_poll_u64(const sky_u64* p) { return *(const volatile sky_u64*)p; }
SKY_DEV void sky_lds_order() { emu_collective(EMU_WAVESYNC, 0, 0); }      // (see sky_wave_fence)
// a wavefront that waits for another one: the scheduler parks it and runs the other wavefronts of the workgroup first
SKY_DEV void sky_wave_yield() { emu_collective(EMU_YIELD, 0, 0); }
SKY_DEV sky_u64 sky_atomic_min_u64(sky_u64* p, sky_u64 v) { sky_u64 o = *p; if (v < o) *p = v; return o; }
SKY_DEV sky_u64 sky_atomic_cas_u64(sky_u64* p, sky_u64 e, sky_u64 d) { sky_u64 o = *p; if (o == e) *p = d; return o; }
SKY_DEV sky_u64 sky_atomic_load_u64(const sky_u64* p) { return *p; }

SKY_DEV int sky_ctz64(sky_u64 x) { return __builtin_ctzll(x); }
SKY_DEV int sky_popc64(sky_u64 x) { return __builtin_popcountll(x); }
SKY_DEV uint32_t sky_mul24(uint32_t a, uint32_t b) { return (uint32_t)((sky_u64)(a & 0xFFFFFFu) * (sky_u64)(b & 0xFFFFFFu)); }
SKY_DEV uint32_t sky_mad24(uint32_t a, uint32_t b, uint32_t c) { return sky_mul24(a, b) + c; }
SKY_DEV uint32_t sky_byte_x8(uint32_t w, int k) { return ((w >> (8 * k)) & 0xFFu) << 3; }
SKY_DEV uint32_t sky_perm(uint32_t hi, uint32_t lo, uint32_t sel) {      // v_perm_b32 (selectors 0-7 and 0x0c only: what the kernels use)
    const sky_u64 both = ((sky_u64)hi << 32) | lo;
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t sl = (sel >> (8 * k)) & 0xFFu;
        const uint32_t b = sl < 8u ? (uint32_t)(both >> (8 * sl)) & 0xFFu : (sl == 0x0cu ? 0u : 0xFFu);
        r |= b << (8 * k);
    }
    return r;
}
SKY_DEV uint32_t sky_shl1_lt(uint32_t bits, uint32_t a, uint32_t b) { return bits + bits + (a < b ? 1u : 0u); }
SKY_DEV uint32_t sky_shl1_eq(uint32_t bits, uint32_t a, uint32_t b) { return bits + bits + (a == b ? 1u : 0u); }
SKY_DEV void sky_keep(uint32_t) {}
SKY_DEV uint32_t sky_undef32() { return 0xDEADBEEFu; }
SKY_DEV uint32_t sky_opaque(uint32_t v) { return v; }
#define SKY_RESTRICT
SKY_DEV sky_u64 sky_clock() { return 0; }
SKY_DEV uint32_t sky_uniform(uint32_t v) { return v; }
SKY_DEV sky_u64 sky_uniform64(sky_u64 v) { return v; }   // uniform by contract: nothing to do
SKY_DEV bool sky_lanebit(sky_u64 uniform_mask) { return (uniform_mask >> (emu_cur->tid & 63)) & 1ull; }

// Launch: run `body(args)` for every thread of every workgroup, one workgroup at a time.
// `smem` is the workgroup's LDS (zero-filled is NOT guaranteed on hardware; the emulator poisons it).
typedef void (*EmuKernelBody)(void* args, uint8_t* smem);
void emu_launch(int grid, int block, size_t lds_bytes, EmuKernelBody body, void* args);
