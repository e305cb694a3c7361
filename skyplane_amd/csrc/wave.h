// wave.h -- the small set of wavefront primitives the skyhip kernels are written against.
//
// Device build (hipcc, gfx950): each primitive is a CDNA4 builtin -- 64-wide ballots, readlane,
// ds_bpermute shuffles and DPP row-shift scans.  There is no 32-wide path and no CUDA spelling.
//
// SKY_EMU build (g++, tests/emu only): the same names are provided by tests/emu/emu.h, a fiber-based
// SIMT emulator that runs every lane of a workgroup as a coroutine and resolves each collective when
// all live lanes of the wave have arrived.  That lets the CPU test-suite execute the *shipping kernel
// source* on a box with no GPU.  The emulator is test infrastructure: nothing in the product links it.
//
// Contract for kernel authors: every collective (ballot / readlane / shfl / scan / barrier) is called
// in wave-uniform control flow.
#pragma once
#include <stdint.h>

#ifdef SKY_EMU
#include "emu.h"
#else
#include <hip/hip_runtime.h>

#define SKY_DEV __device__ __forceinline__
#define SKY_WAVE 64

typedef unsigned long long sky_u64;

SKY_DEV int sky_tid() { return (int)threadIdx.x; }
SKY_DEV int sky_bid() { return (int)blockIdx.x; }
SKY_DEV int sky_bdim() { return (int)blockDim.x; }
SKY_DEV int sky_gdim() { return (int)gridDim.x; }
SKY_DEV int sky_lane() { return (int)(threadIdx.x & 63u); }
SKY_DEV int sky_wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

SKY_DEV sky_u64 sky_ballot(bool p) { return __ballot(p); }
SKY_DEV uint32_t sky_readlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
SKY_DEV uint32_t sky_readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// v_writelane_b32: lane `lane` (uniform) of the result holds the uniform value `val`, every other lane keeps `old`
// (no clang builtin in this toolchain; the lane select goes through M0, which does not count against gfx9's
// one-SGPR-per-VALU constant bus limit)
SKY_DEV uint32_t sky_writelane(uint32_t old, uint32_t val, int lane) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"
                 : "+v"(old)
                 : "s"(__builtin_amdgcn_readfirstlane((int)val)), "s"(__builtin_amdgcn_readfirstlane(lane)));   // M0 is reserved: the compiler keeps nothing live in it
    return old;
}
// arbitrary gather across lanes (ds_bpermute_b32)
SKY_DEV uint32_t sky_shfl(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }
SKY_DEV void sky_syncthreads() { __syncthreads(); }
// a barrier that protects LDS contents only: it does not wait for this wavefront's global stores (s_waitcnt vmcnt(0) is part of __syncthreads)
SKY_DEV void sky_syncthreads_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// compiler-only: nothing is scheduled across this point (keeps unrolled load groups from being merged and spilled)
SKY_DEV void sky_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// s_setprio: issue priority of this wave among the waves of its SIMD (0 = default ... 3)
template <int P> SKY_DEV void sky_setprio() { __builtin_amdgcn_s_setprio(P); }
// compile-time ordering of this wave's LDS/global accesses (lanes of one wave execute DS ops in issue order)
SKY_DEV void sky_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// DPP inclusive add-scan over the 64 lanes: 4 row_shr steps inside each 16-lane row, then row_bcast:15 into rows
// 1,3 and row_bcast:31 into rows 2,3 (gfx9-family DPP controls).  Written as six in-place v_add_u32_dpp (hipcc
// lowers the builtin form to mov+mov_dpp+add per step); "s_nop 1" covers the VALU-write -> DPP-read hazard.
SKY_DEV uint32_t sky_scan_incl_add(uint32_t x) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x));
    return x;
}
// the same scan with max instead of add (unsigned: the zero that bound_ctrl shifts in is the identity)
SKY_DEV uint32_t sky_scan_incl_max(uint32_t x) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x));
    return x;
}
SKY_DEV uint32_t sky_wave_max_u32(uint32_t x) { return sky_readlane(sky_scan_incl_max(x), 63); }
// the value of the lane before (0 in lane 0): one DPP move across the whole wavefront (wave_shr:1), no LDS round trip
SKY_DEV uint32_t sky_wave_shr1(uint32_t x) {
    uint32_t r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0" : "=&v"(r) : "v"(x));
    return r;
}
SKY_DEV uint32_t sky_scan_incl_add_shfl(uint32_t x) {
    const int lane = sky_lane();
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = sky_shfl(x, (lane - d) & 63);
        if (lane >= d) x += t;
    }
    return x;
}

SKY_DEV uint32_t sky_atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
// LDS atomics (workgroup scope): ds_min_u32 without return, ds_add_rtn_u32
SKY_DEV void sky_lds_min_u32(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SKY_DEV uint32_t sky_lds_add_u32(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SKY_DEV void sky_lds_or_u32(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }      // ds_or_b32, no return
// flags in LDS that one wavefront of a workgroup stores and the others poll (sky_lz4_link): a real ds_read per poll; sky_wave_yield gives the SIMD to the
// wavefronts that are being waited for
SKY_DEV uint32_t sky_lds_poll_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SKY_DEV sky_u64 sky_lds_poll_u64(const sky_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SKY_DEV void sky_wave_yield() { __builtin_amdgcn_s_sleep(1); }
// compiler-only ordering of this wavefront's LDS accesses: the DS queue of a wavefront is served in order, so a load issued after a store sees it and a
// flag stored after data is seen after it -- nothing to wait for (sky_wave_fence costs an s_waitcnt lgkmcnt(0))
SKY_DEV void sky_lds_order() { asm volatile("" ::: "memory"); }
SKY_DEV sky_u64 sky_atomic_min_u64(sky_u64* p, sky_u64 v) { return atomicMin(p, v); }
SKY_DEV sky_u64 sky_atomic_cas_u64(sky_u64* p, sky_u64 expect, sky_u64 desired) { return atomicCAS(p, expect, desired); }
SKY_DEV sky_u64 sky_atomic_load_u64(const sky_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

SKY_DEV int sky_ctz64(sky_u64 x) { return __builtin_ctzll(x); }
SKY_DEV int sky_popc64(sky_u64 x) { return __builtin_popcountll(x); }
SKY_DEV int sky_clz64(sky_u64 x) { return __builtin_clzll(x); }
// bit (i mod 64) of a wave-uniform mask set: one s_bitset1_b64 (no shift, no or, no mask of the index)
SKY_DEV sky_u64 sky_bitset64(sky_u64 m, uint32_t i) { asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(i)); return m; }
// set bits of a wave-uniform mask below my lane
SKY_DEV uint32_t sky_mbcnt64(sky_u64 m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// forward permute: every lane sends v to lane `to`; a lane nobody sends to receives 0, the highest sender wins a conflict (ds_permute_b32; no LDS memory involved)
SKY_DEV uint32_t sky_push(uint32_t to, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_permute((int)(to << 2), (int)v); }
// full-rate 24-bit integer multiply(-add): v_mul_u32_u24 / v_mad_u32_u24 use the low 24 bits of a and b and return the low 32 bits of the product
// (a 32-bit v_mul_lo_u32 issues at a quarter of that rate)
SKY_DEV uint32_t sky_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
SKY_DEV uint32_t sky_mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
// byte k (0..3, compile-time) of w, times 8 -- the byte offset of entry [byte] in a table of 8-byte entries -- in ONE instruction: the shift reads its
// operand through SDWA's byte select (the compiler emits v_bfe_u32 + v_lshl_add_u32 for the same thing)
SKY_DEV uint32_t sky_byte_x8(uint32_t w, int k) {
    uint32_t r;
    const uint32_t three = 3u;
    if (k == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
    else if (k == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
    else if (k == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
    return r;
}
// v_perm_b32: byte k of the result is byte sel[k] of the 8 bytes {hi, lo} (0-3 = lo's bytes, 4-7 = hi's, 0x0c = 0x00)
SKY_DEV uint32_t sky_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// bits = 2 * bits + (a < b)  /  + (a == b): a compare into VCC and v_addc_co_u32 bits, bits, bits, vcc -- two VALU instructions per position and no
// scalar ones, where a compare + select + or costs three and building the union of two conditions an s_or_b64 on top (the scalar unit is the
// compressor's second-busiest, profiles/r3_pmc_lz4s.txt).  Collecting a lane's bit mask this way fills it from the top: callers walk positions downwards.
SKY_DEV uint32_t sky_shl1_lt(uint32_t bits, uint32_t a, uint32_t b) {
    asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(a), "v"(b) : "vcc");
    return bits;
}
SKY_DEV uint32_t sky_shl1_eq(uint32_t bits, uint32_t a, uint32_t b) {
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(a), "v"(b) : "vcc");
    return bits;
}
// keep a prefetched value alive up to this point (the load warms L2/L1 for a later batch; nothing reads it)
// returns v, but the compiler may assume nothing about the result (stops CSE / hoisting across this point)
SKY_DEV uint32_t sky_opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
SKY_DEV void sky_keep(uint32_t v) { asm volatile("" ::"v"(v)); }
// a register whose content nobody will look at (no instruction: spares the zeroing of values that only some paths define)
SKY_DEV uint32_t sky_undef32() { uint32_t v; asm volatile("" : "=v"(v)); return v; }
#define SKY_RESTRICT __restrict__
// shader clock (s_memtime) for the SKY_PROF phase-timing build only
SKY_DEV sky_u64 sky_clock() { sky_u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
// tell the compiler a value is wave-uniform (keeps it in SGPRs; folds away when it already is)
SKY_DEV uint32_t sky_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
SKY_DEV sky_u64 sky_uniform64(sky_u64 v) {
    return (sky_u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((sky_u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
// bit `lane` of a wave-uniform mask as a per-lane predicate: the SGPR pair IS the predicate (no shifts)
SKY_DEV bool sky_lanebit(sky_u64 uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }
#endif  // SKY_EMU

// index of the lowest set bit, 0xFFFFFFFF for x == 0 (v_ffbl_b32 semantics on both builds)
#ifdef SKY_EMU
SKY_DEV uint32_t sky_ffbl32(uint32_t x) { return (uint32_t)(__builtin_ffs((int)x) - 1); }
#else
// one instruction: the hardware already returns -1 for zero, which C's ctz/ffs cannot express without a select
SKY_DEV uint32_t sky_ffbl32(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#endif
// leading-zero count, 0xFFFFFFFF for x == 0 (v_ffbh_u32 semantics on both builds)
#ifdef SKY_EMU
SKY_DEV uint32_t sky_ffbh32(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 0xFFFFFFFFu; }
#else
SKY_DEV uint32_t sky_ffbh32(uint32_t x) { uint32_t r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#endif

// ---- helpers common to both builds ---------------------------------------------------------------
// Unaligned little-endian loads.  gfx950 under amdhsa runs with unaligned access mode enabled, so the
// compiler turns these memcpys into single global_load_dword / dwordx2 / dwordx4 instructions.
SKY_DEV uint32_t sky_ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
SKY_DEV sky_u64 sky_ld64u(const uint8_t* p) { sky_u64 v; __builtin_memcpy(&v, p, 8); return v; }
SKY_DEV void sky_st16u(uint8_t* p, uint32_t v) { uint16_t s = (uint16_t)v; __builtin_memcpy(p, &s, 2); }
SKY_DEV void sky_st32u(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

struct sky_u128 { uint32_t x, y, z, w; };
SKY_DEV sky_u128 sky_ld128u(const uint8_t* p) { sky_u128 v; __builtin_memcpy(&v, p, 16); return v; }
// the same as ONE 16-byte access whatever the optimizer thinks of the four fields' uses (a memcpy into a struct may be split into two 8-byte
// loads, and on LDS two misaligned loads cost twice what one does: a lane per cycle each)
#ifdef SKY_EMU
SKY_DEV sky_u128 sky_ld128u1(const uint8_t* p) { return sky_ld128u(p); }
#else
typedef uint32_t sky_v4u_b1 __attribute__((ext_vector_type(4), aligned(1)));
SKY_DEV sky_u128 sky_ld128u1(const uint8_t* p) { const sky_v4u_b1 t = *(const sky_v4u_b1*)p; sky_u128 v; v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; return v; }
#endif
SKY_DEV void sky_st128u(uint8_t* p, const sky_u128& v) { __builtin_memcpy(p, &v, 16); }
SKY_DEV void sky_st64u(uint8_t* p, sky_u64 v) { __builtin_memcpy(p, &v, 8); }
// naturally aligned forms (one ds_read_b128 / ds_write_b128 / b64 / b32 on LDS)
#ifdef SKY_EMU
SKY_DEV sky_u128 sky_ld128a(const void* p) { sky_u128 v; __builtin_memcpy(&v, p, 16); return v; }
SKY_DEV void sky_st128a(void* p, const sky_u128& v) { __builtin_memcpy(p, &v, 16); }
#else
typedef uint32_t sky_v4u __attribute__((ext_vector_type(4)));     // a 16-byte aligned type: one b128 / dwordx4 access
SKY_DEV sky_u128 sky_ld128a(const void* p) { const sky_v4u t = *(const sky_v4u*)p; sky_u128 v; v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; return v; }
SKY_DEV void sky_st128a(void* p, const sky_u128& v) { sky_v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(sky_v4u*)p = t; }
#endif
SKY_DEV sky_u64 sky_ld64a(const void* p) { sky_u64 v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 8), 8); return v; }
SKY_DEV void sky_st64a(void* p, sky_u64 v) { __builtin_memcpy(__builtin_assume_aligned(p, 8), &v, 8); }
SKY_DEV uint32_t sky_ld32a(const void* p) { uint32_t v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 4), 4); return v; }
