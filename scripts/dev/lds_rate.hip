// Development probe (not part of the product): what does one gfx950 CU's LDS sustain for the access patterns of the LZ4 compressor?
// One 1024-lane workgroup (16 waves, like sky_lz4s_compress), every lane at its own pseudo-random address inside 128 KiB, eight independent
// operations per loop iteration.  Reported: CU cycles per wave-level instruction (wall clock x reported clock / instructions of all 16 waves),
// with the address-generation-only loop measured beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP (1 << 12)
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
enum { NONE, R32, R64A, R64U, R96A, R128A, R128U, MIN32, W8, W32, W64U, R32_SAME_ROW };
template <int OP> __global__ void __launch_bounds__(1024) probe(uint32_t* out, uint32_t seed, int lanes_active) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    for (uint32_t i = threadIdx.x; i < 32768u; i += 1024u) ((uint32_t*)sm)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t st = threadIdx.x * 747796405u + seed, acc = 0;
    if ((int)(threadIdx.x & 63u) < lanes_active) {
        for (int it = 0; it < REP; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                st = st * 1664525u + 1013904223u;
                const uint32_t a = (st >> 12) & 0x1FFF0u;            // 16-byte aligned offset inside 128 KiB
                const uint32_t odd = (st >> 8) & 15u;                 // byte skew for the unaligned forms
                if (OP == NONE) acc ^= a + odd;
                if (OP == R32) acc ^= *(const uint32_t*)(sm + a + (odd & 12u));
                if (OP == R32_SAME_ROW) acc ^= *(const uint32_t*)(sm + a);
                if (OP == R64A) { const v2u v = *(const v2u*)(sm + a + (odd & 8u)); acc ^= v.x ^ v.y; }
                if (OP == R64U) { v2u v; __builtin_memcpy(&v, sm + a + odd, 8); acc ^= v.x ^ v.y; }
                if (OP == R96A) { v3u v; __builtin_memcpy(&v, __builtin_assume_aligned(sm + a, 16), 12); acc ^= v.x ^ v.y ^ v.z; }
                if (OP == R128A) { const v4u v = *(const v4u*)(sm + a); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
                if (OP == R128U) { v4u v; __builtin_memcpy(&v, sm + a + odd, 16); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
                if (OP == MIN32) __hip_atomic_fetch_min((uint32_t*)__builtin_assume_aligned(sm + a + (odd & 12u), 4), st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OP == W8) sm[a + odd] = (uint8_t)st;
                if (OP == W32) *(uint32_t*)(sm + a + (odd & 12u)) = st;
                if (OP == W64U) { const v2u v = {st, st}; __builtin_memcpy(sm + a + odd, &v, 8); }
            }
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + sm[threadIdx.x];
}
template <typename F> double timed(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int OP> void run(const char* name, uint32_t* d_out, double ghz, double base[3]) {
    const int act[3] = {64, 32, 16};
    printf("%-34s", name);
    for (int k = 0; k < 3; k++) {
        hipFuncSetAttribute((const void*)probe<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 64);
        const double ms = timed([&] { hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(1024), 131072 + 64, 0, d_out, 12345u, act[k]); });
        const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)REP * 8 * 16);
        if (OP == NONE) base[k] = cyc;
        printf("  %2d lanes: %6.2f (-addr %6.2f)", act[k], cyc, cyc - base[k]);
    }
    printf("\n");
}
int main() {
    uint32_t* d_out; hipMalloc(&d_out, 1 << 20);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk / 1e6;
    double base[3] = {0, 0, 0};
    printf("reported clock %.2f GHz; CU cycles per wave-level LDS instruction, 16 waves, random addresses, by active lanes per wave\n", ghz);
    run<NONE>("address arithmetic only", d_out, ghz, base);
    run<R32>("ds_read_b32", d_out, ghz, base);
    run<R32_SAME_ROW>("ds_read_b32 (16-byte aligned)", d_out, ghz, base);
    run<R64A>("ds_read_b64 aligned", d_out, ghz, base);
    run<R64U>("ds_read_b64 unaligned", d_out, ghz, base);
    run<R96A>("ds_read_b96 (16-byte aligned)", d_out, ghz, base);
    run<R128A>("ds_read_b128 aligned", d_out, ghz, base);
    run<R128U>("ds_read_b128 unaligned", d_out, ghz, base);
    run<MIN32>("ds_min_u32", d_out, ghz, base);
    run<W8>("ds_write_b8", d_out, ghz, base);
    run<W32>("ds_write_b32", d_out, ghz, base);
    run<W64U>("ds_write_b64 unaligned", d_out, ghz, base);
    return 0;
}
