// Development probe (not part of the product): how fast does one SIMD of a gfx950 CU issue integer VALU / SALU work?
//   * one wave, K independent dependency chains (K = 1, 2, 4): does a lone wave issue more than one instruction per ~4.5 cycles?
//   * 1, 2, 4 waves per SIMD, each running one dependent chain: what does the SIMD sustain in total?
//   * VALU waves beside SALU waves: do the two pipes overlap?
// Times are wall clock (hipEvent) over many iterations, converted with the reported clock; ratios are what matters.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP (1 << 16)
template <int K> __global__ void valu_chains(uint32_t* out, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < REP; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(b));
            if (K > 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(b));
            if (K > 2) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(b)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
// waves with (wave index & 1) == 1 run a scalar chain instead of a vector one when MIX != 0
template <int MIX> __global__ void mix(uint32_t* out, uint32_t b) {
    uint32_t x = threadIdx.x;
    uint32_t s = __builtin_amdgcn_readfirstlane(b);
    const bool scalar = MIX && ((threadIdx.x >> 6) & 1);
    if (!scalar) {
        for (int i = 0; i < REP; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        }
    } else {
        for (int i = 0; i < REP; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s) : : "scc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + s;
}
template <typename F> double timed(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    uint32_t* d_out; hipMalloc(&d_out, 1 << 20);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk / 1e6;
    const double n = (double)REP * 16;
    printf("reported clock %.2f GHz\n", ghz);
    for (int threads : {64, 256, 512, 1024}) {
        const int wps = threads >= 256 ? threads / 256 : 1;
        double m1 = timed([&] { hipLaunchKernelGGL(valu_chains<1>, dim3(1), dim3(threads), 0, 0, d_out, 3u); });
        double m2 = timed([&] { hipLaunchKernelGGL(valu_chains<2>, dim3(1), dim3(threads), 0, 0, d_out, 3u); });
        double m4 = timed([&] { hipLaunchKernelGGL(valu_chains<4>, dim3(1), dim3(threads), 0, 0, d_out, 3u); });
        printf("%4d threads (%d wave(s) per SIMD): cycles per v_add per wave: 1 chain %.2f | 2 chains %.2f | 4 chains %.2f   -> SIMD issues one VALU per %.2f cycles at best\n",
               threads, wps, m1 * 1e-3 * ghz * 1e9 / n, m2 * 1e-3 * ghz * 1e9 / (2 * n), m4 * 1e-3 * ghz * 1e9 / (4 * n), m4 * 1e-3 * ghz * 1e9 / (4 * n) / wps);
    }
    for (int threads : {512, 1024}) {
        double v = timed([&] { hipLaunchKernelGGL(mix<0>, dim3(1), dim3(threads), 0, 0, d_out, 3u); });
        double x = timed([&] { hipLaunchKernelGGL(mix<1>, dim3(1), dim3(threads), 0, 0, d_out, 3u); });
        printf("%4d threads: all-VALU waves %.3f ms; every second wave SALU instead %.3f ms (equal = the pipes do not overlap, lower = they do)\n", threads, v, x);
    }
    return 0;
}
