// Development probe (not part of the product): dependent-chain latency of the VALU ops the MD5 round uses, one wave per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 4096
template <int OP> __global__ void chain(uint32_t* out, uint64_t* cyc, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t x = a + threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < REP; i++) {
        if (OP == 0) x = x + b;                                                // v_add_u32
        else if (OP == 1) x = __builtin_amdgcn_alignbit(x, x, 25);             // v_alignbit_b32 (rotate)
        else if (OP == 2) x = x + b + c;                                        // v_add3_u32
        else if (OP == 3) x = (x & b) | (~x & c);                               // v_bfi_b32
        else if (OP == 4) x = c ^ (x | ~b);                                     // v_bitop3
        else if (OP == 5) x = (x << 7) | (x >> 25);                             // hipcc's rotate
        else if (OP == 6) { uint32_t f = (x & b) | (~x & c); uint32_t t = a + f + 0x12345678u; t = __builtin_amdgcn_alignbit(t, t, 25); x = t + x; }  // one MD5-like step (4 dependent ops)
        else if (OP == 7) { uint32_t f = (x & b) | (~x & c); uint32_t t = a + f + 0x12345678u; uint32_t hi = t >> 25; t = (t << 7) | hi; x = t + x; }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char* name, uint32_t* d_out, uint64_t* d_cyc, int ops) {
    hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 3u, 0x9e3779b9u, 0x85ebca6bu);
    hipDeviceSynchronize();
    uint64_t c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.2f ticks per iteration (%d dependent op(s))\n", name, (double)c / REP, ops);
}
int main() {
    uint32_t* d_out; uint64_t* d_cyc; hipMalloc(&d_out, 256); hipMalloc(&d_cyc, 8);
    for (int r = 0; r < 2; r++) {
        run<0>("v_add_u32", d_out, d_cyc, 1); run<1>("v_alignbit_b32", d_out, d_cyc, 1); run<2>("v_add3_u32", d_out, d_cyc, 1);
        run<3>("v_bfi_b32", d_out, d_cyc, 1); run<4>("v_bitop3", d_out, d_cyc, 1); run<5>("rotate (hipcc)", d_out, d_cyc, 1);
        run<6>("md5-like step (alignbit)", d_out, d_cyc, 4); run<7>("md5-like step (shifts)", d_out, d_cyc, 4);
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clock %d kHz; readcyclecounter ticks may be a fixed 100 MHz clock\n", clk);
    return 0;
}
