// Development probe (not part of the product): what does the chip sustain in MD5 blocks per second when nothing but the round instructions runs?
// W wavefronts per SIMD, every lane hashing blocks out of registers (no memory).  Reported: wave-level VALU instructions per cycle and SIMD
// (an MD5 block is 64 steps x ~5.2 instructions as compiled) and the GB/s of message bytes that corresponds to.
//   hipcc --offload-arch=gfx950 -O3 -I skyplane_amd/csrc -I include -o scripts/dev/md5_rate scripts/dev/md5_rate.hip && ./scripts/dev/md5_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "md5_kernel.inc"
#define REP 4096
__global__ void __launch_bounds__(64) probe(uint32_t* out) {
    uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint32_t M[16];
    for (int i = 0; i < 16; i++) M[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int r = 0; r < REP; r++) {
        sky_md5_block(st, M);
        M[r & 15] ^= st[r & 3];
    }
    if ((st[0] ^ st[1] ^ st[2] ^ st[3]) == 0x12345u) out[0] = st[0];
}
int main() {
    uint32_t* d_out; hipMalloc(&d_out, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int w : {1, 2, 4, 6, 8}) {
        const int grid = cus * 4 * w;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d_out); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d_out); hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double blocks = (double)grid * 64.0 * REP;
        printf("  %d wave(s) per SIMD: %8.3f ms  %7.1f GB/s of message bytes  (%.2f cycles per block and SIMD-wave at %.2f GHz)\n", w, ms, blocks * 64.0 / ms / 1e6,
               ms * 1e-3 * p.clockRate * 1e3 / ((double)REP * w), p.clockRate / 1e6);
    }
    return 0;
}
