// Development probe (not part of the product): what does gfx950's vector-memory path sustain for the load patterns of the streaming kernels
// (sky_md5_chunks, sky_segment_md5, sky_gear_candidates), whose lanes each read their OWN byte stream 16 bytes at a time?
//   P0 lane-stream : lane l reads 64 contiguous bytes (4 x 16) of stream l per trip                      -- what the digest kernels do
//   P1 quad-coop   : the four lanes of a quad read the 64 bytes of ONE of their four streams per load     -- 16 sectors of 64 B per instruction
//   P2 octet-coop  : eight lanes read 128 bytes (a cache line) of one of their eight streams per load    -- 8 lines per instruction
//   P3 coalesced   : the wavefront reads 1 KiB contiguous per load                                       -- the ceiling
// Streams are S bytes apart (S = 4608: a CDC segment; S = 512: a Gear run).  Every byte of a 4 GiB buffer is read once.
//   hipcc --offload-arch=gfx950 -O3 -o gload_rate gload_rate.hip && ./gload_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define LD(p) (*(const v4u*)(p))

// a wavefront owns 64 streams of S bytes, back to back: 64 * S bytes; `trips` = S / 64 (P0, P1) or S / 128 (P2)
template <int P> __global__ void __launch_bounds__(64) probe(const uint8_t* in, uint32_t S, uint32_t waves_total, uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t w = blockIdx.x; w < waves_total; w += gridDim.x) {
        const uint8_t* base = in + (size_t)w * 64u * S;
        if (P == 0) {
            const uint8_t* p = base + (size_t)lane * S;
            for (uint32_t t = 0; t < S / 64u; t++) {
                const v4u a = LD(p + 64u * t), b = LD(p + 64u * t + 16), c = LD(p + 64u * t + 32), d = LD(p + 64u * t + 48);
                acc ^= a.x ^ b.y ^ c.z ^ d.w;
            }
        } else if (P == 1) {
            const uint32_t q = lane >> 2, i = lane & 3u;
            for (uint32_t t = 0; t < S / 64u; t++) {
                v4u v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) v[k] = LD(base + (size_t)(4u * q + k) * S + 64u * t + 16u * i);
                acc ^= v[0].x ^ v[1].y ^ v[2].z ^ v[3].w;
            }
        } else if (P == 2) {
            const uint32_t o = lane >> 3, i = lane & 7u;
            for (uint32_t t = 0; t < S / 128u; t++) {
                v4u v[8];
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) v[k] = LD(base + (size_t)(8u * o + k) * S + 128u * t + 16u * i);
                acc ^= v[0].x ^ v[1].y ^ v[2].z ^ v[3].w ^ v[4].x ^ v[5].y ^ v[6].z ^ v[7].w;
            }
        } else {
            for (uint32_t t = 0; t < S / 16u; t++) { const v4u a = LD(base + (size_t)t * 1024u + 16u * lane); acc ^= a.x ^ a.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int P> void run(const char* name, const uint8_t* d, size_t bytes, uint32_t S, uint32_t grid, uint32_t* d_out) {
    const uint32_t waves = (uint32_t)(bytes / (64ull * S));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<P>, dim3(grid), dim3(64), 0, 0, d, S, waves, d_out); hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<P>, dim3(grid), dim3(64), 0, 0, d, S, waves, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("  %-12s S=%5u  grid %6u  %8.3f ms  %7.1f GB/s\n", name, S, grid, ms, (double)waves * 64.0 * S / ms / 1e6);
}
int main() {
    const size_t bytes = 4ull << 30;
    uint8_t* d; uint32_t* d_out;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&d_out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    for (uint32_t S : {4608u, 512u}) {
        for (uint32_t grid : {256u * 4u, 256u * 8u, 256u * 32u}) {
            run<0>("lane-stream", d, bytes, S, grid, d_out);
            run<1>("quad-coop", d, bytes, S, grid, d_out);
            run<2>("octet-coop", d, bytes, S, grid, d_out);
            run<3>("coalesced", d, bytes, S, grid, d_out);
        }
    }
    return 0;
}
