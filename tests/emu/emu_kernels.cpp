// emu_kernels.cpp -- compiles the SHIPPING kernel bodies (skyplane_amd/csrc/*.inc) for the host under the
// SIMT emulator and exposes a C entry that mirrors libskyhip's launch sequence.  TEST INFRASTRUCTURE ONLY.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "wave.h"          // picks up emu.h because SKY_EMU is defined
#include "skyhip_kernels.h"
#include "lz4_kernel.inc"
#include "md5_kernel.inc"
#include "frame_kernel.inc"
#ifdef SKY_WITH_CDC
#include "gear_kernel.inc"
#endif

static void k_lz4(void* a, uint8_t* smem) { sky_lz4_compress_body(*(SkyLz4Args*)a, smem); }
static void k_md5(void* a, uint8_t*) { sky_md5_body(*(SkyMd5Args*)a); }
static void k_layout(void* a, uint8_t*) { sky_frame_layout_body(*(SkyFrameArgs*)a); }
static void k_gather(void* a, uint8_t*) { sky_frame_gather_body(*(SkyFrameArgs*)a); }

extern "C" {

// flags: 1 = lz4 frame, 2 = md5.  All pointers are host memory.  blk_skew: put that many dummy blocks in
// front of the prefix array to exercise the sub-batch (non-zero base) path.
int emu_process(const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, int n, uint8_t* out, const uint64_t* out_off,
                uint64_t* frame_len, uint8_t* md5, uint32_t flags, uint32_t blk_skew, uint32_t* csize_out) {
    std::vector<sky_u64> off(in_off, in_off + n), ooff(n);
    std::vector<uint32_t> len(n), prefix(n + 1);
    uint32_t nb = 0;
    for (int i = 0; i < n; i++) {
        len[i] = (uint32_t)in_len[i];
        ooff[i] = out_off ? out_off[i] : 0;
        prefix[i] = blk_skew + nb;
        nb += (len[i] + SKY_LZ4_BLOCK - 1) / SKY_LZ4_BLOCK;
    }
    prefix[n] = blk_skew + nb;
    if (flags & 2u) {
        SkyMd5Args ma; ma.in = in; ma.off = off.data(); ma.len = len.data(); ma.n = (uint32_t)n; ma.digest = md5;
        emu_launch((n + 63) / 64, 64, 0, k_md5, &ma);
    }
    if (flags & 1u) {
        std::vector<uint8_t> scratch((size_t)(nb ? nb : 1) * SKY_LZ4_SLOT, 0xCD);
        std::vector<uint32_t> csize(nb ? nb : 1), word(nb ? nb : 1);
        std::vector<sky_u64> bdst(nb ? nb : 1), flen(n);
        SkyLz4Args la; la.in = in; la.in_off = off.data(); la.in_len = len.data(); la.blk_prefix = prefix.data();
        la.n_chunks = (uint32_t)n; la.n_blocks = nb; la.scratch = scratch.data(); la.csize = csize.data(); la.ablate = 0; la.prof = nullptr;
        if (nb) emu_launch((nb + SKY_LZ4_WAVES - 1) / SKY_LZ4_WAVES, SKY_LZ4_WAVES * 64, SKY_LZ4_LDS_BYTES, k_lz4, &la);
        SkyFrameArgs fa; fa.in = in; fa.in_off = off.data(); fa.in_len = len.data(); fa.blk_prefix = prefix.data(); fa.n_chunks = (uint32_t)n;
        fa.n_blocks = nb; fa.scratch = scratch.data(); fa.csize = csize.data(); fa.out = out; fa.out_off = ooff.data(); fa.frame_len = flen.data();
        fa.blk_dst = bdst.data(); fa.blk_word = word.data();
        emu_launch((n + 3) / 4, 256, 0, k_layout, &fa);
        if (nb) emu_launch(nb, 256, 0, k_gather, &fa);
        for (int i = 0; i < n; i++) frame_len[i] = flen[i];
        if (csize_out) for (uint32_t b = 0; b < nb; b++) csize_out[b] = csize[b];
    }
    return 0;
}

// raw single-block entry (no frame): returns compressed size
uint32_t emu_lz4_block(const uint8_t* src, uint32_t n, uint8_t* dst /* SKY_LZ4_SLOT bytes */) {
    sky_u64 off = 0; uint32_t len = n; uint32_t prefix[2] = {0, 1}; uint32_t cs = 0;
    SkyLz4Args la; la.in = src; la.in_off = &off; la.in_len = &len; la.blk_prefix = prefix; la.n_chunks = 1; la.n_blocks = 1; la.scratch = dst; la.csize = &cs; la.ablate = 0; la.prof = nullptr;
    emu_launch(1, SKY_LZ4_WAVES * 64, SKY_LZ4_LDS_BYTES, k_lz4, &la);
    return cs;
}

uint32_t emu_slot_bytes(void) { return SKY_LZ4_SLOT; }

}  // extern "C"
