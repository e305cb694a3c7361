// emu_kernels.cpp -- compiles the SHIPPING kernel bodies (skyplane_amd/csrc/*.inc) for the host under the
// SIMT emulator and exposes a C entry that mirrors libskyhip's launch sequence.  TEST INFRASTRUCTURE ONLY.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "wave.h"          // picks up emu.h because SKY_EMU is defined
#include "skyhip_kernels.h"
#include "lz4_common.inc"
#include "lz4s_kernel.inc"
#include "md5_kernel.inc"
#include "frame_kernel.inc"
#include "lz4d_kernel.inc"
extern "C" uint32_t sky_d_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // see SKY_D_STAT
#ifdef SKY_WITH_CDC
#include "gear_kernel.inc"
#endif

static void k_lz4s(void* a, uint8_t* smem) { sky_lz4s_compress_body(*(SkyLz4Args*)a, smem); }
static void k_lz4s_frames(void* a, uint8_t* smem) { sky_lz4s_frames_body(*(SkyLz4FArgs*)a, smem); }
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
    if ((flags & 5u) == 5u) {      // frames written in place: a workgroup per chunk at a time (what libskyhip does for large device-resident batches)
        std::vector<sky_u64> flen(n);
        SkyLz4FArgs fa; fa.in = in; fa.in_off = off.data(); fa.in_len = len.data(); fa.n_chunks = (uint32_t)n; fa.out = out; fa.out_off = ooff.data();
        fa.frame_len = flen.data(); fa.prof = nullptr;
        uint32_t qhead = 0;
        fa.queue = &qhead;
        if (n) emu_launch(n < 2 ? n : 2, LZ4S_LANES, LZ4S_LDS_BYTES, k_lz4s_frames, &fa);      // two workgroups: each walks several chunks
        for (int i = 0; i < n; i++) frame_len[i] = flen[i];
    } else if (flags & 1u) {
        std::vector<uint8_t> scratch((size_t)(nb ? nb : 1) * SKY_LZ4_SLOT, 0xCD);
        std::vector<uint32_t> csize(nb ? nb : 1), word(nb ? nb : 1);
        std::vector<sky_u64> bdst(nb ? nb : 1), flen(n);
        SkyLz4Args la; la.in = in; la.in_off = off.data(); la.in_len = len.data(); la.blk_prefix = prefix.data();
        la.n_chunks = (uint32_t)n; la.n_blocks = nb; la.scratch = scratch.data(); la.csize = csize.data(); la.ablate = 0; la.prof = nullptr;
        const int grid = nb < 3u ? (int)nb : 3;      // a persistent grid smaller than the block count: every workgroup walks several blocks
        uint32_t qhead = 0;
        la.queue = &qhead;
        if (nb) emu_launch(grid, LZ4S_LANES, LZ4S_LDS_BYTES, k_lz4s, &la);
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

// the slice-parallel kernel on one block; dst = SKY_LZ4_SLOT bytes, written only when the block shrinks.  Returns the size.
uint32_t emu_lz4s_block(const uint8_t* src, uint32_t n, uint8_t* dst) {
    sky_u64 off = 0; uint32_t len = n; uint32_t prefix[2] = {0, 1}; uint32_t cs = 0;
    SkyLz4Args la; la.in = src; la.in_off = &off; la.in_len = &len; la.blk_prefix = prefix; la.n_chunks = 1; la.n_blocks = 1; la.scratch = dst; la.csize = &cs; la.ablate = 0; la.prof = nullptr;
    uint32_t qhead = 0;
    la.queue = &qhead;
    emu_launch(1, LZ4S_LANES, LZ4S_LDS_BYTES, k_lz4s, &la);
    return cs;
}

uint32_t emu_slot_bytes(void) { return SKY_LZ4_SLOT; }

static void k_dscan(void* a, uint8_t*) { sky_lz4f_scan_body(*(SkyLz4dArgs*)a); }
static void k_ddec(void* a, uint8_t* smem) { sky_lz4_decode_body(*(SkyLz4dRun*)a, smem); }
static void k_dseq(void* a, uint8_t* smem) { sky_lz4_decode_seq_body(*(SkyLz4dRun*)a, smem); }
static void k_dparse(void* a, uint8_t* smem) { sky_lz4_parse_body(*(SkyLz4dLink*)a, smem); }
static void k_dlink(void* a, uint8_t* smem) { sky_lz4_link_body(*(SkyLz4dLink*)a, smem); }
static void k_dresolve(void* a, uint8_t* smem) { sky_lz4_resolve_body(*(SkyLz4dResolve*)a, smem); }
static void k_dchain(void* a, uint8_t* smem) { sky_lz4_chain_body(*(SkyLz4dResolve*)a, smem); }
static int emu_link_resolve = 1;      // which of the library's two ways linked frames take (sky_lz4d_run decides by batch size): 1 = resolve + chain, 0 = sky_lz4_link
extern "C" void emu_set_link_resolve(int v) { emu_link_resolve = v; }

// mirrors sky_lz4d_run: scan, build work items, decode.  status[i] = decoder code, out_len[i] = decoded bytes.
int emu_decompress(const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, int n, uint8_t* out, const uint64_t* out_off,
                   const uint64_t* out_cap, uint64_t* out_len, int32_t* status) {
    std::vector<sky_u64> ioff(in_off, in_off + n), ilen(in_len, in_len + n), ooff(out_off, out_off + n), ocap(out_cap, out_cap + n), content(n);
    std::vector<uint32_t> prefix(n + 1), nblk(n), flags(n), bmax(n), err(n);
    uint32_t slots = 0;
    for (int i = 0; i < n; i++) { prefix[i] = slots; slots += 2 * (uint32_t)(ocap[i] / SKY_LZ4_BLOCK) + 16; }
    prefix[n] = slots;
    std::vector<sky_u64> bsrc(slots); std::vector<uint32_t> bword(slots), bframe(slots);
    SkyLz4dArgs a; a.in = in; a.in_off = ioff.data(); a.in_len = ilen.data(); a.out = out; a.out_off = ooff.data(); a.out_cap = ocap.data();
    a.blk_prefix = prefix.data(); a.n = (uint32_t)n; a.f_nblk = nblk.data(); a.f_flags = flags.data(); a.f_bmax = bmax.data(); a.f_content = content.data();
    a.f_err = err.data(); a.b_src = bsrc.data(); a.b_word = bword.data(); a.b_frame = bframe.data(); a.n_slots = slots;
    emu_launch((n + 63) / 64, 64, 0, k_dscan, &a);
    std::vector<uint32_t> items, seq, lframes, lfirst, litems;
    auto kind = [&](int i) -> int {        // see sky_lz4d_run
        if (!nblk[i]) return 0;
        const bool b64 = bmax[i] == SKY_LZ4_BLOCK;
        if ((flags[i] & 4u) && b64) return 'C';
        if (!(flags[i] & 1u) && !(flags[i] & 4u) && b64) return 'B';
        if ((flags[i] & 1u) && !(flags[i] & 4u)) return 'A';
        return 'D';
    };
    for (int i = 0; i < n; i++) {
        const int k = kind(i);
        if (k == 'C') seq.push_back((uint32_t)i);
        else if (k == 'A') for (uint32_t b = 0; b < nblk[i]; b++) items.push_back(prefix[i] + b);
        else if (k == 'D') items.push_back((uint32_t)i | 0x80000000u);
        else if (k == 'B') { lframes.push_back((uint32_t)i); lfirst.push_back((uint32_t)litems.size()); for (uint32_t b = 0; b < nblk[i]; b++) litems.push_back(prefix[i] + b); }
    }
    if (!lframes.empty()) {
        std::vector<sky_u64> desc((size_t)litems.size() * SKY_D_DESC_CAP);
        std::vector<uint32_t> ndesc(litems.size());
        SkyLz4dLink r; r.a = a; r.item_slot = litems.data(); r.n_items = (uint32_t)litems.size(); r.desc = desc.data(); r.ndesc = ndesc.data();
        r.frames = lframes.data(); r.first_item = lfirst.data(); r.n_frames = (uint32_t)lframes.size();
        emu_launch(((int)litems.size() + 3) / 4, 256, 4 * SKY_D_STAGE_LDS, k_dparse, &r);
        if (emu_link_resolve) {
            std::vector<uint16_t> rptr((size_t)litems.size() * SKY_LZ4_BLOCK, 0xCDCD);
            std::vector<uint8_t> rx((size_t)litems.size() * (SKY_LZ4_BLOCK / 8u), 0xCD);
            SkyLz4dResolve rr; rr.l = r; rr.ptr = rptr.data(); rr.xbits = rx.data();
            emu_launch((int)litems.size(), (int)SKY_LZ4R_LANES, SKY_LZ4R_LDS, k_dresolve, &rr);
            emu_launch((int)lframes.size(), (int)SKY_LZ4R_LANES, SKY_LZ4C_LDS, k_dchain, &rr);
        } else {
            emu_launch((int)lframes.size(), (int)SKY_LZ4D_LINK_LANES, SKY_LZ4D_LINK_LDS, k_dlink, &r);
        }
    }
    if (!items.empty()) {
        SkyLz4dRun r; r.a = a; r.item_slot = items.data(); r.n_items = (uint32_t)items.size();
        emu_launch(((int)items.size() + 3) / 4, 256, 4 * SKY_D_STAGE_LDS, k_ddec, &r);
    }
    if (!seq.empty()) {
        SkyLz4dRun r; r.a = a; r.item_slot = seq.data(); r.n_items = (uint32_t)seq.size();
        emu_launch((int)seq.size(), 64, SKY_LZ4D_SEQ_LDS, k_dseq, &r);
    }
    seq.clear();       // frames without a content size that turned out to hold short blocks: second, sequential decode (see sky_lz4d_run)
    for (int i = 0; i < n; i++)
        if (err[i] != 0u && (kind(i) == 'A' || kind(i) == 'B') && (flags[i] & 2u) && nblk[i] > 1u && bmax[i] == SKY_LZ4_BLOCK) { seq.push_back((uint32_t)i); err[i] = 0; }
    if (!seq.empty()) {
        SkyLz4dRun r; r.a = a; r.item_slot = seq.data(); r.n_items = (uint32_t)seq.size();
        emu_launch((int)seq.size(), 64, SKY_LZ4D_SEQ_LDS, k_dseq, &r);
    }
    int rc = 0;
    for (int i = 0; i < n; i++) { status[i] = (int32_t)err[i]; out_len[i] = err[i] ? 0 : content[i]; if (err[i]) rc = -8; }
    return rc;
}

#ifdef SKY_WITH_CDC
static void k_gcand(void* a, uint8_t* smem) { sky_gear_candidates_body(*(SkyGearArgs*)a, smem); }
static void k_gsel(void* a, uint8_t*) { sky_gear_select_body(*(SkyGearArgs*)a); }
static void k_segpre(void* a, uint8_t* smem) { sky_seg_prefix_body(*(SkySegPrefixArgs*)a, smem); }
static void k_segdesc(void* a, uint8_t*) { sky_seg_desc_body(*(SkySegDescArgs*)a); }
static void k_segmd5(void* a, uint8_t*) { sky_segment_md5_body(*(SkySegMd5Args*)a); }
static void k_segmd5x(void* a, uint8_t* smem) { sky_segment_md5x_body(*(SkySegMd5Args*)a, smem); }
static int emu_segmd5_staged = 1;      // which of the library's two segment-digest kernels emu_cdc runs (SKYHIP_SEGMD5_STAGED there)
extern "C" void emu_set_segmd5_staged(int v) { emu_segmd5_staged = v; }
static void k_dins(void* a, uint8_t*) { sky_dedup_insert_body(*(SkyDedupArgs*)a); }
static void k_dres(void* a, uint8_t*) { sky_dedup_resolve_body(*(SkyDedupArgs*)a); }
static void k_litplan(void* a, uint8_t*) { sky_lit_plan_body(*(SkyLitArgs*)a); }
static void k_litgather(void* a, uint8_t*) { sky_lit_gather_body(*(SkyLitArgs*)a); }
static void k_gruns(void* a, uint8_t*) { sky_gather_runs_body(*(SkyRunArgs*)a); }

// Mirrors sky_cdc_run's launch sequence.  gear: 256 x u64 table (the caller passes the ORACLE's table so a
// generator mismatch shows up in the GPU tests, not here).  Returns total segments or <0.
long emu_cdc(const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, int n, const uint64_t* gear, uint32_t* seg_prefix_out,
             uint32_t* seg_end_out, size_t seg_cap, uint8_t* fps_out, uint64_t* first_seen_out, uint64_t* key_lo, uint64_t* key_hi,
             uint64_t* first, uint32_t slots_log2, uint64_t seg_base, int dedup, uint32_t* cand_cnt_out, uint8_t* lit_out, uint64_t lit_stride,
             uint32_t* lit_len_out) {
    std::vector<sky_u64> off(in_off, in_off + n);
    std::vector<uint32_t> len(n), tile_prefix(n + 1), cut_prefix(n + 1), ncuts(n), seg_prefix(n + 1);
    uint32_t tiles = 0, slots = 0;
    for (int i = 0; i < n; i++) {
        len[i] = (uint32_t)in_len[i]; tile_prefix[i] = tiles; cut_prefix[i] = slots;
        tiles += (len[i] + SKY_GEAR_TILE - 1) / SKY_GEAR_TILE; slots += len[i] / SKY_CDC_MIN + 2;
    }
    tile_prefix[n] = tiles; cut_prefix[n] = slots;
    std::vector<uint32_t> cand((size_t)(tiles ? tiles : 1) * SKY_GEAR_CAND_CAP), cand_cnt(tiles ? tiles : 1), cuts(slots), seg_end(slots), seg_slot(slots), err(4, 0);
    std::vector<uint8_t> fps((size_t)slots * 16);
    std::vector<sky_u64> fs(slots);
    SkyGearArgs ga; ga.in = in; ga.in_off = off.data(); ga.in_len = len.data(); ga.tile_prefix = tile_prefix.data(); ga.n_chunks = (uint32_t)n; ga.n_tiles = tiles;
    ga.gear = (const sky_u64*)gear; ga.cand = cand.data(); ga.cand_cnt = cand_cnt.data(); ga.cut_prefix = cut_prefix.data(); ga.cuts = cuts.data(); ga.n_cuts = ncuts.data();
    uint32_t tile_queue = 0;
    ga.queue = &tile_queue;
    if (tiles) emu_launch(tiles < 3u ? (int)tiles : 3, SKY_GEAR_THREADS, SKY_GEAR_LDS_BYTES, k_gcand, &ga);      // a persistent grid smaller than the tile count
    if (cand_cnt_out) for (uint32_t t = 0; t < tiles; t++) cand_cnt_out[t] = cand_cnt[t];
    emu_launch(n, 64, 0, k_gsel, &ga);
    SkySegPrefixArgs pa; pa.n_cuts = ncuts.data(); pa.seg_prefix = seg_prefix.data(); pa.n_chunks = (uint32_t)n;
    emu_launch(1, 256, 64, k_segpre, &pa);
    const uint32_t total = seg_prefix[n];
    if (total > seg_cap) return -5;
    std::vector<sky_u64> desc(slots ? slots : 1);
    SkySegDescArgs sa; sa.in_off = off.data(); sa.cut_prefix = cut_prefix.data(); sa.cuts = cuts.data(); sa.n_cuts = ncuts.data(); sa.seg_prefix = seg_prefix.data();
    sa.n_chunks = (uint32_t)n; sa.desc = desc.data(); sa.seg_end = seg_end.data();
    emu_launch((n + 3) / 4, 256, 0, k_segdesc, &sa);
    SkySegMd5Args ma; ma.in = in; ma.desc = desc.data(); ma.seg_total = &seg_prefix[n]; ma.max_segs = slots; ma.fps = fps.data();
    {
        sky_u64 end = 0;      // one past the last input byte: the staged kernel may read a row's last piece up to there (the guard tests put an unmapped page behind it)
        for (int i = 0; i < n; i++) if (off[i] + len[i] > end) end = off[i] + len[i];
        ma.in_end = in + end;
    }
    if (emu_segmd5_staged) emu_launch(total > 200u ? 2 : 1, 64, SKY_SEGX_LDS, k_segmd5x, &ma);
    else emu_launch(total > 200u ? 2 : 1, 64, 0, k_segmd5, &ma);      // a persistent grid: every lane walks several segments (two wavefronts when there are enough)
    if (dedup) {
        SkyDedupArgs da; da.key_lo = (sky_u64*)key_lo; da.key_hi = (sky_u64*)key_hi; da.first = (sky_u64*)first; da.slot_mask = (1u << slots_log2) - 1u;
        da.fps = fps.data(); da.seg_total = &seg_prefix[n]; da.max_segs = slots; da.seg_base = seg_base; da.seg_slot = seg_slot.data();
        da.first_seen = fs.data(); da.err = err.data();
        emu_launch((slots + 255) / 256, 256, 0, k_dins, &da);
        emu_launch((slots + 255) / 256, 256, 0, k_dres, &da);
        // err[0] = segments that found no slot within the probe bound: reported as "not seen before", not an error
    }
    if (dedup && lit_out) {      // dedup on the wire, source side (skyhip_dedup_literals): every chunk's NEW segments back to back, at lit_out + i * lit_stride
        std::vector<uint32_t> lit_off(slots ? slots : 1), lit_len(n ? n : 1);
        SkyLitArgs la; la.in = in; la.desc = desc.data(); la.first_seen = fs.data(); la.seg_prefix = seg_prefix.data(); la.seg_base = seg_base; la.n_chunks = (uint32_t)n;
        la.lit_off = lit_off.data(); la.lit_len = lit_len.data(); la.lit = lit_out; la.lit_stride = lit_stride;
        emu_launch((n + 3) / 4, 256, 0, k_litplan, &la);
        if (total) emu_launch(total > 9u ? 2 : 1, 256, 0, k_litgather, &la);      // (fewer wavefronts than segments: each copies several)
        for (int i = 0; i < n; i++) lit_len_out[i] = lit_len[i];
    }
    for (int i = 0; i <= n; i++) seg_prefix_out[i] = seg_prefix[i];
    for (uint32_t i = 0; i < total; i++) { seg_end_out[i] = seg_end[i]; if (dedup) first_seen_out[i] = fs[i]; }
    memcpy(fps_out, fps.data(), (size_t)total * 16);
    return (long)total;
}

// skyhip_gather_md5's copy step: run k = len[k] bytes from src[k] to dst[k] (host addresses here)
void emu_gather_runs(const uint64_t* src, const uint64_t* dst, const uint32_t* len, uint32_t n_runs) {
    SkyRunArgs ra; ra.src = (const sky_u64*)src; ra.dst = (const sky_u64*)dst; ra.len = len; ra.n_runs = n_runs;
    if (n_runs) emu_launch(n_runs > 5u ? 2 : 1, 256, 0, k_gruns, &ra);
}
#endif

}  // extern "C"
