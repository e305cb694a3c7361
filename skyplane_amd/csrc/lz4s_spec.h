// lz4s_spec.h -- constants of the slice-parallel LZ4 block parse ("lz4s").
//
// Shared by the kernel (lz4s_kernel.inc) and by the sequential C restatement of the same parse that the tests use as
// a bit-exact checker (tests/model/lz4s_model.c).  The parse is deterministic by construction -- every table update is
// a commutative min, every slice is parsed from read-only tables -- so the GPU, the SIMT emulator and the model must
// produce identical bytes.  What the reference requires of those bytes is only that lz4.frame.decompress
// (skyplane/gateway/gateway_receiver.py:195-201) reproduces the chunk; the parse itself is ours.
#pragma once
#include <stdint.h>

#define LZ4S_BLOCK 65536u      // one LZ4 frame block (BD = 4), one workgroup
#define LZ4S_SLICE 64u         // bytes parsed by one lane
#define LZ4S_LANES 1024u       // lanes per workgroup = slices per block
// Table shape.  Rounds 2-3 kept one candidate per 16 KiB REGION of the block (4096 buckets x 4 regions): four candidates per position, and waves whose
// slices lie in the last region paid for four entries per row and up to four verifications per visit.  Round 4: ONE region, 16384 buckets -- the same
// 64 KiB of LDS, one candidate per position.  On the model that costs 1.0 % of ratio on the Silesia-like stream (frames +1.6 % over the reference's
// block-linked ones, every class within 9 %: scripts/dev/ratio_classes.py) and makes every wavefront as cheap as the first region's used to be.
#ifndef LZ4S_RLOG
#define LZ4S_RLOG 16           // log2 of a table region (16 = the whole block)
#endif
#ifndef LZ4S_Q
#define LZ4S_Q 1               // regions per block
#endif
#ifndef LZ4S_LOGB
#define LZ4S_LOGB 14           // log2 buckets; table = buckets x Q entries of 4 bytes = 64 KiB
#endif
#ifndef LZ4S_EXT
#define LZ4S_EXT 256u          // a match may run this far past the end of its slice (overlaps are trimmed afterwards).  Measured again in round 3: 128 takes 1.0 % and 64
                               // takes 1.8 % off the kernel for 0.03 % / 0.09 % of ratio on the Silesia-like stream, but sparse data (long runs) then exceeds the reference's
                               // frames by 10.1 % / 14.6 % against 8.4 % -- over the 10 % every class is held to (tests/test_gpu_parity.py): stays at 256
#endif
#ifndef LZ4S_BACK
#define LZ4S_BACK 8u           // a match start may move back over at most this many pending literals
#endif
#ifndef LZ4S_INS_STEP
#define LZ4S_INS_STEP 2u       // only every second position enters the table; every position is still looked up.  With a single entry per (bucket, region)
                               // fewer insertions mean fewer evictions: the ratio on the Silesia-like stream is 0.3 % BETTER than with every position
                               // (step 4: 0.8 % worse), a match that starts on an odd position is found one byte later and moved back over the literal
#endif
#define LZ4S_K1 2654435761u
#define LZ4S_K3 0x9E3779u      // 24-bit: the fifth byte goes through a full-rate 24-bit multiply-add on the GPU
#define LZ4S_INF 0xFFFFFFFFu   // empty table entry
#define LZ4S_TAGMASK 0x7FFFu   // 15-bit tags: an empty entry (tag bits 0xFFFF) can never look like a hit

// hash of the five bytes at a position: g = little-endian dword, b4 = the byte after it
#ifndef LZ4S_HASH
#define LZ4S_HASH(g, b4) ((uint32_t)(g) * LZ4S_K1 + (uint32_t)(b4) * LZ4S_K3)
#endif
#define LZ4S_BUCKET(x) ((uint32_t)(x) >> (32 - LZ4S_LOGB))
#define LZ4S_TAG(x) (((uint32_t)(x) >> (32 - LZ4S_LOGB - 15)) & LZ4S_TAGMASK)
// table entry: tag in the high half so that min() keeps the EARLIEST position among equal tags; the position is
// relative to its region
#define LZ4S_ENTRY(tag, relpos) (((uint32_t)(tag) << 16) | (uint32_t)(relpos))

// sequence record produced by the slice parse: offset | literals before the match (from the previous match of the
// slice or the slice start) << 16 | match length << 22
#define LZ4S_REC(off, lit, len) ((uint32_t)(off) | ((uint32_t)(lit) << 16) | ((uint32_t)(len) << 22))
#define LZ4S_REC_OFF(r) ((r) & 0xFFFFu)
#define LZ4S_REC_LIT(r) (((r) >> 16) & 0x3Fu)
#define LZ4S_REC_LEN(r) ((r) >> 22)
