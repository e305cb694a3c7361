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
#ifndef LZ4S_PER1
#define LZ4S_PER1 0            // 1 = distance-1 candidates (runs found at their second byte, where the distance-4 test needs their fifth).  Built and measured in
                               // round 4 together with LZ4S_INS_STEP 3 (which needs them to keep the sparse class inside its guard): the third fewer table updates
                               // save what the extra test costs in probe and parse -- 14.76 against 14.79 ms, for 0.5 % of ratio (profiles/r4_lz4s_variants.txt)
#endif
#ifndef LZ4S_BACK
#define LZ4S_BACK 4u           // a match start may move back over at most this many pending literals (8 until round 4: +0.02 % of frame bytes on the model for one dword
                               // compare instead of two and one select less per visit)
#endif
#ifndef LZ4S_INS_STEP
#define LZ4S_INS_STEP 2u       // positions 0, STEP, 2 STEP, ... of every 64-byte slice enter the table; every position is still looked up.  With a single entry per (bucket, region)
                               // fewer insertions mean fewer evictions: the ratio on the Silesia-like stream is 0.3 % BETTER than with every position
                               // (step 4: 0.8 % worse), a match that starts on an odd position is found one byte later and moved back over the literal
#endif
// Hash of the five bytes at a position (round 4): two FULL-RATE 24-bit multiplies instead of a quarter-rate 32-bit one plus a byte mask --
//   x = lo24(g0) * KA + lo24(g2) * KB   (mod 2^32),   g0 = little-endian dword at p, g2 = the dword at p + 2
// lo24(g0) covers bytes p..p+2 and lo24(g2) bytes p+2..p+4; the kernel has both dwords anyway (g2 is position p + 2's g0) and v_mul_u32_u24 /
// v_mad_u32_u24 ignore the top byte of their operands, so no masking instruction is needed.  The top 14 bits of x depend on all five bytes: bucket.
// Bits 23..8 are the tag: whole bytes, so that "tag << 16 | position" is ONE v_perm_b32 (the six bits the tag shares with the bucket are dead weight; the
// other ten -- and the low 16 bits, a hash of four bytes, would do worse: +0.3 % of frame bytes).  On the model every class compresses as well or better than
// with the 32-bit multiply (scripts/dev/ratio_classes.py).
#ifndef LZ4S_KA
#define LZ4S_KA 0x9E3779u
#endif
#ifndef LZ4S_KB
#define LZ4S_KB 0xC2B2AEu
#endif
#define LZ4S_INF 0xFFFFFFFFu   // empty table entry (tag 0xFFFF, position 0xFFFF: position 65535 never enters the table, so no hit can look like it)
// Positions below this never enter the table (the block's first slice is nobody's candidate).  With every entry >= 64 the probe's test "same tag and
// entry below position s0 + i" is  (entry - (tag << 16 | i)) < s0  -- the constant i folds into the v_perm_b32 that builds the tag word and no
// per-position "s0 + i" is needed: one instruction per position saved for 64 of 65536 candidates.
#define LZ4S_FIRST_INS 64u
#define LZ4S_MUL24(a, b) ((uint32_t)((uint64_t)((uint32_t)(a) & 0xFFFFFFu) * (uint64_t)((uint32_t)(b) & 0xFFFFFFu)))
#define LZ4S_HASH2(g0, g2) (LZ4S_MUL24(g0, LZ4S_KA) + LZ4S_MUL24(g2, LZ4S_KB))
// the same from the dword at p and the byte after it (model, scalar callers)
#ifndef LZ4S_HASH
#define LZ4S_HASH(g, b4) LZ4S_HASH2(g, ((uint32_t)(g) >> 16) | ((uint32_t)(b4) << 16))
#endif
#ifndef LZ4S_BUCKET
#define LZ4S_BUCKET(x) ((uint32_t)(x) >> (32 - LZ4S_LOGB))
#endif
#ifndef LZ4S_TAG
#define LZ4S_TAG(x) (((uint32_t)(x) >> 8) & 0xFFFFu)
#endif
// table entry: tag in the high half so that min() keeps the EARLIEST position among equal tags; the position is
// relative to its region
#define LZ4S_ENTRY(tag, relpos) (((uint32_t)(tag) << 16) | (uint32_t)(relpos))

// sequence record produced by the slice parse: offset | literals before the match (from the previous match of the
// slice or the slice start) << 16 | match length << 22
#define LZ4S_REC(off, lit, len) ((uint32_t)(off) | ((uint32_t)(lit) << 16) | ((uint32_t)(len) << 22))
#define LZ4S_REC_OFF(r) ((r) & 0xFFFFu)
#define LZ4S_REC_LIT(r) (((r) >> 16) & 0x3Fu)
#define LZ4S_REC_LEN(r) ((r) >> 22)
