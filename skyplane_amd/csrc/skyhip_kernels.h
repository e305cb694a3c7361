// skyhip_kernels.h -- constants shared by the kernel bodies, the device wrappers and the test emulator.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define SKY_LZ4_BLOCK 65536u     // LZ4 frame BD=4: 64 KiB blocks (what python-lz4's default emits)
#define SKY_LZ4_SLOT 66048u      // per-block scratch slot >= LZ4_COMPRESSBOUND(65536) = 65809, 512-aligned

#define SKY_FRAME_HDR 15u        // magic 4 + FLG + BD + content size 8 + HC
#define SKY_FRAME_FLG 0x68u      // version 01, block-independent, content size present
#define SKY_FRAME_BD 0x40u       // 64 KiB block max

#define SKY_RAW_FLAG 0x80000000u
