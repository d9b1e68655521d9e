// FlashAttention-2 forward for gfx950 on v_mfma_f32_16x16x32_f16: the kernels of
// attn_fwd_mfma16_body.hpp in three wave layouts.
//   m16   four waves (one per SIMD), 64 query rows each: the least LDS traffic per FLOP.  Ragged N, d = 64
//         without whole 512-row tiles, fp32 callers (whose Q it reads as fp32).
//   m16w  eight waves (two per SIMD), 32 query rows each: every K/V fragment is read from LDS by twice
//         as many waves, but the two waves of a SIMD fill each other's matrix-pipe gaps, most of all in
//         the vector-only stretch between two work items (output scaling and stores, first score block
//         of the next item).  10 % fewer cycles; on boxes at the power cap the chip answers with a 10 %
//         lower clock on long sequences (same time), elsewhere it is time: +0.2 ... +4 % at N = 4096 and
//         beyond depending on the box, +10 % at N = 512, +3 % at N = 1024 for (4, 96, N, 128).  The layout
//         of every whole-tile d = 128 shape.
//   m16x  four waves, 128 query rows each, 512-row query tile: d = 64 only (O^T and Q then take the 192
//         accumulation registers they take at d = 128 with 64 rows; the score and probability tiles fill
//         all 256 vector registers).  Half the K/V staging traffic and half the LDS fragment reads per
//         FLOP, which is where d = 64 spends twice the energy per FLOP of d = 128: +7.5 % at
//         (16, 32, 2048, 64), +8 % at N = 8192.  Whole 512-row tiles only.
#pragma once
#include "attn_fwd_mfma.hpp"

#define FLASH_M16_NS m16
#define FLASH_M16_ROWS 64
#define FLASH_M16_WAVES 4
#include "attn_fwd_mfma16_body.hpp"
#undef FLASH_M16_NS
#undef FLASH_M16_ROWS
#undef FLASH_M16_WAVES

#define FLASH_M16_NS m16w
#define FLASH_M16_ROWS 32
#define FLASH_M16_WAVES 8
#include "attn_fwd_mfma16_body.hpp"
#undef FLASH_M16_NS
#undef FLASH_M16_ROWS
#undef FLASH_M16_WAVES

#define FLASH_M16_NS m16x
#define FLASH_M16_ROWS 128
#define FLASH_M16_WAVES 4
#include "attn_fwd_mfma16_body.hpp"
#undef FLASH_M16_NS
#undef FLASH_M16_ROWS
#undef FLASH_M16_WAVES
