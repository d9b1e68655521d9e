// FlashAttention-2 forward for gfx950 on v_mfma_f32_16x16x32_f16: the kernels of
// attn_fwd_mfma16_body.hpp in two wave layouts of the same 256-row query tile.
//   m16   four waves (one per SIMD), 64 query rows each: the least LDS traffic per FLOP.
//   m16w  eight waves (two per SIMD), 32 query rows each: every K/V fragment is read from LDS by twice
//         as many waves, but while one wave of a SIMD is in the vector-only stretch between two work
//         items (output scaling and stores, first score block of the next item) the other one keeps
//         the matrix pipe busy.  Same speed on long sequences (the chip is power-limited there: 10 %
//         fewer cycles at a 10 % lower clock), faster where items are short: +10 % at N = 512, +3 % at
//         N = 1024, +1 % at N = 2048 for (4, 96, N, 128).
#pragma once
#include "attn_fwd_mfma.hpp"

#define FLASH_M16_NS m16
#define FLASH_M16_ROWS 64
#include "attn_fwd_mfma16_body.hpp"
#undef FLASH_M16_NS
#undef FLASH_M16_ROWS

#define FLASH_M16_NS m16w
#define FLASH_M16_ROWS 32
#include "attn_fwd_mfma16_body.hpp"
#undef FLASH_M16_NS
#undef FLASH_M16_ROWS
