// FlashAttention-2 forward for gfx950, head-dim tiles 64 and 128, fp16 rows of a multiple of 8
// elements, built on v_mfma_f32_16x16x32_f16.
//
// Same structure as attn_fwd_mfma.hpp (transposed products S^T = K Q^T and O^T = V^T P^T so that a
// lane owns one query column; K/V in 3-slot LDS rings, 4-slot in the persistent kernel, filled by
// global->LDS DMA; software pipeline over 32-key blocks with a deferred reference
// update), but with the 16x16x32 MFMA shape instead of 32x32x16.  Under dense MFMA load the chip
// holds a clock well below nominal and the achieved rate is set by energy per MFMA, not by issue
// cycles; the 16x16x32 shape moves half as many accumulator registers per FLOP and sustains a
// visibly higher FLOP rate at the same power (tools/microbench/mfma_clock.cpp).  Its 4-register
// accumulator tiles also make the softmax reference cheap to carry: the first k-step of every score
// tile takes the per-query constant -reference as its srcC operand (four VGPRs per query block), so
// no rank-1 MFMA and no per-score subtraction is needed.
//
// Fragment maps (lane l: n = l & 15, g = l >> 4):
//   A (16 x 32): row n, k = 8g + j.   B (32 x 16): k = 8g + j, column n.   C/D: column n, rows 4g + i.
//   S^T tile (16 keys x 16 queries), k-step ks of KS = D/32: A = K[key n][16-byte chunk ks + KS*g]
//     (one ds_read_b128; the chunk permutation makes the read conflict-free), B = Q[query n][same
//     chunk] from AGPRs.
//     The lane ends up with keys 4g + i of key block kb for its query.
//   P^T fragment of a 32-key block: element j < 4 is key 4g + j of key block 0, j >= 4 key 4g + j - 4
//     of key block 1: exactly the two S^T tiles of the block converted to fp16, no lane movement.
//   O^T tile (16 dv x 16 queries): A = V^T with the same key order: two ds_read_b64_tr_b16 (keys
//     4g.. of rows 0..15 and of rows 16..31 of the block), B = P^T fragment, K = all 32 keys.
//
// This file is the body of every wave layout and is included once per layout by attn_fwd_mfma16.hpp,
// which sets FLASH_M16_NS (the namespace), FLASH_M16_ROWS (query rows per wave) and FLASH_M16_WAVES
// (64 x 4: one wave per SIMD; 32 x 8: two per SIMD; 128 x 4: one per SIMD, d = 64 only).  Names of the
// wave geometry declared here shadow the ones of attn_fwd_mfma.hpp inside the namespace.

namespace flash_hip {
namespace FLASH_M16_NS {

constexpr int kRowsPerWave = FLASH_M16_ROWS;
constexpr int kWaves = FLASH_M16_WAVES;
constexpr int kBr = kWaves * kRowsPerWave;  // query rows per workgroup: 256, or 512 in the 128-rows-per-wave layout
constexpr int kThreads = kWaves * 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Timing-only diagnostic builds (-DFLASH_KNOCKOUT=<mask>, results are wrong): leave one kind of
// per-step work out of the pipeline steps to see what it costs.  1 exponentials, 2 K/V staging,
// 4 LDS fragment reads, 8 conversions of P to 16 bit, 16 row sums, 32 running maximum, 64 the score
// multiply, 128 the output path (scaling, conversion and stores of O).  tools/ceiling.sh builds and times
// the first seven.
#ifndef FLASH_KNOCKOUT
#define FLASH_KNOCKOUT 0
#endif
constexpr int kKnock = FLASH_KNOCKOUT;
#ifndef FLASH_TRACK_MODE
#define FLASH_TRACK_MODE 1
#endif
constexpr int kTrackMode = FLASH_TRACK_MODE;
// Comparison builds (-DFLASH_ROWSUM_DOT2): in the eight-wave layout the row sums are taken on the vector ALU
// (v_dot2_f32_f16 of every packed pair of probabilities with ones, fp32 accumulation, one partial sum per
// lane) instead of on the matrix pipe.
#if defined(FLASH_ROWSUM_DOT2)
constexpr bool kRowsumDot2 = FLASH_M16_ROWS == 32;
#else
constexpr bool kRowsumDot2 = false;
#endif

constexpr int kQ16 = kRowsPerWave / 16;  // 16-row query blocks per wave
constexpr int kRing = 3;                 // tiles per LDS ring
static_assert((kWaves == 4 && kQ16 == 4) || (kWaves == 8 && kQ16 == 2) || (kWaves == 4 && kQ16 == 8),
              "one wave per SIMD with 64 query rows, two with 32 each, or one with 128 (d = 64 only)");

template <int D>
struct Cfg {
  static_assert(D == 64 || D == 128, "compiled head-dim tiles are 64 and 128");
  static constexpr int kDv16 = D / 16;                 // 16-row blocks of O^T
  static constexpr int kKS = D / 32;                   // k-steps of a score tile
  static constexpr int kChunks = D / 8;                // 16-byte chunks per row
  static constexpr int kRowBytes = D * 2;
  static constexpr int kTileBytes = kBc * kRowBytes;
  static constexpr int kLdsBytes = 2 * kRing * kTileBytes;
  static constexpr int kPiecesPerWave = kTileBytes / 1024 / kWaves;
  static constexpr int kBlkBytes = 32 * kRowBytes;     // one 32-key block
};

// 16-byte-chunk XOR swizzle f(row) of an LDS row, chosen per row length so that
//  - ds_read_b128 of chunk ks + KS*g of rows 0..15 (K operand): every 16-lane service group of the
//    instruction touches 16 different 16-byte slots of the 256-byte bank line;
//  - ds_read_b64_tr_b16 of 32 bytes of each of rows 4g..4g+3 (V operand): the 8 rows of a half-wave
//    land on 8 different 32-byte slot pairs.
// 256-byte rows: f = 2*(row & 7) | bit 3 of row.  128-byte rows (two per bank line, the row parity
// picks the half): f = 2*((row >> 1) & 3) | bit 3 of row.
template <int D>
__device__ __forceinline__ int swz(int row) {
  if constexpr (D == 128) {
    return ((row & 7) << 1) | ((row >> 3) & 1);
  } else {
    return (((row >> 1) & 3) << 1) | ((row >> 3) & 1);
  }
}
template <int D>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * Cfg<D>::kRowBytes + ((chunk ^ swz<D>(row)) << 4);
}

// Operand element kind of the matrix instructions: IEEE binary16 or bfloat16.  Registers that hold
// packed 16-bit operands are typed f16x8 / f16x4v in both cases (bit containers); only the instruction
// mnemonics and the float <-> 16-bit conversions differ.
constexpr int kF16 = 0;
constexpr int kBF16 = 1;
struct bf16_out {  // tag type: O is written as bfloat16
  uint16_t bits;
};

// S^T tile: first k-step takes srcC from `c0` (the bias registers or 0), later ones accumulate.
template <int EK = kF16>
__device__ __forceinline__ void mfma_s_first(f32x4& acc, const f16x8& a, const f16x8& b, const f32x4& c0) {
  if constexpr (EK == kBF16) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c0));
  } else {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c0));
  }
}
template <int EK = kF16>
__device__ __forceinline__ void mfma_s_first0(f32x4& acc, const f16x8& a, const f16x8& b) {
  if constexpr (EK == kBF16) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
  } else {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
  }
}
template <bool PAD, int EK = kF16>
__device__ __forceinline__ void mfma_s(f32x4& acc, const f16x8& a, const f16x8& b) {
  if constexpr (EK == kBF16) {
    if constexpr (PAD) {
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
    } else {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
    }
  } else {
    if constexpr (PAD) {
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
    } else {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
    }
  }
}
template <bool PAD, int EK = kF16>
__device__ __forceinline__ void mfma_o(f32x4& acc, const f16x8& a, const f16x8& b) {
  if constexpr (EK == kBF16) {
    if constexpr (PAD) {
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    } else {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
  } else {
    if constexpr (PAD) {
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    } else {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
  }
}

// First P V product of a work item: srcC is the constant 0, so the accumulators need no clearing.
template <int EK = kF16>
__device__ __forceinline__ void mfma_o_first0(f32x4& acc, const f16x8& a, const f16x8& b) {
  if constexpr (EK == kBF16) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
  } else {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
  }
}

// What the P V slots of a pipeline step do
constexpr int kPvAccumulate = 0;  // O += V^T P^T
constexpr int kPvSkip = 1;        // nothing: the very first step of an item has no P yet
constexpr int kPvFirst = 2;       // O  = V^T P^T (first product of an item)

// Row sums on the matrix pipe: one more 16x16x32 product per query block and 32-key block, with an A
// operand whose row 0 is all ones and whose other rows are zero.  Row 0 of the result tile is then the
// sum over the block's 32 keys of every query's probabilities, accumulated in fp32 (exactly: the
// addends are the fp16 values the P V product uses); rows 1..15 stay zero.  In the tile layout row 0
// is register 0 of the lanes with g = 0, so the sum of a query sits in one of its four lanes and the
// other three hold zeros: the four-lane sum of the epilogue collects it unchanged.
// (v_mfma_f32_4x4x4_16b_f16 with an all-ones A sums a lane's own four values and looks cheaper, 2 passes
// against 4, but it holds up the issue of the wave's next instruction for 13 cycles against 8, and two of
// them are needed per block.)
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
// The row sums of query block q live in fixed accumulation registers, right behind O^T and Q
// (D = 128: a[192 + 4q ..], D = 64: a[96 + 4q ..]), named in the operand constraints.  Left to itself
// the register allocator gives the four sums different registers in different pipeline steps and
// rotates them with v_accvgpr moves inside the loop, next to matrix instructions whose result
// latency it does not know.
#define FLASH_LSUM_CASE(Q, REGS128, REGS64, TEXT, ...)                   \
  case Q:                                                                 \
    if constexpr (D == 128) {                                             \
      asm volatile(TEXT : "+{" REGS128 "}"(acc) : __VA_ARGS__);          \
    } else {                                                              \
      asm volatile(TEXT : "+{" REGS64 "}"(acc) : __VA_ARGS__);           \
    }                                                                     \
    break;
#if FLASH_M16_ROWS == 64
#define FLASH_LSUM_SWITCH(q, TEXT, ...)                                   \
  switch (q) {                                                            \
    FLASH_LSUM_CASE(0, "a[192:195]", "a[96:99]", TEXT, __VA_ARGS__)       \
    FLASH_LSUM_CASE(1, "a[196:199]", "a[100:103]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(2, "a[200:203]", "a[104:107]", TEXT, __VA_ARGS__)     \
    default:                                                              \
    FLASH_LSUM_CASE(3, "a[204:207]", "a[108:111]", TEXT, __VA_ARGS__)     \
  }
#elif FLASH_M16_ROWS == 32  // two query blocks per wave: O^T and Q take half the registers
#define FLASH_LSUM_SWITCH(q, TEXT, ...)                                   \
  switch (q) {                                                            \
    FLASH_LSUM_CASE(0, "a[96:99]", "a[48:51]", TEXT, __VA_ARGS__)         \
    default:                                                              \
    FLASH_LSUM_CASE(1, "a[100:103]", "a[52:55]", TEXT, __VA_ARGS__)       \
  }
#else  // eight query blocks per wave, d = 64 only: O^T 128 registers, Q 64
#define FLASH_LSUM_SWITCH(q, TEXT, ...)                                   \
  switch (q) {                                                            \
    FLASH_LSUM_CASE(0, "a[192:195]", "a[192:195]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(1, "a[196:199]", "a[196:199]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(2, "a[200:203]", "a[200:203]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(3, "a[204:207]", "a[204:207]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(4, "a[208:211]", "a[208:211]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(5, "a[212:215]", "a[212:215]", TEXT, __VA_ARGS__)     \
    FLASH_LSUM_CASE(6, "a[216:219]", "a[216:219]", TEXT, __VA_ARGS__)     \
    default:                                                              \
    FLASH_LSUM_CASE(7, "a[220:223]", "a[220:223]", TEXT, __VA_ARGS__)     \
  }
#endif
// `q` must be a compile-time constant at the call site after inlining (the other cases fold away).
template <int D, int EK = kF16>
__device__ __forceinline__ void mfma_rowsum(f32x4& acc, int q, const f16x8& ones_row, const f16x8& p) {
  if constexpr (EK == kBF16) {
    FLASH_LSUM_SWITCH(q, "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0", "v"(ones_row), "v"(p))
  } else {
    FLASH_LSUM_SWITCH(q, "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0", "v"(ones_row), "v"(p))
  }
}
// The accumulation registers of query block q's row sum, tied to their place (no instruction).
template <int D>
__device__ __forceinline__ void pin_lsum(f32x4& acc, int q) {
  if constexpr (kRowsumDot2) {
    asm volatile("" : "+v"(acc));
  } else {
    FLASH_LSUM_SWITCH(q, "", "n"(0))
  }
}

// Two floats to one register of packed 16-bit values of kind EK, round to nearest even
// (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int EK>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2_t x = {lo, hi};
  if constexpr (EK == kBF16) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
  } else {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2));
  }
}
// 1.0 in both halves
template <int EK>
constexpr uint32_t kOnes2 = EK == kBF16 ? 0x3F803F80u : 0x3C003C00u;

// Wait states for the last MFMAs, with the accumulators as operands so that their readers stay
// behind the wait.  The statements are volatile, hence ordered among themselves: the first one
// carries the wait, the others only tie the remaining registers to that order.
template <int NB>
__device__ __forceinline__ void drain_o(f32x4 (&o)[kQ16][NB]) {
  static_assert(NB == 4 || NB == 8, "operand lists are written out for these sizes");
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
    asm volatile("" : "+a"(o[q][0]), "+a"(o[q][1]), "+a"(o[q][2]), "+a"(o[q][3]));
    if constexpr (NB == 8) asm volatile("" : "+a"(o[q][4]), "+a"(o[q][5]), "+a"(o[q][6]), "+a"(o[q][7]));
  }
}
__device__ __forceinline__ void drain_s(f32x4 (&s)[2][kQ16]) {
  if constexpr (kQ16 == 4) {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]),
                   "+v"(s[1][2]), "+v"(s[1][3]));
  } else if constexpr (kQ16 == 2) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));
  } else {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[0][4]), "+v"(s[0][5]),
                   "+v"(s[0][6]), "+v"(s[0][7]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[1][2]), "+v"(s[1][3]),
                   "+v"(s[1][4]), "+v"(s[1][5]), "+v"(s[1][6]), "+v"(s[1][7]));
  }
}

// max / sum over the four lanes (g = 0..3) that share a query column, result in all four.  The 16-lane
// rows are exchanged with v_permlane16_swap / v_permlane32_swap, which run on the vector ALU: no LDS
// round trip.  With both operands equal to x, permlane16_swap leaves rows {0,0,2,2} of x in one
// register and rows {1,1,3,3} in the other; permlane32_swap leaves the lower half twice / the upper
// half twice.  Every lane combines the four values in the same order.
__device__ __forceinline__ float quad_lane_max(float x, int) {
  const uint32_t u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const uint32_t m = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  const auto t = __builtin_amdgcn_permlane32_swap(m, m, false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
__device__ __forceinline__ float quad_lane_sum(float x, int) {
  const uint32_t u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const uint32_t m = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  const auto t = __builtin_amdgcn_permlane32_swap(m, m, false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

template <int D>
struct LaneAddr {
  uint32_t k[Cfg<D>::kKS];    // K[row n][chunk ks + KS*g] of slot 0 / row block 0
  uint32_t v[Cfg<D>::kDv16];  // V rows 4g + (i16 >> 2), 8-byte piece (i16 & 3) of dv block b
};

template <int D>
__device__ __forceinline__ LaneAddr<D> make_lane_addr(int lane, lds_char* k_ring, lds_char* v_ring) {
  constexpr int kKS = Cfg<D>::kKS, kDv16 = Cfg<D>::kDv16;
  LaneAddr<D> la;
  const int n = lane & 15, g = lane >> 4;
  const int tq = n >> 2, tp = n & 3;
  const uint32_t kbase = (uint32_t)(uintptr_t)k_ring, vbase = (uint32_t)(uintptr_t)v_ring;
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) {
    uint32_t a = kbase + (uint32_t)lds_off<D>(n, ks + kKS * g);
    asm volatile("" : "+v"(a));
    la.k[ks] = a;
  }
#pragma unroll
  for (int b = 0; b < kDv16; ++b) {
    uint32_t a = vbase + (uint32_t)(lds_off<D>(4 * g + tq, 2 * b + (tp >> 1)) + 8 * (tp & 1));
    asm volatile("" : "+v"(a));
    la.v[b] = a;
  }
  return la;
}

// K fragment: 16-row block rb (0..3 inside a 64-row tile) of the tile that starts `koff` bytes into the ring.
template <int D>
__device__ __forceinline__ f16x8 read_k(uint32_t koff, const LaneAddr<D>& la, int rb, int ks) {
  return *(FLASH_LDS_CAST(f16x8, (uintptr_t)(la.k[ks] + koff + (uint32_t)(rb * 16 * Cfg<D>::kRowBytes))));
}
// V^T fragment of dv block b for the 32-key block that starts `voff` bytes into the ring.
template <int D>
__device__ __forceinline__ f16x8 read_v(uint32_t voff, const LaneAddr<D>& la, int b) {
  f16x8 a;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const uint32_t addr = la.v[b] + voff + (uint32_t)(jj * 16 * Cfg<D>::kRowBytes);
    const fp16x4_b t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(FLASH_LDS_CAST(fp16x4_b, (uintptr_t)addr));
    const f16x4 tv = __builtin_bit_cast(f16x4, t);
    a[4 * jj + 0] = tv[0];
    a[4 * jj + 1] = tv[1];
    a[4 * jj + 2] = tv[2];
    a[4 * jj + 3] = tv[3];
  }
  return a;
}

// Eight consecutive fp32 values of a buffer (32 bytes at byte offset `voff`, range-checked) as eight fp16
// values in two register pairs: a typed buffer load with a 16-bit destination lets the memory pipeline
// convert (32_32_32_32 FLOAT data, four values per instruction).  It rounds to nearest even like
// v_cvt_pk_f16_f32 (tools/microbench/d16_load.cpp compares them value by value, ties and subnormals
// included); finite values beyond the fp16 range arrive as +-65504 where the instruction gives +-inf, NaN
// stays NaN, out-of-range addresses read as zeros.  It streams at the rate of a plain 128-bit load.
// Issued from inline asm (the compiler has no builtin for it) and therefore not tracked: the caller waits
// with s_waitcnt vmcnt before it uses the registers.
__device__ __forceinline__ void load_kv_f32_as_f16(u32x2& lo, u32x2& hi, srd_t srd, uint32_t voff) {
  asm volatile(
      "s_nop 4\n\t"
      "tbuffer_load_format_d16_xyzw %0, %2, %3, 0 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen\n\t"
      "tbuffer_load_format_d16_xyzw %1, %2, %3, 0 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen offset:16"
      : "=&v"(lo), "=&v"(hi)
      : "v"(voff), "s"(srd)
      : "memory");
}

struct State {
  float mc;       // reference of the lane's query in this query block, in accumulator units (exp2 units / s_scale)
  float alpha;    // pending O factor
};

// Register i of S^T tile kb holds key kv0 + 16*kb + 4*g + i.
__device__ __forceinline__ void mask_tile(f32x4& s, int kv_tile0, int N, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (kv_tile0 + 4 * g + i >= N) s[i] = -INFINITY;
  }
}

// One pipeline step: QK(j+1) (32 MFMAs), PV(j-1) (32 MFMAs), exp(j), in-lane max(j+1).
//   s_cur/s_nxt[kb][q]: S^T tiles of key block kb (16 keys) and query block q (16 queries)
//   p_prev/p_cur[q]:    P^T fragments (32 keys) of query block q
template <int D, bool HAS_QK, bool HAS_PRELOAD, bool MASKED, int PV_MODE = kPvAccumulate, int EK = kF16, bool TRACK = true,
          typename Hook>
__device__ __forceinline__ void pipeline_step(const Hook& hook, uint32_t kb_off, uint32_t vb_off, uint32_t kb_next_off,
                                              const LaneAddr<D>& la, const f32x4 (&bias)[kQ16],
                                              const f16x8 (&qf)[kQ16][Cfg<D>::kKS], f16x8 (&ka)[4],
                                              f32x4 (&o)[kQ16][Cfg<D>::kDv16], const f32x4 (&s_cur)[2][kQ16],
                                              f32x4 (&s_nxt)[2][kQ16], const f16x8 (&p_prev)[kQ16],
                                              f16x8 (&p_cur)[kQ16], float& mx_all, f32x4 (&lsum)[kQ16],
                                              const ScaleSplit sc, const uint64_t pending, uint64_t& vote,
                                              int kv_next0, int N, int lane) {
  constexpr int kKS = Cfg<D>::kKS, kDv16 = Cfg<D>::kDv16;
  constexpr int QK_SLOTS = 2 * kKS * kQ16;  // 32 (D = 128), 16 (D = 64)
  constexpr int PV_SLOTS = kDv16 * kQ16;    // 32, 16
  constexpr int SLOTS = QK_SLOTS + PV_SLOTS;
  constexpr int ELEMS = 8 * kQ16;           // scores per lane and step
  constexpr int MAX_OPS = ELEMS / 2;        // one running maximum over all scores: 3 values, then 2 per op
  constexpr int MAX_DELAY = 2;

  f16x8 va[4];
  float mx = 0.f;

  // exp(j), software-pipelined over the slots.  Element g = 8q + e: key block e / 4, register e % 4.
  float t_exp = 0.f, t_exp_prev = 0.f;
  // A operand of the row-sum product: row n = lane & 15 of the 16 x 32 block, ones in row 0 only
  const uint32_t one2 = (lane & 15) == 0 ? kOnes2<EK> : 0u;
  f16x8 ones_row = __builtin_bit_cast(f16x8, u32x4{one2, one2, one2, one2});
  asm volatile("" : "+v"(ones_row));
  auto pin = [](f16x2& x) __attribute__((always_inline)) {
    uint32_t w = __builtin_bit_cast(uint32_t, x);
    asm volatile("" : "+v"(w));
    x = __builtin_bit_cast(f16x2, w);
  };
  // row sum of query block q's fragment, whose last pair was packed two elements ago
  auto group_add = [&](int q) __attribute__((always_inline)) {
    if constexpr (!(kKnock & 16) && !kRowsumDot2) mfma_rowsum<D, EK>(lsum[q], q, ones_row, p_cur[q]);
  };
  auto finish = [&](int g) __attribute__((always_inline)) {
    const int q = g / 8, e = g % 8;
    if ((g & 7) == 1 && g >= 9) group_add(g / 8 - 1);
    if ((g & 1) != 0 && !(kKnock & 8)) {
      f16x2 w = __builtin_bit_cast(f16x2, pack2<EK>(t_exp_prev, t_exp));
      pin(w);
      p_cur[q][e - 1] = w[0];
      p_cur[q][e] = w[1];
      if constexpr (kRowsumDot2 && EK == kF16) {
        f16x2 one2v;
        one2v[0] = one2v[1] = (f16)1.f;
        lsum[q][0] = __builtin_amdgcn_fdot2(w, one2v, lsum[q][0], false);
      }
    }
    t_exp_prev = t_exp;
  };
  // The scale factor is one v_mul_f32 per score (a packed v_pk_mul_f32 over register pairs issues in
  // twice the time of a plain multiply on this chip and measured 1-2 % slower), issued one score ahead of the
  // exponential that consumes it so that the exponential never waits for its operand.
  float x_cur = 0.f, x_nxt = 0.f;
  auto scaled = [&](int g) __attribute__((always_inline)) {
    // (no register pin here: an asm statement that names the product makes the compiler pad the
    // exponential behind it with a wait state)
    if constexpr (kKnock & 64) return s_cur[(g % 8) / 4][g / 8][g % 4];
    return s_cur[(g % 8) / 4][g / 8][g % 4] * sc.c;
  };
  auto exp_share = [&](int n) __attribute__((always_inline)) {
#pragma unroll
    for (int g = n * ELEMS / SLOTS; g < (n + 1) * ELEMS / SLOTS; ++g) {
      if (g > 0) finish(g - 1);
      if (g == 0) x_cur = scaled(0);
      t_exp = (kKnock & 1) ? x_cur : fast_exp2(x_cur);
      asm volatile("" : "+v"(t_exp));
      if (g + 1 < ELEMS) x_nxt = scaled(g + 1);
      x_cur = x_nxt;
    }
  };
  // In-lane maximum of the next block's scores, all query blocks in one chain: the hot path only asks
  // whether ANY score of the wave exceeds the deferral threshold (the rare path works out which query
  // moved).  Value v = 8q + e is register e % 4 of tile e / 4 of query block q; op 0 folds values 0..2,
  // op u values 2u+1 and 2u+2, the last op value 31.
  auto score = [&](int v) __attribute__((always_inline)) { return s_nxt[(v % 8) / 4][v / 8][v % 4]; };
  auto max_op = [&](int u) __attribute__((always_inline)) {
    if constexpr (kKnock & 32) return;
    if constexpr (MASKED) {
      // keys beyond N of a query block, right before the chain first touches the block
      const int q_new = u == 0 ? 0 : ((2 * u + 2) % 8 == 0 ? (2 * u + 2) / 8 : -1);
      if (q_new >= 0 && q_new < kQ16) {
        mask_tile(s_nxt[0][q_new], kv_next0, N, lane);
        mask_tile(s_nxt[1][q_new], kv_next0 + 16, N, lane);
      }
    }
    if (u == 0) {
      mx = fmaxf(fmaxf(score(0), score(1)), score(2));
    } else if (2 * u + 2 < ELEMS) {
      mx = fmaxf(fmaxf(mx, score(2 * u + 1)), score(2 * u + 2));
    } else {
      mx = fmaxf(mx, score(ELEMS - 1));
      mx_all = mx;
      asm volatile("" : "+v"(mx_all));
      return;
    }
    asm volatile("" : "+v"(mx));
  };

  // ---- QK slots: key block kb, k-step ks, query block q; K fragment index f = 4 kb + ks -------------
#pragma unroll
  for (int f = 0; f < 2 * kKS; ++f) {
#pragma unroll
    for (int q = 0; q < kQ16; ++q) {
      const int n = f * kQ16 + q;
      const int kb = f / kKS, ks = f % kKS;
      if constexpr (HAS_QK) {
        if (ks == 0) {
          mfma_s_first<EK>(s_nxt[kb][q], ka[f % 4], qf[q][ks], bias[q]);
        } else {
          mfma_s<false, EK>(s_nxt[kb][q], ka[f % 4], qf[q][ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (q == kQ16 - 1 && !(kKnock & 4)) {
          if (f + 4 < 2 * kKS) {
            ka[f % 4] = read_k<D>(kb_off, la, (f + 4) / kKS, (f + 4) % kKS);
          } else if constexpr (HAS_PRELOAD) {
            ka[f % 4] = read_k<D>(kb_next_off, la, (f + 4 - 2 * kKS) / kKS, (f + 4) % kKS);
          }
        }
      }
      // the first four V^T fragments, shortly before the PV slots begin
      constexpr int V_FIRST = QK_SLOTS >= 10 ? QK_SLOTS - 10 : 0;
      if (PV_MODE != kPvSkip && !(kKnock & 4) && n >= V_FIRST && n < V_FIRST + 8 && ((n - V_FIRST) & 1) == 0) {
        const int b = (n - V_FIRST) / 2;
        va[b] = read_v<D>(vb_off, la, b);
      }
      exp_share(n);
      if constexpr (!(kKnock & 2)) hook(n);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- PV slots: dv block b, query block q -------------------------------------------------------------
#pragma unroll
  for (int b = 0; b < kDv16; ++b) {
#pragma unroll
    for (int q = 0; q < kQ16; ++q) {
      const int pslot = b * kQ16 + q;
      if constexpr (PV_MODE == kPvFirst) {
        mfma_o_first0<EK>(o[q][b], va[b % 4], p_prev[q]);
      } else if constexpr (PV_MODE == kPvAccumulate) {
        mfma_o<false, EK>(o[q][b], va[b % 4], p_prev[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (PV_MODE != kPvSkip && !(kKnock & 4) && q == kQ16 - 1 && b + 4 < kDv16) va[b % 4] = read_v<D>(vb_off, la, b + 4);
      exp_share(QK_SLOTS + pslot);
      if constexpr (!(kKnock & 2)) hook(QK_SLOTS + pslot);
      if constexpr (HAS_QK && TRACK) {
#pragma unroll
        for (int u = 0; u < MAX_OPS; ++u) {
          if (MAX_DELAY + u * (PV_SLOTS - MAX_DELAY) / MAX_OPS == pslot) max_op(u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  finish(ELEMS - 1);
  // The last pair was packed by the instruction just before; the row-sum MFMA that reads it needs two
  // wait states.  They are filled with the step's vote: one vector compare (any lane whose maximum
  // exceeds the deferral threshold) joined with the pending-rescale mask by one scalar OR.  Written
  // out: from `pending || ballot(...)` the compiler builds the same condition out of five scalar
  // instructions, and at one wave per SIMD every instruction is issue time.
  if constexpr (HAS_QK && TRACK) {
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_or_b64 %0, vcc, %3"
                 : "=s"(vote)
                 : "s"(sc.thr), "v"(mx_all), "s"(pending)
                 : "vcc", "scc", "memory");
  } else {
    vote = pending;
    asm volatile("s_nop 1" : "+s"(vote) : : "memory");
  }
  group_add(kQ16 - 1);
}

// Wait states before the row sums are read by anything but the MFMA that accumulates them.
template <int D>
__device__ __forceinline__ void drain_lsum(f32x4 (&lsum)[kQ16]) {
  asm volatile("s_nop 11" ::: "memory");  // result latency of the row-sum product: 8 passes + 3
#pragma unroll
  for (int q = 0; q < kQ16; ++q) pin_lsum<D>(lsum[q], q);
}

// The bias registers are written by the vector ALU and read as srcC by MFMAs that the compiler does not
// recognise as such: the wait states between the two are provided here, tied to the registers.
__device__ __forceinline__ void bias_ready(f32x4 (&bias)[kQ16]) {
  if constexpr (kQ16 == 4) {
    asm volatile("s_nop 4" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]));
  } else if constexpr (kQ16 == 2) {
    asm volatile("s_nop 4" : "+v"(bias[0]), "+v"(bias[1]));
  } else {
    asm volatile("s_nop 4"
                 : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]), "+v"(bias[4]), "+v"(bias[5]),
                   "+v"(bias[6]), "+v"(bias[7]));
  }
}

// Between two steps: pending O rescale, then the decision whether the next block moves a reference.
// The score tiles and the bias registers, named at a point of the program without an instruction (for the
// register allocator only: at this point they sit in vector registers, whole).
__device__ __forceinline__ void pin_tiles(f32x4 (&s)[2][kQ16], f32x4 (&bias)[kQ16]) {
#pragma unroll
  for (int q = 0; q < kQ16; ++q) asm volatile("" : "+v"(s[0][q]), "+v"(s[1][q]), "+v"(bias[q]));
}

// Steps that do not take the running maximum leave `mx_all` alone: it must then hold a value that moves no
// reference (the compiler cannot see which value it holds, and keeps the rare path of every step).
__device__ __forceinline__ void forget_max(float& mx_all) {
  mx_all = -INFINITY;
  asm volatile("" : "+v"(mx_all));
}

template <int NB, bool HAS_NEXT, bool PIN = false>
__device__ __forceinline__ void step_end(f32x4 (&o)[kQ16][NB], State (&st)[kQ16], const float mx_all,
                                         f32x4 (&lsum)[kQ16], f32x4 (&bias)[kQ16], f32x4 (&s_nxt)[2][kQ16],
                                         uint64_t& pending, const uint64_t vote, const ScaleSplit sc, int lane) {
  // `vote` = pending-rescale mask | lanes whose next-block maximum exceeds the threshold (pipeline_step)
  if constexpr (PIN) pin_tiles(s_nxt, bias);
  if (__builtin_expect(vote != 0, 0)) {
    bool any_need = false;
    if constexpr (HAS_NEXT) any_need = __builtin_amdgcn_ballot_w64(mx_all > sc.thr) != 0;
    if (pending != 0) {
      drain_o(o);
#pragma unroll
      for (int q = 0; q < kQ16; ++q) {
        const float al = st[q].alpha;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          // one tile at a time, fenced: the accumulators pass through vector registers for the multiply,
          // and left alone the compiler reads dozens of them ahead, which this rare path cannot afford
#pragma unroll
          for (int i = 0; i < 4; ++i) o[q][b][i] *= al;
          asm volatile("" : "+a"(o[q][b]));
        }
      }
    }
    pending = 0;
    if (any_need) {
      drain_lsum<NB * 16>(lsum);
#pragma unroll
      for (int q = 0; q < kQ16; ++q) {
        // this query block's own maximum (keys beyond N already carry -inf)
        const f32x4 &t0 = s_nxt[0][q], &t1 = s_nxt[1][q];
        const float mq = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])),
                               fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3])));
        const float rise = fmaxf(quad_lane_max(mq, lane), 0.f);  // references only move up
        st[q].alpha = fast_exp2(-rise * sc.c);
        st[q].mc += rise;
#pragma unroll
        for (int i = 0; i < 4; ++i) lsum[q][i] *= st[q].alpha;
        pin_lsum<NB * 16>(lsum[q], q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s_nxt[0][q][i] -= rise;
          s_nxt[1][q][i] -= rise;
          bias[q][i] = -st[q].mc;
        }
      }
      bias_ready(bias);
      pending = ~0ull;
    }
  }
}

// ---- epilogue of one 16-query block: O = O^T / l, written as fp16 or as fp32 ----------------------------
// Tile b of the block holds dv = 16b + 4g + 0..3 of the lane's query.
//  * fp16 output: two neighbouring tiles are exchanged between 16-lane rows (v_permlane16_swap) so
//    that a lane owns 8 consecutive columns of one of them and writes one 128-bit store.
//  * fp32 output (callers whose tensors are fp32): the four accumulators of a tile are 16 consecutive
//    bytes of the output row already, one 128-bit store per tile.
// Rows beyond N and columns beyond d fall outside the descriptor range and are dropped.
template <typename TO>
struct OutStores;
template <>
struct OutStores<f16> {
  static constexpr int per_block(int dv16) { return dv16 / 2; }
};
template <>
struct OutStores<float> {
  static constexpr int per_block(int dv16) { return dv16; }
};
template <>
struct OutStores<bf16_out> {
  static constexpr int per_block(int dv16) { return dv16 / 2; }
};
// operand element kind that goes with an output type (bfloat16 in -> bfloat16 out; fp32 callers feed
// fp16 copies)
template <typename TO>
constexpr int kElemKindOf = std::is_same<TO, bf16_out>::value ? kBF16 : kF16;

template <int NB, typename TO, int AUX = 0>
__device__ __forceinline__ void store_o_block(const f32x4 (&oq)[NB], float inv, uint32_t qrow, int d, int g,
                                              __amdgpu_buffer_rsrc_t o_rs) {
  if constexpr ((kKnock & 128) != 0) return;  // timing only: what scaling, converting and storing O costs
  if constexpr (sizeof(TO) == 2) {
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      // packed multiplies, then packed conversions: the pairs come out in the layout the stores need
      const f32x4 xs = oq[b] * inv, ys = oq[b + 1] * inv;
      constexpr int EK = kElemKindOf<TO>;
      const u32x2 xu = {pack2<EK>(xs[0], xs[1]), pack2<EK>(xs[2], xs[3])};
      const u32x2 yu = {pack2<EK>(ys[0], ys[1]), pack2<EK>(ys[2], ys[3])};
      u32x4 out;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        // odd rows of the first operand trade places with even rows of the second
        auto r = __builtin_amdgcn_permlane16_swap(xu[w], yu[w], false, false);
        out[w] = r[0];      // g even: own tile b, cols 4g..      g odd: tile b+1 of row g-1
        out[2 + w] = r[1];  // g even: tile b of row g+1          g odd: own tile b+1
      }
      const int dv = 16 * (b + (g & 1)) + 8 * (g >> 1);
      const uint32_t off = (dv < d) ? (qrow * (uint32_t)d + (uint32_t)dv) * 2u : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(out, o_rs, off, 0, AUX);
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4 x;
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = oq[b][j] * inv;
      const int dv = 16 * b + 4 * g;
      const uint32_t off = (dv < d) ? (qrow * (uint32_t)d + (uint32_t)dv) * 4u : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), o_rs, off, 0, AUX);
    }
  }
}

// fp32 callers (TO = float): K and V come as fp16 copies out of the conversion pass (every K/V row is
// read by N / Br workgroups), Q is read by exactly one workgroup and is taken from the caller's fp32
// tensor directly, rounded to fp16 (nearest even, as the conversion pass rounds) on its way into the
// registers: the pass moves a third fewer bytes.
// (In the four-wave, 64-rows-per-wave layout only.  The 128-rows-per-wave layout has no vector registers
// left to land fp32 Q in and neither has the eight-wave layout, which fp32 callers only get at N <= 512:
// there the pass converts Q as well.)
template <typename TO>
constexpr bool kQFromF32 = std::is_same<TO, float>::value && kQ16 == 4;

// Q fragment (8 consecutive head-dim elements of one query row) from its two 16-byte fp32 halves
__device__ __forceinline__ f16x8 q_frag_from_f32(const u32x4& lo, const u32x4& hi) {
  u32x4 w;
  w[0] = pack2<kF16>(__uint_as_float(lo[0]), __uint_as_float(lo[1]));
  w[1] = pack2<kF16>(__uint_as_float(lo[2]), __uint_as_float(lo[3]));
  w[2] = pack2<kF16>(__uint_as_float(hi[0]), __uint_as_float(hi[1]));
  w[3] = pack2<kF16>(__uint_as_float(hi[2]), __uint_as_float(hi[3]));
  return __builtin_bit_cast(f16x8, w);
}

// KV32: K and V are the caller's fp32 tensors as well (TO = float, Q read as fp32 too): every wave loads
// its share of a tile with 128-bit loads, rounds it to fp16 (nearest even) and writes the swizzled LDS
// image itself, where the fp16 kernels let the global->LDS DMA do it.  No conversion pass, no workspace:
// the route of fp32 callers whose K/V rows have few readers (short sequences) or who bring no scratch.
template <int D, typename TO = f16, bool KV32 = false>
__global__ __launch_bounds__(kThreads, 1) void attn_fwd_kernel_m16(const FwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using C = Cfg<D>;
  constexpr int EK = kElemKindOf<TO>;
  constexpr int kKS = C::kKS, kDv16 = C::kDv16, kTileBytes = C::kTileBytes, kBlkBytes = C::kBlkBytes;
  constexpr int kPiecesPerWave = C::kPiecesPerWave;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_char* const smem = (lds_char*)smem_raw;
  FLASH_STAMP(stamp_start);

  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot_id = bid >> 3;
  const uint32_t per = p.total_wg >> 3, rem = p.total_wg & 7u;
  const uint32_t item = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot_id;
  const uint32_t head = item / (uint32_t)p.nqt;
  const uint32_t qt = item - head * (uint32_t)p.nqt;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15;
  const int g = lane >> 4;
  const int N = p.N, d = p.d;
  const ScaleSplit sc{p.s_scale, p.defer_thr};

  const size_t head_elems = (size_t)N * (size_t)d;
  const uint32_t head_bytes = (uint32_t)(head_elems * 2);
  constexpr bool QF32 = kQFromF32<TO>;
  static_assert(!KV32 || QF32, "fp32 K and V come with fp32 Q and fp32 O (four-wave layout)");
  constexpr uint32_t kQElem = QF32 ? 4u : 2u;
  constexpr uint32_t kKVElem = KV32 ? 4u : 2u;
  const char* qh = (const char*)p.q + (size_t)head * head_elems * kQElem;
  const char* kh = (const char*)p.k + (size_t)head * head_elems * kKVElem;
  const char* vh = (const char*)p.v + (size_t)head * head_elems * kKVElem;
  TO* oh = (TO*)p.o + (size_t)head * head_elems;
  const __amdgpu_buffer_rsrc_t q_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)qh, 0, (uint32_t)(head_elems * kQElem), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)oh, 0, (uint32_t)(head_elems * sizeof(TO)), 0x00020000);
  const srd_t k_srd = make_srd(kh, head_bytes * (kKVElem / 2u));
  const srd_t v_srd = make_srd(vh, head_bytes * (kKVElem / 2u));

  const uint32_t row_bytes = (uint32_t)d * kKVElem;
  const uint32_t tile_step = (uint32_t)kBc * row_bytes;

  lds_char* const k_ring = smem;
  lds_char* const v_ring = smem + kRing * kTileBytes;

  // staging: piece pc = 1 KiB of consecutive LDS chunks; lane i of it fills chunk (64 pc + i) % kChunks
  // of row (64 pc + i) / kChunks with the source chunk that the swizzle maps there (8 elements: 16 bytes
  // of an fp16 row, 32 bytes of an fp32 row)
  uint32_t dma_off[kPiecesPerWave];
#pragma unroll
  for (int i = 0; i < kPiecesPerWave; ++i) {
    const int ci = (wave * kPiecesPerWave + i) * 64 + lane;
    const int row = ci / C::kChunks;
    const int src = (ci % C::kChunks) ^ swz<D>(row);
    dma_off[i] = (src * 8 < d) ? (uint32_t)row * row_bytes + (uint32_t)src * 8u * kKVElem : kOob;
  }
  auto piece = [&](const srd_t srd, lds_char* ring, int slot, int t, int i) __attribute__((always_inline)) {
    const int pc = wave * kPiecesPerWave + i;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ring + slot * kTileBytes + pc * 1024));
    dma_piece(srd, dma_off[i] + (uint32_t)t * tile_step, lds_base);
  };
  // fp32 K/V: the memory pipeline converts (typed loads with a 16-bit destination, see load_kv_f32_as_f16):
  // the two halves of piece i's chunk arrive as fp16 in two register pairs ...
  u32x2 land[KV32 ? kPiecesPerWave : 1][2];
  auto load_piece = [&](const srd_t srd, int t, int i) __attribute__((always_inline)) {
    // (the tile offset is part of the per-lane offset here, so that rows beyond N fail the range check and
    // read as zeros; an out-of-range chunk keeps an out-of-range offset: kOob plus a tile offset stays
    // below 2^31)
    load_kv_f32_as_f16(land[i][0], land[i][1], srd, dma_off[i] + (uint32_t)t * tile_step);
  };
  // ... and are written to the place the DMA would have filled.  The compiler does not see these loads:
  // the wait counts the younger loads of the same tile (vector memory loads complete in order).
  const uint32_t lds_lane = (uint32_t)(wave * kPiecesPerWave * 1024 + lane * 16);
  auto write_piece = [&](lds_char* ring, int slot, int i) __attribute__((always_inline)) {
    switch (2 * (kPiecesPerWave - 1 - i)) {  // (an immediate operand)
      case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(land[i][0]), "+v"(land[i][1]) : : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(land[i][0]), "+v"(land[i][1]) : : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(land[i][0]), "+v"(land[i][1]) : : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(6)" : "+v"(land[i][0]), "+v"(land[i][1]) : : "memory"); break;
    }
    *FLASH_LDS_CAST(u32x4, ring + slot * kTileBytes + i * 1024 + lds_lane) =
        u32x4{land[i][0][0], land[i][0][1], land[i][1][0], land[i][1][1]};
  };
  auto stage = [&](const srd_t srd, lds_char* ring, int slot, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) piece(srd, ring, slot, t, i);
  };
  auto staged_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nt = (N + kBc - 1) / kBc;

  // ---- Q fragments, as the caller stored them: softmax_scale * log2(e) multiplies the fp32 scores -------
  // The loads are issued before the K/V staging traffic (loads return in order).
  int qrow[kQ16];
  f16x8 qf[kQ16][kKS];
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
    qrow[q] = (int)qt * kBr + wave * kRowsPerWave + q * 16 + n16;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      const int col = 8 * (ks + kKS * g);
      const uint32_t off = (col < d) ? ((uint32_t)qrow[q] * (uint32_t)d + (uint32_t)col) * kQElem : kOob;
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(q_rs, off, 0, 0);
      if constexpr (QF32) {
        qf[q][ks] = q_frag_from_f32(raw, __builtin_amdgcn_raw_buffer_load_b128(q_rs, off + 16u, 0, 0));
      } else {
        qf[q][ks] = __builtin_bit_cast(f16x8, raw);
      }
    }
  }
  if constexpr (KV32) {
    // all three tiles' loads in flight at once (the score and probability registers are still free):
    // one memory round trip for the cold start instead of three.  Tile 1 of a one-tile sequence lies
    // beyond the head: its loads return zeros.
    u32x2 cold[3][kPiecesPerWave][2];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i) {
        load_kv_f32_as_f16(cold[w][i][0], cold[w][i][1], w == 1 ? v_srd : k_srd, dma_off[i] + (w == 2 ? tile_step : 0u));
      }
    }
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(cold[w][i][0]), "+v"(cold[w][i][1]) : : "memory");
        *FLASH_LDS_CAST(u32x4, (w == 1 ? v_ring : k_ring) + (w == 2 ? kTileBytes : 0) + i * 1024 + lds_lane) =
            u32x4{cold[w][i][0][0], cold[w][i][0][1], cold[w][i][1][0], cold[w][i][1][1]};
      }
    }
  } else {
    stage(k_srd, k_ring, 0, 0);
    stage(v_srd, v_ring, 0, 0);
    if (nt > 1) stage(k_srd, k_ring, 1, 1);
  }
  // Step 0 multiplies P(-1) = 0 with the second block of "V tile -1" (slot 2): make that finite.
  for (int i = tid * 16; i < kBlkBytes; i += kThreads * 16) {
    *FLASH_LDS_CAST(u32x4, v_ring + 2 * kTileBytes + kBlkBytes + i) = u32x4{0u, 0u, 0u, 0u};
  }
  f32x4 o[kQ16][kDv16];
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) asm volatile("" : "+a"(qf[q][ks]));
#pragma unroll
    for (int b = 0; b < kDv16; ++b) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[q][b][i] = 0.f;
      asm volatile("" : "+a"(o[q][b]));
    }
  }
  const LaneAddr<D> la = make_lane_addr<D>(lane, k_ring, v_ring);

  staged_barrier();

  // ---- prologue: S(0), its row max, the first K fragments of block 1 ----------------------------------
  f32x4 sA[2][kQ16], sB[2][kQ16];
  f16x8 pA[kQ16], pB[kQ16];
  f16x8 ka[4];
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) pB[q][j] = (f16)0.f;
    asm volatile("" : "+v"(pB[q]));  // see the persistent kernel: keeps the zeros from being re-materialised
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) ka[f] = read_k<D>(0u, la, f / kKS, f % kKS);
#pragma unroll
  for (int f = 0; f < 2 * kKS; ++f) {
#pragma unroll
    for (int q = 0; q < kQ16; ++q) {
      if (f % kKS == 0) {
        mfma_s_first0<EK>(sA[f / kKS][q], ka[f % 4], qf[q][f % kKS]);
      } else {
        mfma_s<true, EK>(sA[f / kKS][q], ka[f % 4], qf[q][f % kKS]);
      }
    }
    if (f + 4 < 2 * kKS) {
      ka[f % 4] = read_k<D>(0u, la, (f + 4) / kKS, (f + 4) % kKS);
    } else {
      ka[f % 4] = read_k<D>((uint32_t)kBlkBytes, la, (f + 4 - 2 * kKS) / kKS, (f + 4) % kKS);
    }
  }
  drain_s(sA);
  State st[kQ16];
  float mx_all = 0.f;  // in-lane maximum of the block whose scores were computed last
  uint64_t vote = 0;   // see pipeline_step / step_end
  f32x4 bias[kQ16];
  f32x4 lsum[kQ16];  // row sums: register 0 of the lanes with g = 0, zeros elsewhere (see mfma_rowsum)
  uint64_t pending = 0;
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
    mask_tile(sA[0][q], 0, N, lane);
    mask_tile(sA[1][q], 16, N, lane);
    float mx = fmaxf(fmaxf(sA[0][q][0], sA[0][q][1]), fmaxf(sA[0][q][2], sA[0][q][3]));
    mx = fmaxf(mx, fmaxf(fmaxf(sA[1][q][0], sA[1][q][1]), fmaxf(sA[1][q][2], sA[1][q][3])));
    const float mc = quad_lane_max(mx, lane);  // block 0 always holds key 0, so this is finite
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sA[0][q][i] -= mc;
      sA[1][q][i] -= mc;
      bias[q][i] = -mc;
    }
    st[q].mc = mc;
#pragma unroll
    for (int i = 0; i < 4; ++i) lsum[q][i] = 0.f;
    pin_lsum<D>(lsum[q], q);
    st[q].alpha = 1.f;
  }
  bias_ready(bias);

  FLASH_STAMP(stamp_loop);
  auto tile_iter = [&](auto slot_tag, auto masked_tag, auto last_tag, int t, int slot_rt) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    const int s0 = SLOT >= 0 ? SLOT : slot_rt;
    const int s1 = s0 == 2 ? 0 : s0 + 1;
    const int s2 = s1 == 2 ? 0 : s1 + 1;
    // staging spread over the slots of the first step: K tile t+2, then V tile t+1
    auto dma_hook = [&](int n) __attribute__((always_inline)) {
      constexpr int kSpacing = (2 * kKS + kDv16) * kQ16 / (2 * kPiecesPerWave);
      if (n % kSpacing == 1) {
        const int i = n / kSpacing;
        if (i < kPiecesPerWave) {
          if (t + 2 < nt) piece(k_srd, k_ring, s2, t + 2, i);
        } else if (i < 2 * kPiecesPerWave) {
          if (t + 1 < nt) piece(v_srd, v_ring, s1, t + 1, i - kPiecesPerWave);
        }
      }
    };
    // fp32 K/V: one tile per step, K tile t+2 in the first and V tile t+1 in the second.  Its loads go out
    // in the step's first slots and are converted and written in its last ones, a step's worth of matrix
    // work (about a microsecond) later.  Both targets are slots nobody reads in this iteration.
    auto vec_hook = [&](int n, bool v_side) __attribute__((always_inline)) {
      constexpr int kSlots = (2 * kKS + kDv16) * kQ16;
      constexpr int kFirstWrite = kSlots - 2 * kPiecesPerWave - 2;
      const bool go = !MASKED || (v_side ? t + 1 < nt : t + 2 < nt);  // the unmasked iterations always stage
      if (n >= 1 && n < 1 + 2 * kPiecesPerWave && (n - 1) % 2 == 0) {
        if (go) load_piece(v_side ? v_srd : k_srd, v_side ? t + 1 : t + 2, (n - 1) / 2);
      }
      if (n >= kFirstWrite && n < kFirstWrite + 2 * kPiecesPerWave && (n - kFirstWrite) % 2 == 0) {
        if (go) write_piece(v_side ? v_ring : k_ring, v_side ? s1 : s2, (n - kFirstWrite) / 2);
      }
    };
    auto hook1 = [&](int n) __attribute__((always_inline)) {
      if constexpr (KV32) {
        vec_hook(n, false);
      } else {
        dma_hook(n);
      }
    };
    auto hook2 = [&](int n) __attribute__((always_inline)) {
      if constexpr (KV32) vec_hook(n, true);
    };
    const uint32_t k_cur = (uint32_t)(s0 * kTileBytes);
    const uint32_t k_nxt = (uint32_t)(s1 * kTileBytes);
    const uint32_t v_cur = (uint32_t)(s0 * kTileBytes);
    const uint32_t v_prv = (uint32_t)(s2 * kTileBytes);
    pipeline_step<D, true, !LAST, MASKED, kPvAccumulate, EK>(hook1, k_cur + kBlkBytes, v_prv + kBlkBytes, k_nxt, la, bias, qf, ka, o, sA, sB,
                                       pB, pA, mx_all, lsum, sc, pending, vote, t * kBc + 32, N, lane);
    step_end<kDv16, true>(o, st, mx_all, lsum, bias, sB, pending, vote, sc, lane);
    pipeline_step<D, !LAST, !LAST, MASKED, kPvAccumulate, EK>(hook2, k_nxt, v_cur, k_nxt + kBlkBytes, la, bias, qf, ka, o, sB, sA, pA, pB, mx_all,
                                        lsum, sc, pending, vote, (t + 1) * kBc, N, lane);
    step_end<kDv16, !LAST>(o, st, mx_all, lsum, bias, sA, pending, vote, sc, lane);
    staged_barrier();
  };
  using std::integral_constant;
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  int t = 0;
  for (; t + 5 <= nt; t += 3) {
    tile_iter(integral_constant<int, 0>{}, no, no, t, 0);
    tile_iter(integral_constant<int, 1>{}, no, no, t + 1, 1);
    tile_iter(integral_constant<int, 2>{}, no, no, t + 2, 2);
  }
  for (; t + 1 < nt; ++t) tile_iter(integral_constant<int, -1>{}, yes, no, t, t % 3);
  tile_iter(integral_constant<int, -1>{}, yes, yes, t, t % 3);
  FLASH_STAMP(stamp_tail);
  // PV of the last block
  {
    drain_o(o);
    const uint32_t vb = (uint32_t)(((nt - 1) % 3) * kTileBytes + kBlkBytes);
#pragma unroll
    for (int b = 0; b < kDv16; ++b) {
      const f16x8 vf = read_v<D>(vb, la, b);
#pragma unroll
      for (int q = 0; q < kQ16; ++q) mfma_o<true, EK>(o[q][b], vf, pB[q]);
    }
    drain_o(o);
    drain_lsum<D>(lsum);
  }

  // ---- epilogue: O = O^T / l ---------------------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
    const float inv = __builtin_amdgcn_rcpf(quad_lane_sum(lsum[q][0], lane));  // v_rcp_f32, 1 ulp
    store_o_block<kDv16, TO>(o[q], inv, (uint32_t)qrow[q], d, g, o_rs);
  }
  FLASH_STAMP(stamp_end);
  FLASH_STAMP_STORE(p, wave, lane, stamp_start, stamp_loop, stamp_tail, stamp_end);
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- persistent variant ---------------------------------------------------------------------------
// For N % 256 == 0 (whole query tiles, a K/V tile count that is a multiple of four, no key mask).
// One workgroup per CU works through a list of (head, query tile) items.  The K/V staging never
// stops at an item boundary: the rings have four slots, tile t of every item sits in slot t % 4, and
// the last two tile iterations of an item already stage the first tiles of the next one, so the next
// item finds K tiles 0, 1 and V tile 0 in LDS.  Its Q is fetched (raw, straight into the
// accumulation registers that hold Q) while the current item's last P V product and output stores
// run; the matrix instructions read it there as the caller stored it (the whole softmax scale is
// applied to the fp32 scores).  What remains between two items is S(0) of the next item; the HBM
// round trips of a cold prologue are paid once per workgroup instead of once per item.
constexpr int kRingP = 4;

template <int D>
struct CfgP : Cfg<D> {
  static constexpr int kLdsBytes = 2 * kRingP * Cfg<D>::kTileBytes;
};

// 16 bytes per lane from a raw buffer straight into accumulation registers.  Not tracked by the
// compiler: the caller waits (s_waitcnt vmcnt) before the registers are used.
// `soff` is a wave-uniform byte offset that the range check does not see (it checks `voff`): the caller knows
// that voff + soff stays inside the buffer whenever voff does.
__device__ __forceinline__ void load_q_raw(f16x8& dst, srd_t srd, uint32_t voff, uint32_t soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// Eight 16-bit values, held as two register pairs of the accumulation file, to 16 bytes of LDS at
// addr + off (off: a compile-time constant after inlining, below 64 KiB).  Not tracked by the compiler: a
// barrier that publishes the data waits with its own s_waitcnt lgkmcnt(0).
__device__ __forceinline__ void lds_write_chunk(uint32_t addr, const u32x2& lo, const u32x2& hi, uint32_t off) {
  asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%3+8"
               :
               : "v"(addr), "a"(lo), "a"(hi), "i"(off)
               : "memory");
}

// Wait for the Q loads: all vector memory operations but the YOUNGER most recent ones.  The registers
// are operands of the statement: their readers must not be moved in front of the wait.
template <int YOUNGER, int KS>
__device__ __forceinline__ void wait_q(f16x8 (&qf)[kQ16][KS]) {
  static_assert(KS == 2 || KS == 4, "operand lists are written out for these sizes");
#pragma unroll
  for (int q = 0; q < kQ16; ++q) {
    if constexpr (KS == 4) {
      asm volatile("s_waitcnt vmcnt(%4)"
                   : "+a"(qf[q][0]), "+a"(qf[q][1]), "+a"(qf[q][2]), "+a"(qf[q][3])
                   : "n"(YOUNGER)
                   : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%2)" : "+a"(qf[q][0]), "+a"(qf[q][1]) : "n"(YOUNGER) : "memory");
    }
  }
}

// KV32: as in attn_fwd_kernel_m16, K and V are the caller's fp32 tensors and every wave converts its share
// of a tile on its way into LDS (TO = float, four-wave layout).
template <int D, typename TO = f16, bool KV32 = false>
__global__ __launch_bounds__(kThreads, 1) void attn_fwd_kernel_m16p(const FwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using C = CfgP<D>;
  constexpr int EK = kElemKindOf<TO>;
  constexpr int kKS = C::kKS, kDv16 = C::kDv16, kTileBytes = C::kTileBytes, kBlkBytes = C::kBlkBytes;
  constexpr int kPiecesPerWave = C::kPiecesPerWave;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_char* const smem = (lds_char*)smem_raw;
  // set by any wave whose item overflowed in the first sweep (see below)
  __shared__ uint32_t overflowed;
  FLASH_STAMP(stamp_entry);

  // Items are dealt to the 8 XCDs in contiguous ranges (workgroup ids equal mod 8 share an XCD);
  // the workgroups of an XCD take the items of its range round-robin, so at any time they work on
  // neighbouring query tiles of the same few heads and share their K/V through the XCD's L2.
  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot_id = bid >> 3;
  const uint32_t wgs_on_xcd = (gridDim.x - xcd + 7u) >> 3;
  const uint32_t per = p.total_wg >> 3, rem = p.total_wg & 7u;
  const uint32_t lo = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per;
  const uint32_t hi = lo + per + (xcd < rem ? 1u : 0u);
  const uint32_t first_item = lo + slot_id;
  if (first_item >= hi) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15;
  const int g = lane >> 4;
  const int N = p.N, d = p.d;
  const ScaleSplit sc_all{p.s_scale, p.defer_thr};
  const int nt = N / kBc;  // multiple of 4
  if (tid == 0) overflowed = 0u;  // published by the barrier that ends the cold start
  // where the first sweep ended: the item it was working on, and whether it gave that item up
  uint32_t sweep_end_item = 0u;
  bool sweep_gave_up = false;

  const size_t head_elems = (size_t)N * (size_t)d;
  const uint32_t head_bytes = (uint32_t)(head_elems * 2);
  constexpr uint32_t kKVElem = KV32 ? 4u : 2u;
  static_assert(!KV32 || (std::is_same<TO, float>::value && (kQ16 <= 4 || D == 64)),
                "fp32 K and V come with fp32 Q and fp32 O (four- and eight-wave layouts)");
  const uint32_t row_bytes = (uint32_t)d * kKVElem;
  const uint32_t tile_step = (uint32_t)kBc * row_bytes;

  lds_char* const k_ring = smem;
  lds_char* const v_ring = smem + kRingP * kTileBytes;

  // A sweep over work items: `start_item` from a cold start, then every wgs_on_xcd-th item after it with the
  // staging running across the item boundaries.  SPARSE: only a few steps of an item take the running
  // maximum (see below); an item whose probabilities left the fp16 range on the way sets `overflowed`, its
  // output is not valid, and the sweep ends two K/V tiles into the item after it.
  // Everything a lane keeps for a whole sweep (LDS and buffer offsets of its fragments, pieces and Q chunks)
  // is worked out inside, from a lane number the compiler cannot recognise as the same in both sweeps of the
  // kernel: the two then share no vector register, and neither constrains the other's allocation.
  auto sweep = [&](auto sparse_tag, const uint32_t start_item) __attribute__((always_inline)) {
  constexpr bool SPARSE = decltype(sparse_tag)::value;
  // A sweep that looks at few blocks moves its references sooner (half the deferral threshold: 4 exp2 units):
  // what a look has seen then lies at most that far above the reference, and 12 units remain before a
  // probability in a block nobody looked at leaves the fp16 range.
  const ScaleSplit sc{sc_all.c, SPARSE ? 0.5f * sc_all.thr : sc_all.thr};
  // (the lane's number inside its wave, counted by the hardware: no register carries it from sweep to sweep)
  int lane_opaque = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane_opaque));
  const int lane = lane_opaque, n16 = lane & 15, g = lane >> 4;

  uint32_t dma_off[kPiecesPerWave];
#pragma unroll
  for (int i = 0; i < kPiecesPerWave; ++i) {
    const int ci = (wave * kPiecesPerWave + i) * 64 + lane;
    const int row = ci / C::kChunks;
    const int src = (ci % C::kChunks) ^ swz<D>(row);
    dma_off[i] = (src * 8 < d) ? (uint32_t)row * row_bytes + (uint32_t)src * 8u * kKVElem : kOob;
  }
  // fp32 K/V: the memory pipeline converts.  A typed buffer load with a 16-bit destination
  // (tbuffer_load_format_d16_xyzw on 32_32_32_32 FLOAT data) reads four fp32 values per lane and delivers
  // them as four fp16 values in two registers, rounded to nearest even like v_cvt_pk_f16_f32 (compared value
  // by value, ties and subnormals included, in tools/microbench/d16_load.cpp; finite values beyond the fp16
  // range come out as +-65504 where the instruction gives +-inf), at the rate of a plain 128-bit load.  Two
  // of them fetch the 32 source bytes of piece i's chunk; they land in ACCUMULATION registers, which LDS
  // instructions take as data directly: two ds_write_b64 put the chunk where the DMA would have put it.  No
  // vector-ALU instruction touches the tile on its way (the previous form: 128-bit loads, v_accvgpr_read of
  // every register, v_cvt_pk_f16_f32: 24 vector instructions per chunk).  The compiler sees neither the
  // loads nor the LDS writes: write_piece waits by counting the younger loads of the same tile (vector
  // memory loads complete in order), the barriers that publish a tile wait for the LDS writes themselves.
  auto kv_srd = [&](const void* base, uint32_t hd) __attribute__((always_inline)) {
    return make_srd((const char*)base + (size_t)hd * head_elems * kKVElem, head_bytes * (kKVElem / 2u));
  };
  u32x2 land_kv[1][KV32 ? kPiecesPerWave : 1][2];
  // Tile t of the head behind `srd`.  Whole tiles only: the tile offset bypasses the range check as a scalar
  // offset, so no per-lane address arithmetic (a chunk beyond d stays out of range through its per-lane
  // offset).
  auto load_piece = [&](bool v_side, const srd_t srd, int t, int i) __attribute__((always_inline)) {
    u32x2 (&land)[KV32 ? kPiecesPerWave : 1][2] = land_kv[0];
    (void)v_side;
    const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)t * tile_step);
    // (s_nop 4: with the scalar registers as short as they are in this kernel the compiler keeps uniform
    // values in vector registers and moves them over right in front of their use; see settle_scalars)
    asm volatile(
        "s_nop 4\n\t"
        "tbuffer_load_format_d16_xyzw %0, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen\n\t"
        "tbuffer_load_format_d16_xyzw %1, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen offset:16"
        : "=a"(land[i][0]), "=a"(land[i][1])
        : "v"(dma_off[i]), "s"(srd), "s"(soff)
        : "memory");
  };
  // this lane's 16 bytes of its wave's first piece in slot 0 of either ring; slot and piece add a constant
  const uint32_t lds_k_lane = (uint32_t)(uintptr_t)k_ring + (uint32_t)(wave * kPiecesPerWave * 1024 + lane * 16);
  const uint32_t lds_v_lane = (uint32_t)(uintptr_t)v_ring + (uint32_t)(wave * kPiecesPerWave * 1024 + lane * 16);
  // `younger`: vector memory loads issued after this piece's two and before this wait
#define FLASH_WAIT_CASE(N) \
  case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+a"(lo), "+a"(hi) : : "memory"); break;
  auto wait_piece = [&](u32x2& lo, u32x2& hi, int younger) __attribute__((always_inline)) {
    switch (younger) {  // (an immediate operand)
      FLASH_WAIT_CASE(0) FLASH_WAIT_CASE(2) FLASH_WAIT_CASE(4) FLASH_WAIT_CASE(6) FLASH_WAIT_CASE(8)
      FLASH_WAIT_CASE(10) FLASH_WAIT_CASE(12)
      default: asm volatile("s_waitcnt vmcnt(14)" : "+a"(lo), "+a"(hi) : : "memory"); break;
    }
  };
#undef FLASH_WAIT_CASE
  auto write_piece = [&](bool v_side, int slot, int i, int younger) __attribute__((always_inline)) {
    u32x2 (&land)[KV32 ? kPiecesPerWave : 1][2] = land_kv[0];
    if constexpr ((kKnock & 256) == 0) wait_piece(land[i][0], land[i][1], younger);  // (256: timing only, what the waits cost)
    lds_write_chunk(v_side ? lds_v_lane : lds_k_lane, land[i][0], land[i][1], (uint32_t)(slot * kTileBytes + i * 1024));
  };
  // LDS address of this wave's first piece in slot 0 of the K ring; the ring (K or V), the slot and the
  // piece number add a compile-time constant
  const uint32_t lds_wave_base =
      __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(k_ring + wave * kPiecesPerWave * 1024));
  // `soff` = tile number * tile_step, wave-uniform: whole tiles only (N % 64 == 0), so the tile offset can
  // bypass the range check as a scalar offset
  auto piece = [&](const srd_t srd, bool v_side, int slot, uint32_t soff, int i) __attribute__((always_inline)) {
    dma_piece_at(srd, dma_off[i], soff, lds_wave_base, (uint32_t)(((v_side ? kRingP : 0) + slot) * kTileBytes + i * 1024));
  };
  auto staged_barrier = [&]() __attribute__((always_inline)) {
    if constexpr (KV32) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the tile's LDS writes are this wave's own)
    } else if constexpr ((kKnock & 512) != 0) {
      // timing only: what waiting for the tiles staged in this very iteration costs (a ring staged one tile
      // further ahead would wait like this)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kPiecesPerWave) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (1024: timing only, what meeting the other waves once per tile costs)
    if constexpr ((kKnock & 1024) == 0) __syncthreads();
  };
  auto head_of = [&](uint32_t it) __attribute__((always_inline)) { return it / (uint32_t)p.nqt; };
  // (with fp32 K and V the eight-wave layout reads Q as fp32 too: all of a wave's Q lands in the score,
  // probability and fragment registers that are free between two items)
  constexpr bool QF32 = kQFromF32<TO> || KV32;
  constexpr uint32_t kQElem = QF32 ? 4u : 2u;
  // Cache policy of the output stores.  O is written once and never read back: in the long-sequence
  // d = 128 kernel the stores go through to memory (sc0 sc1 nt), so that O lines neither sit dirty in the
  // L2 that serves the K/V re-reads nor wait for the flush at the kernel's end: +0.3 % at (4,96,4096,128),
  // +1 % at (8,96,4096,128), +0.7 % at N = 8192 in the four-wave layout.  Short sequences and d = 64
  // measured 1 % slower with it, the eight-wave layout the same with and without.
  constexpr int kOutAux = (D == 128 && kQ16 == 4) ? 19 : 0;

  // per-lane byte offsets of the Q chunks inside a head, for query block 0 of query tile 0: the tile (kBr
  // rows) and the query block (16 rows) are a wave-uniform offset that the loads take from a scalar
  // register (whole tiles only: every row exists, so only the chunk, per lane, needs the range check; a
  // chunk beyond d stays out of range through its per-lane part).  One register per k-step lives through
  // the kernel instead of one per k-step and query block.
  uint32_t q_off[kKS];
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) {
    const int col = 8 * (ks + kKS * g);
    const uint32_t row = (uint32_t)(wave * kRowsPerWave + n16);
    q_off[ks] = (col < d) ? (row * (uint32_t)d + (uint32_t)col) * kQElem : kOob;
  }
  const uint32_t q_tile_bytes = (uint32_t)kBr * (uint32_t)d * kQElem;
  const uint32_t q_block_bytes = 16u * (uint32_t)d * kQElem;

  f16x8 qf[kQ16][kKS];
  auto fetch_q = [&](uint32_t it) __attribute__((always_inline)) {
    const uint32_t hd = head_of(it), qt = it - hd * (uint32_t)p.nqt;
    srd_t q_srd = make_srd((const f16*)p.q + (size_t)hd * head_elems, head_bytes);
    uint32_t soff[kQ16];
#pragma unroll
    for (int q = 0; q < kQ16; ++q) soff[q] = __builtin_amdgcn_readfirstlane(qt * q_tile_bytes + (uint32_t)q * q_block_bytes);
    settle_scalars(q_srd, soff);
#pragma unroll
    for (int q = 0; q < kQ16; ++q) {
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) load_q_raw(qf[q][ks], q_srd, q_off[ks], soff[q]);
    }
  };

  // fp32 Q: ordinary loads into vector registers (the compiler tracks them), converted later.  They are
  // issued when the score and probability registers of the finished item are free, during its epilogue,
  // in phases of at most 64 landing registers (the epilogue needs room of its own): phase ph covers
  // query blocks [ph * kQPerPhase, (ph + 1) * kQPerPhase) and is converted behind the output stores of
  // the same query blocks.
  // (fp16 K and V out of the conversion pass, fp32 Q; with fp32 K and V as well Q comes through converting
  // loads instead, see fetch_q_direct.)
  constexpr bool QLAND = QF32 && !KV32;
  constexpr int kQPhases = (QLAND && (kQ16 * kKS > 8 || (kWaves == 8 && D == 128))) ? 2 : 1;
  constexpr int kQPerPhase = kQ16 / kQPhases;
  u32x4 q_land[QLAND ? kQPerPhase : 1][QLAND ? kKS : 1][2];
  auto fetch_q32 = [&](uint32_t it, int ph) __attribute__((always_inline)) {
    if constexpr (QLAND) {
      const uint32_t hd = head_of(it), qt = it - hd * (uint32_t)p.nqt;
      const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)p.q + (size_t)hd * head_elems * 4u), 0, (uint32_t)(head_elems * 4u), 0x00020000);
#pragma unroll
      for (int q = 0; q < kQPerPhase; ++q) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          const uint32_t soff = __builtin_amdgcn_readfirstlane(
              qt * q_tile_bytes + (uint32_t)(ph * kQPerPhase + q) * q_block_bytes);
          q_land[q][ks][0] = __builtin_amdgcn_raw_buffer_load_b128(q_rs, q_off[ks], soff, 0);
          q_land[q][ks][1] = __builtin_amdgcn_raw_buffer_load_b128(q_rs, q_off[ks], soff + 16u, 0);
        }
      }
    }
  };
  auto convert_q32 = [&](int ph) __attribute__((always_inline)) {
    if constexpr (QLAND) {
#pragma unroll
      for (int q = 0; q < kQPerPhase; ++q) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          f16x8& dst = qf[ph * kQPerPhase + q][ks];
          dst = q_frag_from_f32(q_land[q][ks][0], q_land[q][ks][1]);
          asm volatile("" : "+a"(dst));
        }
      }
    }
  };

  // fp32 Q with fp32 K and V: two converting loads per fragment (see load_kv_f32_as_f16) deliver its eight
  // fp16 values in two register pairs; all of a wave's Q lands in 4 * kQ16 * kKS vector registers (the score,
  // probability and fragment registers are free between an item's last P V product and the next item's
  // first scores) and moves to the Q registers when it has arrived.  No conversion on the vector ALU, one phase.  The compiler does not see the loads:
  // q_arrived waits (counting the younger vector memory operations) before the pairs are read.
  u32x2 q_pair[KV32 ? kQ16 : 1][KV32 ? kKS : 1][2];
  auto kLandInAcc = [](int q) { return D == 128 && q >= kQ16 / 2; };
  auto fetch_q_direct = [&](uint32_t it) __attribute__((always_inline)) {
    if constexpr (KV32) {
      const uint32_t hd = head_of(it), qt = it - hd * (uint32_t)p.nqt;
      srd_t q_srd = make_srd((const char*)p.q + (size_t)hd * head_elems * 4u, head_bytes * 2u);
      uint32_t soff[kQ16];
#pragma unroll
      for (int q = 0; q < kQ16; ++q) soff[q] = __builtin_amdgcn_readfirstlane(qt * q_tile_bytes + (uint32_t)q * q_block_bytes);
      settle_scalars(q_srd, soff);
#pragma unroll
      for (int q = 0; q < kQ16; ++q) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          // (half of the pairs land in accumulation registers: the Q registers themselves are free then)
          if (kLandInAcc(q)) {
            asm volatile(
                "tbuffer_load_format_d16_xyzw %0, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen\n\t"
                "tbuffer_load_format_d16_xyzw %1, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen offset:16"
                : "=&a"(q_pair[q][ks][0]), "=&a"(q_pair[q][ks][1])
                : "v"(q_off[ks]), "s"(q_srd), "s"(soff[q])
                : "memory");
          } else {
            asm volatile(
                "tbuffer_load_format_d16_xyzw %0, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen\n\t"
                "tbuffer_load_format_d16_xyzw %1, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen offset:16"
                : "=&v"(q_pair[q][ks][0]), "=&v"(q_pair[q][ks][1])
                : "v"(q_off[ks]), "s"(q_srd), "s"(soff[q])
                : "memory");
          }
        }
      }
    }
  };
  auto q_arrived = [&](auto younger_tag) __attribute__((always_inline)) {
    if constexpr (KV32) {
      constexpr int YOUNGER = decltype(younger_tag)::value;
#pragma unroll
      for (int q = 0; q < kQ16; ++q) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          if (kLandInAcc(q)) {
            asm volatile("s_waitcnt vmcnt(%2)" : "+a"(q_pair[q][ks][0]), "+a"(q_pair[q][ks][1]) : "n"(YOUNGER) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q_pair[q][ks][0]), "+v"(q_pair[q][ks][1]) : "n"(YOUNGER) : "memory");
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kQ16; ++q) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          const u32x4 w = {q_pair[q][ks][0][0], q_pair[q][ks][0][1], q_pair[q][ks][1][0], q_pair[q][ks][1][1]};
          qf[q][ks] = __builtin_bit_cast(f16x8, w);
          asm volatile("" : "+a"(qf[q][ks]));
        }
      }
    }
  };

  uint32_t item = start_item;
  // ---- cold start: Q, K tiles 0 and 1, V tile 0 of the first item --------------------------------------
  srd_t k_srd = kv_srd(p.k, head_of(item)), v_srd = kv_srd(p.v, head_of(item));
  settle_scalars(k_srd, v_srd);
  if constexpr (KV32) {
    fetch_q_direct(item);
  } else if constexpr (QF32) {
    fetch_q32(item, 0);
  } else {
    fetch_q(item);
  }
  if constexpr (KV32) {
    // K tiles 0 and 1 and V tile 0 in flight together with Q, through the same converting loads as in the
    // loop; here they land in vector registers (the score and probability registers are still free) and go
    // to LDS through ordinary stores.
    u32x2 cold[3][kPiecesPerWave][2];
    auto cold_load = [&](int w, const srd_t srd, uint32_t toff) __attribute__((always_inline)) {
      const uint32_t soff = __builtin_amdgcn_readfirstlane(toff);
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i) {
        asm volatile(
            "s_nop 4\n\t"
            "tbuffer_load_format_d16_xyzw %0, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen\n\t"
            "tbuffer_load_format_d16_xyzw %1, %2, %3, %4 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen offset:16"
            : "=&v"(cold[w][i][0]), "=&v"(cold[w][i][1])
            : "v"(dma_off[i]), "s"(srd), "s"(soff)
            : "memory");
      }
    };
    auto cold_write = [&](int w, lds_char* dst) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i) {
        const u32x4 chunk = {cold[w][i][0][0], cold[w][i][0][1], cold[w][i][1][0], cold[w][i][1][1]};
        *FLASH_LDS_CAST(u32x4, dst + i * 1024 + (wave * kPiecesPerWave * 1024 + lane * 16)) = chunk;
      }
    };
    cold_load(0, k_srd, 0u);
    cold_load(1, v_srd, 0u);
    cold_load(2, k_srd, tile_step);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(cold[w][i][0]), "+v"(cold[w][i][1]) : : "memory");
    }
    cold_write(0, k_ring);
    cold_write(1, v_ring);
    cold_write(2, k_ring + kTileBytes);
    __builtin_amdgcn_sched_barrier(0);
    q_arrived(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) piece(k_srd, false, 0, 0u, i);
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) piece(v_srd, true, 0, 0u, i);
    uint32_t one_tile = __builtin_amdgcn_readfirstlane(tile_step);
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) piece(k_srd, false, 1, one_tile, i);
  }
  const LaneAddr<D> la = make_lane_addr<D>(lane, k_ring, v_ring);
  if constexpr (KV32) {
    // (arrived above, with the cold K/V loads)
  } else if constexpr (QF32) {
    convert_q32(0);
    if constexpr (kQPhases > 1) {
      fetch_q32(item, 1);
      convert_q32(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staging pieces
  } else {
    wait_q<0>(qf);
  }
  __syncthreads();

  f32x4 o[kQ16][kDv16];
  f32x4 sA[2][kQ16], sB[2][kQ16];
  f16x8 pA[kQ16], pB[kQ16];
  f16x8 ka[4];
  State st[kQ16];
  float mx_all = 0.f;  // in-lane maximum of the block whose scores were computed last
  forget_max(mx_all);
  uint64_t vote = 0;   // see pipeline_step / step_end
  f32x4 bias[kQ16];
  f32x4 lsum[kQ16];

#if defined(FLASH_PHASE_STAMPS)
  unsigned long long acc_pro = 0, acc_loop = 0, acc_tail = 0, n_items = 0;  // summed over the workgroup's items
  FLASH_STAMP(stamp_warm);  // end of the cold start
#endif
  for (;;) {
    FLASH_STAMP(stamp_item);
    const uint32_t next_item = item + wgs_on_xcd;
    const bool has_next = next_item < hi;
    const uint32_t next_head = has_next ? head_of(next_item) : head_of(item);
    const uint32_t head = head_of(item), qt = item - head * (uint32_t)p.nqt;
    TO* oh = (TO*)p.o + (size_t)head * head_elems;
    const __amdgpu_buffer_rsrc_t o_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)oh, 0, (uint32_t)(head_elems * sizeof(TO)), 0x00020000);

    // ---- item prologue: S(0) and its row max.  O is not cleared: the first P V product of the item
    // (second step of tile 0) takes the constant 0 as its addend.
#pragma unroll
    for (int f = 0; f < 4; ++f) ka[f] = read_k<D>(0u, la, f / kKS, f % kKS);
#pragma unroll
    for (int f = 0; f < 2 * kKS; ++f) {
#pragma unroll
      for (int q = 0; q < kQ16; ++q) {
        if (f % kKS == 0) {
          mfma_s_first0<EK>(sA[f / kKS][q], ka[f % 4], qf[q][f % kKS]);
        } else {
          mfma_s<true, EK>(sA[f / kKS][q], ka[f % 4], qf[q][f % kKS]);
        }
      }
      if (f + 4 < 2 * kKS) {
        ka[f % 4] = read_k<D>(0u, la, (f + 4) / kKS, (f + 4) % kKS);
      } else {
        ka[f % 4] = read_k<D>((uint32_t)kBlkBytes, la, (f + 4 - 2 * kKS) / kKS, (f + 4) % kKS);
      }
    }
    drain_s(sA);
    uint64_t pending = 0;
#pragma unroll
    for (int q = 0; q < kQ16; ++q) {
      float mx = fmaxf(fmaxf(sA[0][q][0], sA[0][q][1]), fmaxf(sA[0][q][2], sA[0][q][3]));
      mx = fmaxf(mx, fmaxf(fmaxf(sA[1][q][0], sA[1][q][1]), fmaxf(sA[1][q][2], sA[1][q][3])));
      const float mc = quad_lane_max(mx, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sA[0][q][i] -= mc;
        sA[1][q][i] -= mc;
        bias[q][i] = -mc;
      }
      st[q].mc = mc;
#pragma unroll
      for (int i = 0; i < 4; ++i) lsum[q][i] = 0.f;
      pin_lsum<D>(lsum[q], q);
      st[q].alpha = 1.f;
      }
    bias_ready(bias);

    // ---- tile iterations; SLOT = t % 4 is a compile-time constant -------------------------------------
    FLASH_STAMP(stamp_loop);
    // CROSS marks the last two tile iterations of an item, whose staging targets (K tile t+2, V tile
    // t+1) may already be the first tiles of the next item; all others stage from the current head.
    auto tile_iter = [&](auto slot_tag, auto cross_tag, auto last_tag, auto first_tag, auto track_tag, int t) __attribute__((always_inline)) {
      constexpr int S0 = decltype(slot_tag)::value;
      // which of the tile's two steps take the running maximum of the block they compute (bit 0: first step)
      constexpr bool TRACK1 = (decltype(track_tag)::value & 1) != 0, TRACK2 = (decltype(track_tag)::value & 2) != 0;
      constexpr bool CROSS = decltype(cross_tag)::value;
      constexpr bool LAST = decltype(last_tag)::value;
      // tile 0 of an item: its first step has no P to multiply yet, its second step starts O
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr int S1 = (S0 + 1) % 4, S2 = (S0 + 2) % 4, S3 = (S0 + 3) % 4;
      const bool k_cur_item = !CROSS || t + 2 < nt, v_cur_item = !CROSS || t + 1 < nt;
      srd_t ks_srd = k_srd, vs_srd = v_srd;
      int k_tile = t + 2, v_tile = t + 1;
      if constexpr (CROSS) {
        if (!k_cur_item) {
          ks_srd = kv_srd(p.k, next_head);
          k_tile = t + 2 - nt;
        }
        if (!v_cur_item) {
          vs_srd = kv_srd(p.v, next_head);
          v_tile = t + 1 - nt;
        }
      }
      const bool k_go = k_cur_item || has_next, v_go = v_cur_item || has_next;
      uint32_t k_soff = __builtin_amdgcn_readfirstlane((uint32_t)k_tile * tile_step);
      uint32_t v_soff = __builtin_amdgcn_readfirstlane((uint32_t)v_tile * tile_step);
      // the descriptors and tile offsets of another head are computed here, not in front of their first use
      // (the loads of fp32 tiles carry their own wait states)
      if constexpr (CROSS && !KV32) settle_scalars(ks_srd, vs_srd, k_soff, v_soff);
      auto dma_hook = [&](int n) __attribute__((always_inline)) {
        constexpr int kSpacing = (2 * kKS + kDv16) * kQ16 / (2 * kPiecesPerWave);
        if (n % kSpacing == 1) {
          const int i = n / kSpacing;
          if (i < kPiecesPerWave) {
            if (!CROSS || k_go) piece(ks_srd, false, S2, k_soff, i);
          } else if (i < 2 * kPiecesPerWave) {
            if (!CROSS || v_go) piece(vs_srd, true, S1, v_soff, i - kPiecesPerWave);
          }
        }
      };
      // fp32 K/V: one tile per step, the K tile in the first and the V tile in the second; its loads go out
      // in the step's first slots and its LDS writes sit in the last ones, a step's worth of matrix work
      // (about a microsecond) later.  A piece's wait counts the loads behind it, the rest of its own tile.
      // (Both tiles' loads in the first step and both tiles' writes at the end of the second, or the V loads
      // half a step early, were built and measured: 2-4 % slower and the same, respectively.)
      auto vec_hook = [&](int n, bool v_side) __attribute__((always_inline)) {
        constexpr int kSlots = (2 * kKS + kDv16) * kQ16;
        constexpr int kFirstWrite = kSlots - 2 * kPiecesPerWave - 2;
        const bool go = !CROSS || (v_side ? v_go : k_go);
        if (n >= 1 && n < 1 + 2 * kPiecesPerWave && (n - 1) % 2 == 0) {
          if (go) load_piece(v_side, v_side ? vs_srd : ks_srd, v_side ? v_tile : k_tile, (n - 1) / 2);
        }
        if (n >= kFirstWrite && n < kFirstWrite + 2 * kPiecesPerWave && (n - kFirstWrite) % 2 == 0) {
          const int i = (n - kFirstWrite) / 2;
          if (go) write_piece(v_side, v_side ? S1 : S2, i, 2 * (kPiecesPerWave - 1 - i));
        }
      };
      auto hook1 = [&](int n) __attribute__((always_inline)) {
        if constexpr (KV32) {
          vec_hook(n, false);
        } else {
          dma_hook(n);
        }
      };
      auto hook2 = [&](int n) __attribute__((always_inline)) {
        if constexpr (KV32) vec_hook(n, true);
      };
      constexpr uint32_t k_cur = (uint32_t)(S0 * kTileBytes), k_nxt = (uint32_t)(S1 * kTileBytes);
      constexpr uint32_t v_cur = (uint32_t)(S0 * kTileBytes), v_prv = (uint32_t)(S3 * kTileBytes);
      pipeline_step<D, true, !LAST, false, FIRST ? kPvSkip : kPvAccumulate, EK, TRACK1>(
          hook1, k_cur + kBlkBytes, v_prv + kBlkBytes, k_nxt, la, bias, qf, ka, o, sA, sB, pB, pA, mx_all, lsum, sc, pending, vote, 0, N, lane);
      step_end<kDv16, true>(o, st, mx_all, lsum, bias, sB, pending, vote, sc, lane);
      if constexpr (TRACK1 && !TRACK2) forget_max(mx_all);
      if constexpr (LAST) {
        // The item's last Q K^T product has been issued, Q is dead: fetch the next item's Q now.  The
        // loads have the last step, the last P V product and the output stores to land.
        if constexpr (!QF32) {
          if (has_next) fetch_q(next_item);
        }
      }
      pipeline_step<D, !LAST, !LAST, false, FIRST ? kPvFirst : kPvAccumulate, EK, TRACK2>(
          hook2, k_nxt, v_cur, k_nxt + kBlkBytes, la, bias, qf, ka, o, sB, sA, pA, pB, mx_all, lsum, sc, pending, vote, 0, N, lane);
      step_end<kDv16, !LAST>(o, st, mx_all, lsum, bias, sA, pending, vote, sc, lane);
      if constexpr (TRACK2 && SPARSE) forget_max(mx_all);
      if constexpr (LAST) {
        // wait for this wave's staging pieces, which are older than the Q loads, but not for those
        if (has_next && !QF32) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kQ16 * kKS) : "memory");
        } else if constexpr (KV32) {
          // every fp32 tile load of this iteration has been waited for by the piece that wrote it to LDS: the
          // barrier only needs those writes
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      } else {
        staged_barrier();
      }
    };
    using std::integral_constant;
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    // Steps that take the running maximum (kTrackMode): every step, or tile 0 and the first step of every
    // fourth tile after it, or none.
#ifndef FLASH_TRACK_FIRST
#define FLASH_TRACK_FIRST 3
#endif
#ifndef FLASH_TRACK_LOOP
#define FLASH_TRACK_LOOP 2
#endif
    constexpr integral_constant<int, !SPARSE ? 3 : (kTrackMode == 1 ? FLASH_TRACK_FIRST : 0)> trk_first{};
    constexpr integral_constant<int, !SPARSE ? 3 : (kTrackMode == 1 ? FLASH_TRACK_LOOP : 0)> trk_loop{};
    constexpr integral_constant<int, !SPARSE ? 3 : 0> trk_rest{};
    tile_iter(integral_constant<int, 0>{}, no, no, yes, trk_first, 0);
    // Every wave has passed the barrier that ends tile 0, hence finished the item before, output and overflow
    // included: all waves read the same value here.  The value is looked at one tile later, when it has
    // long arrived.  (Timing-only knock-out builds compute garbage on purpose and carry on.)
    uint32_t seen = 0u;
    if constexpr (SPARSE && kKnock == 0) {
      seen = __hip_atomic_load(&overflowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    tile_iter(integral_constant<int, 1>{}, no, no, no, trk_rest, 1);
    if constexpr (SPARSE && kKnock == 0) {
      if (item != start_item && __builtin_expect(__builtin_amdgcn_readfirstlane(seen) != 0u, 0)) {
        sweep_gave_up = true;
        break;
      }
    }
    for (int t = 2; t + 2 < nt; t += 4) {
      tile_iter(integral_constant<int, 2>{}, no, no, no, trk_loop, t);
      tile_iter(integral_constant<int, 3>{}, no, no, no, trk_rest, t + 1);
      tile_iter(integral_constant<int, 0>{}, no, no, no, trk_rest, t + 2);
      tile_iter(integral_constant<int, 1>{}, no, no, no, trk_rest, t + 3);
    }
    tile_iter(integral_constant<int, 2>{}, yes, no, no, trk_rest, nt - 2);
    tile_iter(integral_constant<int, 3>{}, yes, yes, no, trk_rest, nt - 1);

    FLASH_STAMP(stamp_tail);

    // ---- P V of the last block, epilogue ------------------------------------------------------------------
    {
      drain_o(o);
      constexpr uint32_t vb = (uint32_t)(3 * kTileBytes + kBlkBytes);
      // all V^T fragments of the block first (the score registers are free by now): one LDS round trip
      // instead of one per fragment
      f16x8 vf[kDv16];
#pragma unroll
      for (int b = 0; b < kDv16; ++b) vf[b] = read_v<D>(vb, la, b);
#pragma unroll
      for (int b = 0; b < kDv16; ++b) {
#pragma unroll
        for (int q = 0; q < kQ16; ++q) mfma_o<true, EK>(o[q][b], vf[b], pB[q]);
      }
      drain_o(o);
      drain_lsum<D>(lsum);
    }
    // fp32 K and V: the next item's Q goes out now, into the registers the last product has just left; its
    // loads have the output scaling and stores to land
    if constexpr (KV32) {
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
        fetch_q_direct(next_item);
      } else {
        // (a definition on this path too, without an instruction: otherwise the landing registers' last
        // values stay alive around the whole item loop)
#pragma unroll
        for (int q = 0; q < kQ16; ++q) {
#pragma unroll
          for (int ks = 0; ks < kKS; ++ks) {
            if (kLandInAcc(q)) {
              asm volatile("" : "=a"(q_pair[q][ks][0]), "=a"(q_pair[q][ks][1]));
            } else {
              asm volatile("" : "=v"(q_pair[q][ks][0]), "=v"(q_pair[q][ks][1]));
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t worst = 0u;  // largest row sum of the wave as an integer: infinities and NaNs compare above every number
#pragma unroll
    for (int q = 0; q < kQ16; ++q) {
      if constexpr (QLAND) {
        if (q % kQPerPhase == 0) {
          // (the Q registers have been free since the item's last Q K^T product)
          if (q > 0 && has_next) convert_q32(q / kQPerPhase - 1);
          if (has_next) fetch_q32(next_item, q / kQPerPhase);
        }
      }
      const float l = quad_lane_sum(lsum[q][0], lane);
      if constexpr (SPARSE) {
        const uint32_t lbits = __float_as_uint(l) & 0x7fffffffu;
        worst = worst > lbits ? worst : lbits;
      }
      const float inv = __builtin_amdgcn_rcpf(l);  // v_rcp_f32, 1 ulp
      const uint32_t qrow = qt * (uint32_t)kBr + (uint32_t)(wave * kRowsPerWave + q * 16 + n16);
      store_o_block<kDv16, TO, kOutAux>(o[q], inv, qrow, d, g, o_rs);
      // (one query block at a time while the next item's Q is landing in vector registers: left alone the
      // compiler reads all of O^T out of the accumulation registers first)
      if constexpr (KV32) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SPARSE) {
      // A probability beyond the fp16 range (a score more than 16 exp2 units above a reference that most
      // steps no longer update) shows as an infinite row sum: the flag makes the workgroup start again at this item.
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(worst >= 0x7f800000u) != 0, 0)) {
        if (lane == 0) __hip_atomic_store(&overflowed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#if defined(FLASH_PHASE_STAMPS)
    if (!has_next) {
      acc_pro += stamp_loop - stamp_item;
      acc_loop += stamp_tail - stamp_loop;
      acc_tail += __builtin_amdgcn_s_memtime() - stamp_tail;
      FLASH_STAMP_STORE(p, wave, lane, 0ull, acc_pro, acc_pro + acc_loop, acc_pro + acc_loop + acc_tail);
      if (p.stamps && lane == 0) {
        unsigned long long* d_ = p.stamps + ((size_t)blockIdx.x * kWaves + wave) * 8;
        d_[5] = stamp_warm - stamp_entry;
        d_[6] = n_items + 1;
      }
    }
#endif
    if (!has_next) break;
    // the Q loads are older than this item's output stores: wait for everything but those stores
    if constexpr (KV32) {
      q_arrived(std::integral_constant<int, kQ16 * OutStores<TO>::per_block(kDv16)>{});
    } else if constexpr (QF32) {
      convert_q32(kQPhases - 1);
    } else {
      wait_q<kQ16 * OutStores<TO>::per_block(kDv16)>(qf);
    }
  #if defined(FLASH_PHASE_STAMPS)
    acc_pro += stamp_loop - stamp_item;
    acc_loop += stamp_tail - stamp_loop;
    acc_tail += __builtin_amdgcn_s_memtime() - stamp_tail;
    ++n_items;
#endif
    item = next_item;
    k_srd = kv_srd(p.k, next_head);
    v_srd = kv_srd(p.v, next_head);
  }
  sweep_end_item = item;
  };  // sweep

  if constexpr (kTrackMode == 0) {
    sweep(std::false_type{}, first_item);
  } else {
    sweep(std::true_type{}, first_item);
    // ---- second sweep: from the item whose first pass overflowed to the end of the list, from a cold start,
    // with the running maximum taken in every step (no input makes that pass overflow).  Inputs that overflow
    // in one item tend to do so in the next: the rest of the list is not tried the cheap way again, and the
    // price of such an input is one item (and two K/V tiles) per workgroup computed twice.  The item's output
    // rows are written again by the waves that wrote them first, after those stores have completed; matrix
    // instructions of a sweep that gave up inside an item have drained behind the wait states.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    const bool marked =
        __builtin_amdgcn_readfirstlane(__hip_atomic_load(&overflowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u;
    // a sweep that gave up was one item past the one that overflowed; one that reached the end of its list
    // has the mark of its last item still to read
    if (__builtin_expect(kKnock == 0 && (sweep_gave_up || marked), 0)) {
      sweep(std::false_type{}, sweep_gave_up ? sweep_end_item - wgs_on_xcd : sweep_end_item);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

#undef FLASH_LSUM_CASE
#undef FLASH_LSUM_SWITCH

}  // namespace FLASH_M16_NS
}  // namespace flash_hip
