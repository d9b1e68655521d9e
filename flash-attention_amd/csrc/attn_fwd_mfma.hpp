// FlashAttention-2 forward for gfx950 (MI355X, CDNA4): fp16 operands on
// v_mfma_f32_32x32x16_f16, fp32 accumulation, wave64.
//
// This is the general kernel of the library: it serves fp32 callers and head dims that are not a
// multiple of 8 halves (element-wise staging, FAST = false).  fp16 tensors with d % 8 == 0 run on
// the 16x16x32 kernels of attn_fwd_mfma16.hpp, which share this file's staging, descriptor and
// softmax helpers; building with -DFLASH_MFMA32 routes them here too (FAST = true) for comparison.
//
// Functional counterpart of the reference's kernel_mma_qreg family
// (reference flash/kernel_mma_qreg.cuh:16-267, kernel_mma_qreg_f32x4load.cuh,
// kernel_mma_qreg_async.cuh): one workgroup owns a tile of Q rows of one (batch, head), keeps Q in
// registers, streams K/V tiles through LDS and keeps the running row max / row sum in registers.
// Nothing of the reference's structure is kept beyond that:
//
//   * A workgroup is 4 waves, ONE PER SIMD, each with the whole 512-entry register file and 64
//     query rows (two 32-row query blocks): 256 rows per workgroup (Br).  K/V tiles are 64 rows
//     (Bc), processed as two 32-key blocks.  Every K/V fragment read from LDS feeds two MFMAs.
//   * Scores are produced TRANSPOSED: S^T = K * Q^T, A operand = K rows from LDS (ds_read_b128),
//     B operand = Q held in registers.  In the 32x32 accumulator layout the query index is then
//     the lane (lane & 31) and the key index runs over the 16 registers, so the row max / row sum
//     of the online softmax are in-lane reductions plus ONE v_permlane32_swap between the two
//     half-waves; no LDS round trip for S or P (the reference stores S to shared memory three
//     times per tile, kernel_mma_qreg.cuh:166,185,200).
//   * The output is accumulated TRANSPOSED as well: O^T = V^T * P^T.  P^T is exactly the S^T
//     accumulator converted to fp16 (8 consecutive registers = one B fragment with a permuted key
//     order), V^T fragments come from the row-major V tile with ds_read_b64_tr_b16 using the same
//     permuted key order.  The softmax rescale factor is a per-lane scalar for every register of
//     O^T.  O stays in registers for the whole K/V sweep and is divided by the row sum once
//     (the reference read-modify-writes O in HBM for every K/V tile, kernel_mma_qreg.cuh:207-250).
//   * Software pipeline over 32-key blocks: a step issues the MFMAs of QK(j+1) and PV(j-1) and, in
//     their shadow, the exponentials of block j and the row max of block j+1.  The rescale of O to a
//     new row max is deferred (threshold) and happens between steps, so a step has no branches.
//   * K/V tiles live in two 3-slot LDS rings filled by direct global->LDS loads
//     (buffer_load_dwordx4 ... lds) one whole tile iteration ahead; head dims that are not a
//     multiple of 8 halves and fp32 callers stage through registers instead.  The LDS image is
//     XOR-swizzled at 16-byte granularity so that both the K row reads and the V transposed reads
//     are bank-conflict free.
//   * Out-of-range rows/columns are handled by the buffer descriptor's range check (loads return 0,
//     stores are dropped); only the key-side tail needs an explicit -inf mask.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace flash_hip {

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4_b __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
#define FLASH_LDS_CAST(T, p) ((__attribute__((address_space(3))) T*)(p))

#ifndef FLASH_QB
#define FLASH_QB 2
#endif
constexpr int kQB = FLASH_QB;               // 32-row query blocks per wave
constexpr int kWaves = 8 / kQB;             // waves per workgroup: kQB = 2 -> one per SIMD, 1 -> two per SIMD
constexpr int kRowsPerWave = 32 * kQB;      // 64 query rows per wave
constexpr int kBr = kWaves * kRowsPerWave;  // 256 query rows per workgroup
constexpr int kBc = 64;                     // key/value rows per LDS tile
constexpr int kThreads = kWaves * 64;
constexpr uint32_t kOob = 0x40000000u;      // byte offset beyond any per-head buffer (< 1 GiB each)

enum Staging : int { kStageElem = 0, kStageVec = 1, kStageDma = 2 };

struct FwdParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  int N;              // sequence length
  int d;              // head dim (<= D)
  int nqt;            // query tiles per head = ceil(N / kBr)
  uint32_t total_wg;  // B * H * nqt
  // softmax_scale * log2(e) = q_scale * s_scale.  q_scale multiplies Q before the matrix products and
  // is a power of two, which is exact in 16-bit floating point; s_scale multiplies the fp32 scores just
  // before the exponential.  The 16x16x32 kernels keep the reference of a query in fp32 and get
  // q_scale = 1: Q reaches the matrix cores bit for bit as the caller stored it.  The 32x32x16 kernel
  // carries the reference through fp16 operands and gets the power of two that brings the scores into
  // exp2 range (s_scale in [1, 2)).  (One rounding of Q * softmax_scale * log2(e) to fp16 would save
  // the multiply per score, but it perturbs every score by 2^-11 relative and was measured to be the
  // largest error term of the kernel by far.)
  float q_scale;
  float s_scale;
  float defer_thr;    // kDeferLog2 / s_scale: the deferral threshold in units of the unscaled scores
#if defined(FLASH_PHASE_STAMPS)
  unsigned long long* stamps;  // diagnostic builds only: 8 values per wave (see tools/README.md)
#endif
};

// Diagnostic builds (-DFLASH_PHASE_STAMPS) record the shader clock at a few points of a wave's life;
// the values go to a buffer of their own and nothing is computed from them.
#if defined(FLASH_PHASE_STAMPS)
#define FLASH_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define FLASH_STAMP_STORE(prm, wave_id, lane_id, t0, t1, t2, t3)                                   \
  do {                                                                                              \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    const unsigned long long t4_ = __builtin_amdgcn_s_memtime();                                   \
    if ((prm).stamps && (lane_id) == 0) {                                                           \
      unsigned long long* d_ = (prm).stamps + ((size_t)blockIdx.x * kWaves + (wave_id)) * 8;       \
      d_[0] = t0; d_[1] = t1; d_[2] = t2; d_[3] = t3; d_[4] = t4_;                                  \
    }                                                                                               \
  } while (0)
#else
#define FLASH_STAMP(var)
#define FLASH_STAMP_STORE(prm, wave_id, lane_id, t0, t1, t2, t3)
#endif

template <int D>
struct Geom {
  static_assert(D == 64 || D == 128, "compiled head-dim tiles are 64 and 128");
  static constexpr int kRowBytes = D * 2;             // LDS row of fp16
  static constexpr int kChunks = D / 8;               // 16-byte chunks per row
  static constexpr int kKSteps = D / 16;              // MFMA k-steps of S^T = K Q^T
  static constexpr int kKHold = (D > 64) ? D / 32 : D / 16;  // K fragments held in registers at a time
  static constexpr int kDvBlocks = D / 32;            // 32-wide blocks of O^T rows
  static constexpr int kTileBytes = kBc * kRowBytes;  // one K or V tile
  static constexpr int kSlots = 3;                    // tiles per LDS ring (K ring + V ring)
  static constexpr int kLdsBytes = 2 * kSlots * kTileBytes;
  static constexpr int kPieces = kTileBytes / 1024;   // 1-KiB wave-instructions per tile
  static constexpr int kPiecesPerWave = kPieces / kWaves;
};

// 16-byte-chunk XOR swizzle of an LDS tile row.  Chosen so that
//  - 16 lanes reading the same logical chunk of 16 rows that are distinct mod 16 (K operand,
//    ds_read_b128) land on 16 different 16-byte slots of the 256-byte bank row, and
//  - 32 lanes reading 64 bytes of each of 4 consecutive rows (V operand, ds_read_b64_tr_b16)
//    cover the 256-byte bank row exactly once.
template <int D>
__device__ __forceinline__ int swz(int row) {
  if constexpr (D == 128) {
    return ((row & 3) << 2) | ((row >> 2) & 3);
  } else {
    return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  }
}

template <int D>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * Geom<D>::kRowBytes + ((chunk ^ swz<D>(row)) << 4);
}

// x of this lane's row as held by the lower (lo) and the upper (hi) half-wave.
struct HalfPair {
  float lo, hi;
};
__device__ __forceinline__ HalfPair both_halves(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// MFMAs are issued from inline asm so that the register FILE of each operand is fixed: with one wave
// per SIMD a lane owns 256 architectural VGPRs and 256 accumulation registers (AGPRs); O^T and the Q
// fragments are only ever touched by MFMAs and live in AGPRs for the whole kernel, everything the
// vector ALU works on (scores, probabilities, K/V fragments) lives in VGPRs.  Left to the compiler,
// the B operand is always taken from a VGPR and the AGPRs degrade to spill space (four
// v_accvgpr_read per MFMA).
// The price: the compiler does not know these statements are MFMAs and pads no hazards for them.
// The kernel keeps every consumer of an MFMA result at least two MFMAs behind its producer; where
// that is not structurally guaranteed (prologue, rescale path, epilogue) it calls mfma_drain_acc() / mfma_drain_s().
// A register copy that the compiler places directly in front of such a statement (it does so when it
// keeps an operand in the other register file) would be read too early: v_accvgpr_write -> MFMA
// needs two wait states.  PAD = true prefixes them to every MFMA; the tuned kernels (PAD = false)
// keep all operands resident in the right file instead, which tests/test_kernel_asm.py verifies on
// the generated code.
//   S^T accumulate:  acc (VGPR) += A (VGPR: K fragment) * B (AGPR: Q fragment)
template <bool PAD>
__device__ __forceinline__ void mfma_s_first(f32x16& acc, const f16x8& a, const f16x8& b) {
  if constexpr (PAD) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
  } else {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
  }
}
template <bool PAD>
__device__ __forceinline__ void mfma_s(f32x16& acc, const f16x8& a, const f16x8& b) {
  if constexpr (PAD) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
  } else {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
  }
}
//   O^T accumulate:  acc (AGPR) += A (VGPR: V^T fragment) * B (VGPR: P^T fragment)
template <bool PAD>
__device__ __forceinline__ void mfma_o(f32x16& acc, const f16x8& a, const f16x8& b) {
  if constexpr (PAD) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  } else {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  }
}
// Wait states that cover the result latency of the last MFMA before other instructions read it.
// The registers whose readers must stay behind the wait are operands of the statement: a bare
// statement orders nothing, the compiler is free to move register reads across it.
template <int NQ, int NB>
__device__ __forceinline__ void mfma_drain_acc(f32x16 (&o)[NQ][NB]) {
  static_assert((NQ == 1 || NQ == 2) && (NB == 2 || NB == 4), "operand list is written out for these shapes");
  if constexpr (NQ == 1 && NB == 4) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]));
  } else if constexpr (NQ == 1) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]));
  } else if constexpr (NB == 4) {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]),
                   "+a"(o[1][2]), "+a"(o[1][3]));
  } else {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));
  }
}
__device__ __forceinline__ void mfma_drain_s(f32x16 (&s)[2]) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s[0]), "+v"(s[1]));
}
__device__ __forceinline__ void mfma_drain_s(f32x16 (&s)[1]) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s[0])); }

// Raw buffer descriptor (stride 0, range-checked against `bytes`) as four wave-uniform dwords.
typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ srd_t make_srd(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

// The memory instructions below are issued from inline asm, so the compiler provides none of the wait
// states their scalar operands need: a vector memory instruction must not read a scalar register within
// five wait states of a VECTOR instruction writing it (v_readfirstlane: make_srd, wave-uniform offsets the
// compiler keeps in vector registers when scalar registers run short).  Call this on such operands where
// they may be fresh: it holds the values in scalar registers at this point of the program (the compiler
// cannot sink their computation below it) and spends the wait states.  tests/test_kernel_asm.py checks the
// generated code for the hazard.
__device__ __forceinline__ void settle_scalars(srd_t& a) { asm volatile("s_nop 4" : "+s"(a)); }
__device__ __forceinline__ void settle_scalars(srd_t& a, srd_t& b) { asm volatile("s_nop 4" : "+s"(a), "+s"(b)); }
__device__ __forceinline__ void settle_scalars(srd_t& a, uint32_t (&o)[2]) { asm volatile("s_nop 4" : "+s"(a), "+s"(o[0]), "+s"(o[1])); }
__device__ __forceinline__ void settle_scalars(srd_t& a, uint32_t (&o)[4]) {
  asm volatile("s_nop 4" : "+s"(a), "+s"(o[0]), "+s"(o[1]), "+s"(o[2]), "+s"(o[3]));
}
__device__ __forceinline__ void settle_scalars(srd_t& a, uint32_t (&o)[8]) {
  asm volatile("s_nop 4" : "+s"(a), "+s"(o[0]), "+s"(o[1]), "+s"(o[2]), "+s"(o[3]), "+s"(o[4]), "+s"(o[5]), "+s"(o[6]), "+s"(o[7]));
}
__device__ __forceinline__ void settle_scalars(srd_t& a, srd_t& b, uint32_t& c, uint32_t& d) {
  asm volatile("s_nop 4" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
}

// One direct global->LDS piece: every lane moves 16 bytes from byte offset `voff` of the buffer to
// LDS address lds_base + 16*lane (1 KiB per wave-instruction).  Issued from inline asm on purpose:
// the compiler does not see the load, so it neither waits for it before unrelated ds_reads nor
// drains it at the next barrier; the kernel waits with its own s_waitcnt vmcnt(0) before the
// barrier that publishes the tile.  M0 (the LDS base of the transfer) is written and consumed
// inside the one statement.
__device__ __forceinline__ void dma_piece(srd_t srd, uint32_t voff, uint32_t lds_base) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds"
      :
      : "s"(lds_base), "v"(voff), "s"(srd)
      : "memory");
}

// Same, with the LDS address given as a wave-uniform base register plus a compile-time offset: one
// scalar register serves every piece of every ring slot.  // `soff` is a wave-uniform byte offset added to the address after the range check (which sees `voff`
// only): the caller must know that voff + soff stays inside the buffer whenever voff does.
__device__ __forceinline__ void dma_piece_at(srd_t srd, uint32_t voff, uint32_t soff, uint32_t lds_wave_base,
                                             uint32_t lds_const) {
  asm volatile(
      "s_add_u32 m0, %0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds"
      :
      : "s"(lds_wave_base), "i"(lds_const), "v"(voff), "s"(srd), "s"(soff)
      : "memory", "scc");
}

// ---- element access through a raw buffer descriptor (range-checked) --------------------------
template <typename T>
__device__ __forceinline__ f16 load_elem(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off);
template <>
__device__ __forceinline__ f16 load_elem<f16>(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
  unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rs, byte_off, 0, 0);
  return __builtin_bit_cast(f16, u);
}
template <>
__device__ __forceinline__ f16 load_elem<float>(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
  uint32_t u = __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0);
  return (f16)__uint_as_float(u);
}
__device__ __forceinline__ void store_elem(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, f16 x) {
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, x), rs, byte_off, 0, 0);
}
__device__ __forceinline__ void store_elem(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, float x) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, byte_off, 0, 0);
}

// ---- building blocks of the software pipeline -------------------------------------------------------
// The unit of work is a BLOCK of 32 keys (half an LDS tile), for each of the wave's query blocks:
//   QK       S^T(32 keys x 32 queries) = K_blk * Q^T            D/16 MFMAs, A operand by ds_read_b128
//   softmax  running max, P^T = exp2(S^T*c - m*c) as fp16 B fragments, row sum      (VALU only)
//   PV       O^T += V_blk^T * P^T                                D/16 MFMAs, A operand by ds_read_b64_tr_b16

// Per-lane LDS byte addresses of the fragment reads, for slot 0 / key row 0 of each ring.  Slots,
// blocks and k-steps add multiples of 16 rows, which the swizzle does not see, so they are
// compile-time constants that end up in the 16-bit offset field of the ds_read (this is why the
// addresses are absolute and opaque to the compiler: a constant that it could fold with the ring
// base would exceed that field and cost one address register per slot and block).
template <int D>
struct LaneAddr {
  uint32_t k[Geom<D>::kKSteps];       // K[r][16*ks + 8*h ..]: A fragment of k-step ks of S^T = K Q^T
  uint32_t v[2][Geom<D>::kDvBlocks];  // V rows 8*jj + 4*h + (0..3), dv block b: half jj of an A fragment of O^T
};

template <int D>
__device__ __forceinline__ LaneAddr<D> make_lane_addr(int lane, lds_char* k_ring, lds_char* v_ring) {
  using G = Geom<D>;
  LaneAddr<D> la;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int i16 = lane & 15;
  const int tq = i16 >> 2;         // row of the 4x16 transposed block this lane addresses
  const int tp = i16 & 3;          // 4-column group of that row
  const int tg = (lane >> 4) & 1;  // which 16 of the 32 dv columns
  const uint32_t kbase = (uint32_t)(uintptr_t)k_ring;
  const uint32_t vbase = (uint32_t)(uintptr_t)v_ring;
#pragma unroll
  for (int ks = 0; ks < G::kKSteps; ++ks) {
    uint32_t a = kbase + (uint32_t)lds_off<D>(r, 2 * ks + h);
    asm volatile("" : "+v"(a));
    la.k[ks] = a;
  }
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
    for (int b = 0; b < G::kDvBlocks; ++b) {
      uint32_t a = vbase + (uint32_t)(lds_off<D>(8 * jj + 4 * h + tq, b * 4 + tg * 2 + (tp >> 1)) + 8 * (tp & 1));
      asm volatile("" : "+v"(a));
      la.v[jj][b] = a;
    }
  }
  return la;
}

// A fragment of O^T = V^T P^T for dv block b, k-step ks of the 32-key block that starts `voff` bytes
// into the V ring: element j of half-wave h is key 16*ks + 8*(j>>2) + 4*h + (j&3), the order the
// registers of P^T already have.
template <int D>
__device__ __forceinline__ f16x8 read_v_frag(uint32_t voff, const LaneAddr<D>& la, int b, int ks) {
  f16x8 a;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const uint32_t addr = la.v[jj][b] + (voff + (uint32_t)(16 * ks * Geom<D>::kRowBytes));
    const fp16x4_b t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(FLASH_LDS_CAST(fp16x4_b, (uintptr_t)addr));
    const f16x4 tv = __builtin_bit_cast(f16x4, t);
    a[4 * jj + 0] = tv[0];
    a[4 * jj + 1] = tv[1];
    a[4 * jj + 2] = tv[2];
    a[4 * jj + 3] = tv[3];
  }
  return a;
}

// A fragment of S^T = K Q^T, k-step ks of the 32-key block that starts `koff` bytes into the K ring.
template <int D>
__device__ __forceinline__ f16x8 read_k_frag(uint32_t koff, const LaneAddr<D>& la, int ks) {
  return *(FLASH_LDS_CAST(f16x8, (uintptr_t)(la.k[ks] + koff)));
}

// Running softmax state of one lane for one of its query blocks: the query is lane & 31 of that
// block, the keys are those of the lane's half-wave.  Scores leave the MFMAs multiplied by the
// power-of-two part of softmax_scale * log2(e) (Q is pre-scaled, exactly) and RELATIVE to the
// reference `mc` (a rank-1 MFMA k-step adds -mc to every key of the query); the exponential takes the
// accumulator times the remaining factor s_scale.  mc, mx_next and the deferral threshold are in
// accumulator units (exp2 units / s_scale).
struct SoftmaxState {
  float mc;       // reference that S, P, l and O are relative to
  float l;        // this lane's partial row sum at that scale
  float alpha;    // factor still to be applied to O (valid while the wave's `pending` flag is set)
  float mx_next;  // max of the next block's relative scores over this lane's keys
};

// Softmax scale factors as the pipeline steps use them (FwdParams::s_scale, ::defer_thr).
struct ScaleSplit {
  float c;    // residual factor applied to the fp32 scores before exp2
  float thr;  // deferral threshold in accumulator units
};

// How far (in exp2 units) a score may exceed the reference before the state is moved to a new
// reference.  P then stays below 2^kDeferLog2 = 256, far inside fp16 range, and O / l are fp32, so
// the result is the same up to rounding; it saves the O-wide multiply on most blocks.
constexpr float kDeferLog2 = 8.0f;

// The reference is carried into the MFMA as two fp16 terms (hi + lo ~ 22 significant bits); the
// value every other computation uses is the exact fp32 sum of the two.
struct SplitRef {
  f16 hi, lo;
  float eff;
};
__device__ __forceinline__ SplitRef split_ref(float mc) {
  SplitRef r;
  r.hi = (f16)mc;
  r.lo = (f16)(mc - (float)r.hi);
  r.eff = (float)r.hi + (float)r.lo;
  return r;
}
// B fragment of the rank-1 k-step: row k=0 holds -hi, row k=1 holds -lo of the lane's query (only
// the lower half-wave owns k = 0..7), everything else is zero.  The matching A fragment is `ones`.
__device__ __forceinline__ f16x8 bias_fragment(const SplitRef& r, int lane) {
  f16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (f16)0.f;
  if (lane < 32) {
    b[0] = -r.hi;
    b[1] = -r.lo;
  }
  return b;
}

template <int D>
__device__ __forceinline__ void rescale_o(f32x16 (&o)[Geom<D>::kDvBlocks], float alpha) {
#pragma unroll
  for (int b = 0; b < Geom<D>::kDvBlocks; ++b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) o[b][i] *= alpha;
  }
}

// Keys >= N get a score of -inf.  Register i of an S^T block holds key kv0 + (i&3) + 8*(i>>2) + 4*h.
__device__ __forceinline__ void mask_scores(f32x16& s, int kv0, int N, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int kv = kv0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (kv >= N) s[i] = -INFINITY;
  }
}

// Row max of an S^T block over both half-waves.
__device__ __forceinline__ float block_row_max(const f32x16& s) {
  float mx = fmaxf(s[0], s[1]);
#pragma unroll
  for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(s[i], s[i + 1]));
  const HalfPair mp = both_halves(mx);
  return fmaxf(mp.lo, mp.hi);
}

// One pipeline step.  With one wave per SIMD the step is bound by instruction ISSUE, not by the
// matrix pipe: every non-MFMA instruction costs 4-9 issue cycles (v_exp_f32 9, v_cvt_pk_f16_f32 8,
// an LDS read 7.5, simple VALU 5) against 32 cycles of matrix pipe per MFMA, so the step carries the
// minimum of them: per score one v_exp_f32, half a v_cvt_pk_f16_f32, a quarter each of v_pk_add_f16
// and v_dot2c_f32_f16 (row sum of the rounded probabilities, fp32 accumulation) and half a
// v_max3_f32.  The instruction order below is the intended issue order: the step is cut into SLOTS of
// one MFMA each, and a scheduling fence after each slot keeps the compiler from regrouping them
// (left alone it hoists every LDS read to the top of the step and spills).
//   QK slots ((KS+1) * kQB)  QK(j+1): s_nxt[q] = ones * bias[q] (rank-1: -reference), then
//                            += K_{j+1}[ks] * Q[q][ks]; the K fragment register just consumed by both
//                            query blocks is refilled (rest of this block, then the first fragments
//                            of block j+2)
//   PV slots (2*NB * kQB)    PV(j-1): O^T[q][b] += V_{j-1}^T[b] * P_{j-1}[q]^T, V fragments two dv
//                            blocks ahead; in-lane max of s_nxt once the QK chain has drained
//   all slots                exp(j): a share of p_cur = exp2(s_cur) and of the row sum
template <int D, bool PAD, bool HAS_QK, bool HAS_PRELOAD, bool MASKED, typename Hook>
__device__ __forceinline__ void pipeline_step(const Hook& hook, uint32_t kb, uint32_t vb, uint32_t kb_next,
                                              const LaneAddr<D>& la, const f16x8& ones,
                                              const f16x8 (&bq)[kQB], const f16x8 (&qf)[kQB][Geom<D>::kKSteps],
                                              f16x8 (&ka)[Geom<D>::kKHold], f32x16 (&o)[kQB][Geom<D>::kDvBlocks],
                                              const f32x16 (&s_cur)[kQB], f32x16 (&s_nxt)[kQB],
                                              const f16x8 (&p_prev)[kQB][2], f16x8 (&p_cur)[kQB][2],
                                              SoftmaxState (&st)[kQB], const ScaleSplit sc, int kv_next0, int N,
                                              int lane) {
  using G = Geom<D>;
  constexpr int KS = G::kKSteps;
  constexpr int KH = G::kKHold;
  constexpr int NB = G::kDvBlocks;
  constexpr int QK_SLOTS = (KS + 1) * kQB;
  constexpr int PV_SLOTS = 2 * NB * kQB;
  constexpr int SLOTS = QK_SLOTS + PV_SLOTS;
  constexpr int ELEMS = 16 * kQB;        // scores per lane and step
  constexpr int MAX_OPS = 8 * kQB;       // in-lane max micro-ops: 8 per query block (16 registers)
  constexpr int MAX_DELAY = 2;           // PV slots to wait for the QK chain to drain

  f16x8 va[4];
  float mx[kQB];
#pragma unroll
  for (int q = 0; q < kQB; ++q) mx[q] = 0.f;

  // exp(j) share, software-pipelined over the slots so that no instruction consumes a result
  // produced right before it (a lone wave has nothing else to issue while it waits).  After the
  // exponential of element g:
  //   g odd        pack (g-1, g) to fp16: one register of the P^T fragment
  //   g % 4 == 0   fp16 sum of the two pairs of the group of four before (2-term sums: the error
  //                stays at fp16 rounding of the result, a longer fp16 chain would drop small terms)
  //   g % 4 == 2   that group's two sums into the fp32 row sum (v_dot2c_f32_f16 with ones)
  float t_exp = 0.f, t_exp_prev = 0.f;
  f16x2 pk[2], ps, one2;
  one2[0] = one2[1] = (f16)1.f;
  // keeps the instruction that produced x in the slot it was written in
  auto pin = [](f16x2& x) __attribute__((always_inline)) {
    uint32_t w = __builtin_bit_cast(uint32_t, x);
    asm volatile("" : "+v"(w));
    x = __builtin_bit_cast(f16x2, w);
  };
  auto group_sum = [&]() __attribute__((always_inline)) {
    ps = pk[0] + pk[1];
    pin(ps);
  };
  auto group_add = [&](int group) __attribute__((always_inline)) {
    float& l = st[group / 4].l;
    l = __builtin_amdgcn_fdot2(ps, one2, l, false);
    asm volatile("" : "+v"(l));
  };
  auto finish = [&](int g) __attribute__((always_inline)) {  // element g has its exponential
    const int q = g / 16, i = g % 16;
    if ((g & 3) == 0 && g >= 4) group_sum();
    if ((g & 3) == 2 && g >= 6) group_add(g / 4 - 1);
    if ((g & 1) != 0) {
      f16x2 w;
      w[0] = (f16)t_exp_prev;
      w[1] = (f16)t_exp;
      pin(w);  // otherwise the fp32 exponentials stay alive and are converted several slots later
      p_cur[q][i >> 3][(i & 7) - 1] = w[0];
      p_cur[q][i >> 3][i & 7] = w[1];
      pk[(g >> 1) & 1] = w;
    }
    t_exp_prev = t_exp;
  };
  auto exp_share = [&](int n) __attribute__((always_inline)) {
#pragma unroll
    for (int g = n * ELEMS / SLOTS; g < (n + 1) * ELEMS / SLOTS; ++g) {
      if (g > 0) finish(g - 1);
      t_exp = fast_exp2(s_cur[g / 16][g % 16] * sc.c);
      asm volatile("" : "+v"(t_exp));
    }
  };
  // in-lane max micro-op u of s_nxt: query block u / 8, op u % 8 folds two more registers
  auto max_op = [&](int u) __attribute__((always_inline)) {
    const int q = u / 8, k = u % 8;
    if (k == 0) {
      if constexpr (MASKED) mask_scores(s_nxt[q], kv_next0, N, lane);
      mx[q] = fmaxf(fmaxf(s_nxt[q][0], s_nxt[q][1]), s_nxt[q][2]);
    } else if (k < 7) {
      mx[q] = fmaxf(fmaxf(mx[q], s_nxt[q][2 * k + 1]), s_nxt[q][2 * k + 2]);
    } else {
      mx[q] = fmaxf(mx[q], s_nxt[q][15]);
      st[q].mx_next = mx[q];
    }
    asm volatile("" : "+v"(mx[q]));  // keep the op in this slot
  };

  // ---- QK slots ------------------------------------------------------------------------------------
#pragma unroll
  for (int ks = -1; ks < KS; ++ks) {
#pragma unroll
    for (int q = 0; q < kQB; ++q) {
      const int n = (ks + 1) * kQB + q;
      if constexpr (HAS_QK) {
        if (ks < 0) {
          mfma_s_first<PAD>(s_nxt[q], ones, bq[q]);
        } else {
          mfma_s<PAD>(s_nxt[q], ka[ks % KH], qf[q][ks]);
        }
        __builtin_amdgcn_sched_barrier(0);  // the MFMA opens its slot, the fillers follow it
        if (ks >= 0 && q == kQB - 1) {
          if (ks + KH < KS) {
            ka[ks % KH] = read_k_frag<D>(kb, la, ks + KH);
          } else if constexpr (HAS_PRELOAD) {
            ka[ks % KH] = read_k_frag<D>(kb_next, la, ks + KH - KS);
          }
        }
      }
      // V fragments of the first two dv blocks, a few slots before the PV slots begin
      if (n == (QK_SLOTS >= 6 ? QK_SLOTS - 6 : 0)) {
        va[0] = read_v_frag<D>(vb, la, 0, 0);
        va[1] = read_v_frag<D>(vb, la, 0, 1);
      }
      if (n == (QK_SLOTS >= 6 ? QK_SLOTS - 4 : 1)) {
        va[2] = read_v_frag<D>(vb, la, 1, 0);
        va[3] = read_v_frag<D>(vb, la, 1, 1);
      }
      exp_share(n);
      hook(n);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- PV slots ------------------------------------------------------------------------------------
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int q = 0; q < kQB; ++q) {
        const int pslot = (b * 2 + ks) * kQB + q;
        const int vi = (b & 1) * 2 + ks;
        mfma_o<PAD>(o[q][b], va[vi], p_prev[q][ks]);
        __builtin_amdgcn_sched_barrier(0);
        if (q == kQB - 1 && b + 2 < NB) va[vi] = read_v_frag<D>(vb, la, b + 2, ks);
        exp_share(QK_SLOTS + pslot);
        hook(QK_SLOTS + pslot);
        if constexpr (HAS_QK) {
#pragma unroll
          for (int u = 0; u < MAX_OPS; ++u) {
            if (MAX_DELAY + u * (PV_SLOTS - MAX_DELAY) / MAX_OPS == pslot) max_op(u);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  finish(ELEMS - 1);
  group_sum();
  group_add(ELEMS / 4 - 1);
}

// Between two steps: apply a pending O rescale (the P V product it had to wait for has been issued),
// then decide whether the block whose in-lane maxima are st.mx_next forces a new reference.  Both are
// rare, wave-uniform events; the common path is one vector compare and one scalar branch.
// Moving the reference by d: the scores of the next block, already computed against the old one,
// drop by d; l and (one step later, once the P V product of the current block is in) O shrink by
// 2^-d; the bias fragment of the rank-1 k-step is rebuilt.
template <int D, bool HAS_NEXT>
__device__ __forceinline__ void step_end(f32x16 (&o)[kQB][Geom<D>::kDvBlocks], SoftmaxState (&st)[kQB],
                                         f16x8 (&bq)[kQB], f32x16 (&s_nxt)[kQB], bool& pending, const ScaleSplit sc,
                                         int lane) {
  bool any_need = false;
  if constexpr (HAS_NEXT) {
    float mxx = st[0].mx_next;
#pragma unroll
    for (int q = 1; q < kQB; ++q) mxx = fmaxf(mxx, st[q].mx_next);
    any_need = __builtin_amdgcn_ballot_w64(mxx > sc.thr) != 0;
  }
  if (__builtin_expect(pending || any_need, 0)) {  // rare: keep it out of the fall-through path
    if (pending) {
      mfma_drain_acc(o);  // the last P V MFMAs of the step must have written O
#pragma unroll
      for (int q = 0; q < kQB; ++q) rescale_o<D>(o[q], st[q].alpha);
    }
    pending = false;
    if (any_need) {
#pragma unroll
      for (int q = 0; q < kQB; ++q) {
        const HalfPair mp = both_halves(st[q].mx_next);
        const float rise = fmaxf(fmaxf(mp.lo, mp.hi), 0.f);  // references only move up
        const SplitRef r = split_ref(st[q].mc + rise);
        const float d = r.eff - st[q].mc;
        st[q].alpha = fast_exp2(-d * sc.c);  // 1 for rows whose reference does not move
        st[q].mc = r.eff;
        st[q].l *= st[q].alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) s_nxt[q][i] -= d;
        bq[q] = bias_fragment(r, lane);
      }
      // the rebuilt fragments are MFMA operands of the very next instruction group
      if constexpr (kQB == 2) {
        asm volatile("s_nop 4" : "+a"(bq[0]), "+a"(bq[kQB - 1]));
      } else {
        asm volatile("s_nop 4" : "+a"(bq[0]));
      }
      pending = true;
    }
  }
}

// ---- the kernel ---------------------------------------------------------------------------------
// D: compiled head-dim tile (d <= D, missing columns are zero-filled).
// T: element type of Q/K/V/O in HBM (f16 or float; float is rounded to f16 while staging).
// FAST: T == f16 and d % 8 == 0: 128-bit Q loads, direct global->LDS K/V staging, 64-bit O stores.
//       Otherwise everything goes through range-checked element loads/stores.
//
// LDS: a ring of three K tiles followed by a ring of three V tiles (64 rows = two blocks each).
// Tile iteration t runs
//   step 2t   : QK(2t+1) [K tile t, rows 32..63]  PV(2t-1) [V tile t-1, rows 32..63]  exp(2t)   max(2t+1)
//   step 2t+1 : QK(2t+2) [K tile t+1, rows 0..31] PV(2t)   [V tile t, rows 0..31]     exp(2t+1) max(2t+2)
// and touches K tiles t, t+1 and V tiles t-1, t.  At its start it stages K tile t+2 and V tile t+1
// into the slots of K tile t-1 and V tile t-2; at its end every wave waits for its own staging
// traffic and all waves meet at the one barrier of the iteration.  A staged tile therefore has a
// whole iteration to land, and nothing that a barrier publishes is needed right after that barrier.
template <int D, typename T, bool FAST>
__global__ __launch_bounds__(kThreads, 1) void attn_fwd_kernel(const FwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  using G = Geom<D>;
  static_assert(!FAST || sizeof(T) == 2, "direct-to-LDS staging needs fp16 data in HBM");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_char* const smem = (lds_char*)smem_raw;

  // Work item: consecutive items = query tiles of one head; items are dealt to the 8 XCDs in
  // contiguous ranges (workgroup ids that are equal mod 8 share an XCD and hence an L2), so the
  // workgroups that stream the same K/V run on the same L2 at about the same time.
  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot_id = bid >> 3;
  const uint32_t per = p.total_wg >> 3, rem = p.total_wg & 7u;
  const uint32_t item = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot_id;
  const uint32_t head = item / (uint32_t)p.nqt;
  const uint32_t qt = item - head * (uint32_t)p.nqt;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;
  const int N = p.N, d = p.d;

  const size_t head_elems = (size_t)N * (size_t)d;
  const uint32_t head_bytes = (uint32_t)(head_elems * sizeof(T));
  const T* qh = (const T*)p.q + (size_t)head * head_elems;
  const T* kh = (const T*)p.k + (size_t)head * head_elems;
  const T* vh = (const T*)p.v + (size_t)head * head_elems;
  T* oh = (T*)p.o + (size_t)head * head_elems;
  const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)qh, 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kh, 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vh, 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)oh, 0, head_bytes, 0x00020000);
  const srd_t k_srd = make_srd(kh, head_bytes);
  const srd_t v_srd = make_srd(vh, head_bytes);

  const uint32_t row_bytes = (uint32_t)d * sizeof(T);
  const uint32_t tile_step = (uint32_t)kBc * row_bytes;  // HBM bytes between consecutive K/V tiles

  lds_char* const k_ring = smem;
  lds_char* const v_ring = smem + G::kSlots * G::kTileBytes;

  // ---- K/V staging --------------------------------------------------------------------------
  // FAST: each wave issues kPiecesPerWave 1-KiB pieces per tile; lane i of piece pc fills LDS
  // chunk pc*64 + i, i.e. row (pc*64+i)/kChunks, slot (pc*64+i)%kChunks, with the source chunk that
  // the swizzle maps there.
  uint32_t dma_off[G::kPiecesPerWave];
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < G::kPiecesPerWave; ++i) {
      const int ci = (wave * G::kPiecesPerWave + i) * 64 + lane;
      const int row = ci / G::kChunks;
      const int src = (ci % G::kChunks) ^ swz<D>(row);
      dma_off[i] = (src * 8 < d) ? (uint32_t)row * row_bytes + (uint32_t)src * 16u : kOob;
    }
  }
  auto stage = [&](const __amdgpu_buffer_rsrc_t rs, const srd_t srd, int t, lds_char* dst) __attribute__((always_inline)) {
    const uint32_t toff = (uint32_t)t * tile_step;
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < G::kPiecesPerWave; ++i) {
        const int pc = wave * G::kPiecesPerWave + i;
        const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(dst + pc * 1024));
        dma_piece(srd, dma_off[i] + toff, lds_base);
      }
    } else {
      // two adjacent columns per step, converted to fp16 and written as one dword
      constexpr int kPairs = kBc * D / 2;
#pragma unroll 4
      for (int idx = tid; idx < kPairs; idx += kThreads) {
        const int row = idx / (D / 2);
        const int cp = idx % (D / 2);
        const int col = 2 * cp;
        const uint32_t base = toff + (uint32_t)row * row_bytes;
        const uint32_t o0 = (col < d) ? base + (uint32_t)col * sizeof(T) : kOob;
        const uint32_t o1 = (col + 1 < d) ? base + (uint32_t)(col + 1) * sizeof(T) : kOob;
        f16x2 x;
        x[0] = load_elem<T>(rs, o0);
        x[1] = load_elem<T>(rs, o1);
        *FLASH_LDS_CAST(f16x2, dst + lds_off<D>(row, cp >> 2) + (cp & 3) * 4) = x;
      }
    }
  };
  auto stage_k = [&](int t, int slot) __attribute__((always_inline)) { stage(k_rs, k_srd, t, k_ring + slot * G::kTileBytes); };
  auto stage_v = [&](int t, int slot) __attribute__((always_inline)) { stage(v_rs, v_srd, t, v_ring + slot * G::kTileBytes); };
  auto staged_barrier = [&]() __attribute__((always_inline)) {
    // every wave waits for its own staging traffic, then all waves meet: tiles issued before this
    // point are readable by everyone, tiles read before this point may be overwritten
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nt = (N + kBc - 1) / kBc;
  constexpr int kBlk = 32 * G::kRowBytes;  // bytes of one 32-key block inside a tile
  stage_k(0, 0);
  stage_v(0, 0);
  if (nt > 1) stage_k(1, 1);
  // Step 0 multiplies P(-1) = 0 with the second block of "V tile -1" (slot 2): make that finite.
  for (int i = tid * 16; i < kBlk; i += kThreads * 16) {
    *FLASH_LDS_CAST(u32x4, v_ring + 2 * G::kTileBytes + kBlk + i) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- Q fragments: B operand of S^T = K Q^T: lane (r, h) holds Q[row r][16*ks + 8*h + 0..7] ----
  int qrow[kQB];
  f16x8 qf[kQB][G::kKSteps];
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
    qrow[q] = (int)qt * kBr + wave * kRowsPerWave + q * 32 + r;
#pragma unroll
    for (int ks = 0; ks < G::kKSteps; ++ks) {
      const int col = 16 * ks + 8 * h;
      if constexpr (FAST) {
        const uint32_t off = (col < d) ? ((uint32_t)qrow[q] * (uint32_t)d + (uint32_t)col) * 2u : kOob;
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(q_rs, off, 0, 0);
        qf[q][ks] = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[q][ks][j] = (f16)((float)qf[q][ks][j] * p.q_scale);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t off =
              (col + j < d) ? ((uint32_t)qrow[q] * (uint32_t)d + (uint32_t)(col + j)) * (uint32_t)sizeof(T) : kOob;
          qf[q][ks][j] = (f16)((float)load_elem<T>(q_rs, off) * p.q_scale);
        }
      }
    }
  }

  // Q and O^T live in accumulation registers from here on.
  f32x16 o[kQB][G::kDvBlocks];
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
#pragma unroll
    for (int ks = 0; ks < G::kKSteps; ++ks) asm volatile("" : "+a"(qf[q][ks]));
#pragma unroll
    for (int b = 0; b < G::kDvBlocks; ++b) {
#pragma unroll
      for (int i = 0; i < 16; ++i) o[q][b][i] = 0.f;
      asm volatile("" : "+a"(o[q][b]));
    }
  }
  const LaneAddr<D> la = make_lane_addr<D>(lane, k_ring, v_ring);

  staged_barrier();

  // ---- prologue: S(0), its row max, and the first K fragments of block 1 ------------------------
  f32x16 sA[kQB], sB[kQB];
  f16x8 pA[kQB][2], pB[kQB][2];
  f16x8 ka[G::kKHold];
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pB[q][0][j] = (f16)0.f;
      pB[q][1][j] = (f16)0.f;
    }
  }
#pragma unroll
  for (int ks = 0; ks < G::kKHold; ++ks) ka[ks] = read_k_frag<D>(0u, la, ks);
#pragma unroll
  for (int ks = 0; ks < G::kKSteps; ++ks) {
#pragma unroll
    for (int q = 0; q < kQB; ++q) {
      if (ks == 0) {
        mfma_s_first<true>(sA[q], ka[ks % G::kKHold], qf[q][ks]);
      } else {
        mfma_s<true>(sA[q], ka[ks % G::kKHold], qf[q][ks]);
      }
    }
    if (ks + G::kKHold < G::kKSteps) {
      ka[ks % G::kKHold] = read_k_frag<D>(0u, la, ks + G::kKHold);
    } else {
      ka[ks % G::kKHold] = read_k_frag<D>((uint32_t)kBlk, la, ks + G::kKHold - G::kKSteps);
    }
  }
  mfma_drain_s(sA);
  // The first reference of every query is the row max of block 0 (which always holds key 0, so it
  // is finite); S(0) was computed without a bias, make it relative now.
  SoftmaxState st[kQB];
  f16x8 bq[kQB];
  f16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (f16)((lane < 32 && j < 2) ? 1.f : 0.f);
  bool pending = false;
  const ScaleSplit sc{p.s_scale, p.defer_thr};
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
    mask_scores(sA[q], 0, N, lane);
    const SplitRef r = split_ref(block_row_max(sA[q]));
#pragma unroll
    for (int i = 0; i < 16; ++i) sA[q][i] -= r.eff;
    bq[q] = bias_fragment(r, lane);
    asm volatile("" : "+a"(bq[q]));
    st[q].mc = r.eff;
    st[q].l = 0.f;
    st[q].alpha = 1.f;
    st[q].mx_next = 0.f;
  }

  // One tile iteration.  SLOT >= 0: ring slot of tile t known at compile time (all LDS offsets
  // become instruction immediates); SLOT < 0: taken from `slot_rt`.
  auto tile_iter = [&](auto slot_tag, auto masked_tag, auto last_tag, int t, int slot_rt) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    const int s0 = SLOT >= 0 ? SLOT : slot_rt;
    const int s1 = s0 == 2 ? 0 : s0 + 1;  // slot of tile t+1
    const int s2 = s1 == 2 ? 0 : s1 + 1;  // slot of tile t+2 == slot of tile t-1
    auto no_hook = [](int) __attribute__((always_inline)) {};
    // staging spread over the slots of the first step: piece i of the wave at slot i * spacing + 1
    auto dma_hook = [&](int n) __attribute__((always_inline)) {
      if constexpr (FAST) {
        constexpr int kSlotsPerStep = (G::kKSteps + 1) * kQB + 2 * G::kDvBlocks * kQB;
        constexpr int kSpacing = kSlotsPerStep / (2 * G::kPiecesPerWave);
        if (n % kSpacing == 1) {
          const int i = n / kSpacing;
          if (i >= 2 * G::kPiecesPerWave) {
            // no piece left for this slot
          } else if (i < G::kPiecesPerWave) {
            if (t + 2 < nt) {
              const int pc = wave * G::kPiecesPerWave + i;
              const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(k_ring + s2 * G::kTileBytes + pc * 1024));
              dma_piece(k_srd, dma_off[i] + (uint32_t)(t + 2) * tile_step, lds_base);
            }
          } else {
            if (t + 1 < nt) {
              const int pc = wave * G::kPiecesPerWave + (i - G::kPiecesPerWave);
              const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(v_ring + s1 * G::kTileBytes + pc * 1024));
              dma_piece(v_srd, dma_off[i - G::kPiecesPerWave] + (uint32_t)(t + 1) * tile_step, lds_base);
            }
          }
        }
      } else {
        if (n == 0) {
          if (t + 2 < nt) stage_k(t + 2, s2);
          if (t + 1 < nt) stage_v(t + 1, s1);
        }
      }
    };
    const uint32_t k_cur = (uint32_t)(s0 * G::kTileBytes);  // byte offsets into the K / V ring
    const uint32_t k_nxt = (uint32_t)(s1 * G::kTileBytes);
    const uint32_t v_cur = (uint32_t)(s0 * G::kTileBytes);
    const uint32_t v_prv = (uint32_t)(s2 * G::kTileBytes);
    pipeline_step<D, !FAST, true, !LAST, MASKED>(dma_hook, k_cur + kBlk, v_prv + kBlk, k_nxt, la, ones, bq, qf, ka, o, sA,
                                                 sB, pB, pA, st, sc, t * kBc + 32, N, lane);
    step_end<D, true>(o, st, bq, sB, pending, sc, lane);
    pipeline_step<D, !FAST, !LAST, !LAST, MASKED>(no_hook, k_nxt, v_cur, k_nxt + kBlk, la, ones, bq, qf, ka, o, sB, sA, pA,
                                                  pB, st, sc, (t + 1) * kBc, N, lane);
    step_end<D, !LAST>(o, st, bq, sA, pending, sc, lane);
    staged_barrier();
  };
  using std::integral_constant;
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  int t = 0;
  // main loop: whole triples of tiles (ring slots 0, 1, 2) that end before the last two tiles
  for (; t + 5 <= nt; t += 3) {
    tile_iter(integral_constant<int, 0>{}, no, no, t, 0);
    tile_iter(integral_constant<int, 1>{}, no, no, t + 1, 1);
    tile_iter(integral_constant<int, 2>{}, no, no, t + 2, 2);
  }
  // remaining tiles (at most four, among them the possibly ragged last one): runtime slot, key mask
  // evaluated
  for (; t + 1 < nt; ++t) tile_iter(integral_constant<int, -1>{}, yes, no, t, t % 3);
  tile_iter(integral_constant<int, -1>{}, yes, yes, t, t % 3);
  // PV of the last block (its pending rescale, if any, was applied by the last step_end)
  {
    mfma_drain_acc(o);
    const uint32_t vb = (uint32_t)(((nt - 1) % 3) * G::kTileBytes + kBlk);
#pragma unroll
    for (int b = 0; b < G::kDvBlocks; ++b) {
      const f16x8 v0 = read_v_frag<D>(vb, la, b, 0);
      const f16x8 v1 = read_v_frag<D>(vb, la, b, 1);
#pragma unroll
      for (int q = 0; q < kQB; ++q) {
        mfma_o<true>(o[q][b], v0, pB[q][0]);
        mfma_o<true>(o[q][b], v1, pB[q][1]);
      }
    }
    mfma_drain_acc(o);
  }

  // ---- epilogue: O = O^T / l ----------------------------------------------------------------------
  // Registers 4g..4g+3 of block b are dv = 32b + 8g + 4h + 0..3 of the lane's query row.  The two
  // half-waves trade their 4-column groups (v_permlane32_swap) so that a lane owns 8 consecutive
  // columns and writes them with one 128-bit store: the lower half-wave takes columns 0..7 of each
  // 16-column group, the upper half-wave columns 8..15.
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
    const HalfPair lp = both_halves(st[q].l);
    const float inv = 1.0f / (lp.lo + lp.hi);
#pragma unroll
    for (int b = 0; b < G::kDvBlocks; ++b) {
      if constexpr (FAST) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          f16x4 x, y;  // groups 2gp and 2gp+1
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x[j] = (f16)(o[q][b][8 * gp + j] * inv);
            y[j] = (f16)(o[q][b][8 * gp + 4 + j] * inv);
          }
          const u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
          u32x4 out;
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            auto r = __builtin_amdgcn_permlane32_swap(xu[w], yu[w], false, false);
            out[w] = r[0];      // lower lanes: own group 2gp      upper lanes: partner's group 2gp+1
            out[2 + w] = r[1];  // lower lanes: partner's group 2gp upper lanes: own group 2gp+1
          }
          const int dv = 32 * b + 16 * gp + 8 * h;
          const uint32_t off = (dv < d) ? ((uint32_t)qrow[q] * (uint32_t)d + (uint32_t)dv) * 2u : kOob;
          __builtin_amdgcn_raw_buffer_store_b128(out, o_rs, off, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = 32 * b + 8 * g + 4 * h;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t off = (dv + j < d)
                                     ? ((uint32_t)qrow[q] * (uint32_t)d + (uint32_t)(dv + j)) * (uint32_t)sizeof(T)
                                     : kOob;
            store_elem(o_rs, off, (T)(o[q][b][4 * g + j] * inv));
          }
        }
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace flash_hip
