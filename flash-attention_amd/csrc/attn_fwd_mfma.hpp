// FlashAttention-2 forward for gfx950 (MI355X, CDNA4): fp16 operands on
// v_mfma_f32_32x32x16_f16, fp32 accumulation, wave64.
//
// Functional counterpart of the reference's kernel_mma_qreg family
// (reference flash/kernel_mma_qreg.cuh:16-267, kernel_mma_qreg_f32x4load.cuh,
// kernel_mma_qreg_async.cuh): one workgroup owns a tile of Q rows of one (batch, head), keeps Q in
// registers, streams K/V tiles through LDS and keeps the running row max / row sum in registers.
// Nothing of the reference's structure is kept beyond that:
//
//   * A workgroup is 8 waves x 32 query rows = 256 rows (Br), K/V tiles are 64 rows (Bc).
//   * Scores are produced TRANSPOSED: S^T = K * Q^T, A operand = K rows from LDS (ds_read_b128),
//     B operand = Q held in VGPRs.  In the 32x32 accumulator layout the query index is then the
//     lane (lane & 31) and the key index runs over the 16 registers, so the row max / row sum of
//     the online softmax are in-lane reductions plus ONE v_permlane32_swap between the two
//     half-waves; no LDS round trip for S or P (the reference stores S to shared memory three
//     times per tile, kernel_mma_qreg.cuh:166,185,200).
//   * The output is accumulated TRANSPOSED as well: O^T = V^T * P^T.  P^T is exactly the S^T
//     accumulator converted to fp16 (8 consecutive registers = one B fragment with a permuted key
//     order), V^T fragments come from the row-major V tile with ds_read_b64_tr_b16 using the same
//     permuted key order.  The softmax rescale factor is a per-lane scalar for every register of
//     O^T.  O stays in registers for the whole K/V sweep and is divided by the row sum once
//     (the reference read-modify-writes O in HBM for every K/V tile, kernel_mma_qreg.cuh:207-250).
//   * K/V tiles are double-buffered in LDS.  Staging is one of: direct global->LDS
//     (buffer_load_dwordx4 ... lds), or through registers for head dims that are not a multiple
//     of 8 halves and for fp32 callers.  The LDS image is XOR-swizzled at 16-byte granularity so
//     that both the K row reads and the V transposed reads are bank-conflict free.
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

constexpr int kWaves = 8;                   // waves per workgroup
constexpr int kRowsPerWave = 32;            // query rows per wave (one 32x32 MFMA column block)
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
  float scale_log2e;  // softmax_scale * log2(e)
};

template <int D>
struct Geom {
  static_assert(D == 64 || D == 128, "compiled head-dim tiles are 64 and 128");
  static constexpr int kRowBytes = D * 2;             // LDS row of fp16
  static constexpr int kChunks = D / 8;               // 16-byte chunks per row
  static constexpr int kKSteps = D / 16;              // MFMA k-steps of S^T = K Q^T
  static constexpr int kDvBlocks = D / 32;            // 32-wide blocks of O^T rows
  static constexpr int kTileBytes = kBc * kRowBytes;  // one K or V tile
  static constexpr int kStageBytes = 2 * kTileBytes;  // K + V
  static constexpr int kLdsBytes = 2 * kStageBytes;   // double buffered
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

// x of this lane's row as held by the lower (r[0]) and the upper (r[1]) half-wave.
struct HalfPair {
  float lo, hi;
};
__device__ __forceinline__ HalfPair both_halves(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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

__device__ __forceinline__ f16 to_f16(f16 x) { return x; }
__device__ __forceinline__ f16 to_f16(float x) { return (f16)x; }

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
// The unit of work is a BLOCK of 32 keys (half an LDS tile):
//   QK       S^T(32 keys x 32 queries) = K_blk * Q^T            D/16 MFMAs, A operand by ds_read_b128
//   softmax  running max, P^T = exp2(S^T*c - m*c) as fp16 B fragments, row sum      (VALU only)
//   PV       O^T += V_blk^T * P^T                                D/16 MFMAs, A operand by ds_read_b64_tr_b16
// One STEP of the kernel runs PV(j-1) and QK(j+1) (the "matrix section") and softmax(j) (the
// "vector section") of three consecutive blocks; the two sections are independent of each other.

// Per-lane byte offsets into a K or V tile (for key row 0 of a block; blocks and k-steps add
// multiples of 16 rows, which the swizzle does not see, so they become instruction immediates).
template <int D>
struct LaneAddr {
  int k[Geom<D>::kKSteps];        // K[r][16*ks + 8*h ..]: A fragment of k-step ks of S^T = K Q^T
  int v[2][Geom<D>::kDvBlocks];   // V rows 8*jj + 4*h + (0..3), dv block b: half jj of an A fragment of O^T = V^T P^T
};

template <int D>
__device__ __forceinline__ LaneAddr<D> make_lane_addr(int lane) {
  using G = Geom<D>;
  LaneAddr<D> la;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int i16 = lane & 15;
  const int tq = i16 >> 2;         // row of the 4x16 transposed block this lane addresses
  const int tp = i16 & 3;          // 4-column group of that row
  const int tg = (lane >> 4) & 1;  // which 16 of the 32 dv columns
#pragma unroll
  for (int ks = 0; ks < G::kKSteps; ++ks) la.k[ks] = lds_off<D>(r, 2 * ks + h);
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
    for (int b = 0; b < G::kDvBlocks; ++b)
      la.v[jj][b] = lds_off<D>(8 * jj + 4 * h + tq, b * 4 + tg * 2 + (tp >> 1)) + 8 * (tp & 1);
  }
  return la;
}

// A fragment of O^T = V^T P^T for dv block b, k-step ks (keys 16*ks.. of the block at `vb`): element j
// of half-wave h is key 16*ks + 8*(j>>2) + 4*h + (j&3), the order the registers of P^T already have.
template <int D>
__device__ __forceinline__ f16x8 read_v_frag(lds_char* vb, const LaneAddr<D>& la, int b, int ks) {
  f16x8 a;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const fp16x4_b t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        FLASH_LDS_CAST(fp16x4_b, vb + 16 * ks * Geom<D>::kRowBytes + la.v[jj][b]));
    const f16x4 tv = __builtin_bit_cast(f16x4, t);
    a[4 * jj + 0] = tv[0];
    a[4 * jj + 1] = tv[1];
    a[4 * jj + 2] = tv[2];
    a[4 * jj + 3] = tv[3];
  }
  return a;
}

template <int D>
__device__ __forceinline__ f16x8 read_k_frag(lds_char* kb, const LaneAddr<D>& la, int ks) {
  return *(FLASH_LDS_CAST(f16x8, kb + la.k[ks]));
}

// Matrix section of one step.
//   HAS_PV: O^T += V^T P^T for the 32 keys whose V rows start at `vb`, probabilities `pf`
//   HAS_QK: s_out = K Q^T for the 32 keys whose K rows start at `kb`
// LDS reads are issued ahead of the MFMAs that consume them: all K fragments and the V fragments
// of the first two dv blocks up front, the V fragments of the next two dv blocks as soon as the
// registers of the previous pair are free, with the QK MFMAs in between to cover their latency.
template <int D, bool HAS_PV, bool HAS_QK>
__device__ __forceinline__ void matrix_section(lds_char* vb, const f16x8 (&pf)[2], f32x16 (&o)[Geom<D>::kDvBlocks],
                                               lds_char* kb, const f16x8 (&qf)[Geom<D>::kKSteps], f32x16& s_out,
                                               const LaneAddr<D>& la) {
  using G = Geom<D>;
  constexpr int KS = G::kKSteps;
  constexpr int NB = G::kDvBlocks;
  constexpr int KH = (NB > 2) ? KS / 2 : KS;  // K fragments held at a time
  f16x8 va[4];
  f16x8 ka[KH];
  if constexpr (HAS_PV) {
    va[0] = read_v_frag<D>(vb, la, 0, 0);
    va[1] = read_v_frag<D>(vb, la, 0, 1);
    va[2] = read_v_frag<D>(vb, la, 1, 0);
    va[3] = read_v_frag<D>(vb, la, 1, 1);
  }
  if constexpr (HAS_QK) {
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) ka[ks] = read_k_frag<D>(kb, la, ks);
#pragma unroll
    for (int i = 0; i < 16; ++i) s_out[i] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b0 = 0; b0 < NB; b0 += 2) {
    if constexpr (HAS_PV) {
      o[b0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[0], pf[0], o[b0], 0, 0, 0);
      o[b0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[1], pf[1], o[b0], 0, 0, 0);
      if (b0 + 2 < NB) {
        va[0] = read_v_frag<D>(vb, la, b0 + 2, 0);
        va[1] = read_v_frag<D>(vb, la, b0 + 2, 1);
      }
      o[b0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[2], pf[0], o[b0 + 1], 0, 0, 0);
      o[b0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[3], pf[1], o[b0 + 1], 0, 0, 0);
      if (b0 + 2 < NB) {
        va[2] = read_v_frag<D>(vb, la, b0 + 3, 0);
        va[3] = read_v_frag<D>(vb, la, b0 + 3, 1);
      }
    }
    if constexpr (HAS_QK) {
      // KH steps of the QK chain per dv-block pair; each K fragment register is refilled with the
      // fragment KH steps later as soon as its MFMA has issued
      const int k0 = (b0 / 2) * KH;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        s_out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[ks], qf[k0 + ks], s_out, 0, 0, 0);
        if (k0 + KH + ks < KS) ka[ks] = read_k_frag<D>(kb, la, k0 + KH + ks);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Running softmax state of one lane: its query is lane & 31, its keys are those of its half-wave.
struct SoftmaxState {
  float m;      // reference max of the raw scores that P and O are currently scaled against
  float l;      // this lane's partial row sum at that scale
  float alpha;  // pending rescale factor for O (1 if none); applied by the caller after the
                // previous block's P V product has been accumulated
};

// How far (in exp2 units, i.e. score * scale * log2 e) the row max may run ahead of `m` before the
// state is rescaled.  P then stays below 2^kDeferLog2 = 256, far inside fp16 range, and O / l are
// fp32, so the result is the same up to rounding; it saves the O-wide multiply on most blocks.
constexpr float kDeferLog2 = 8.0f;

// Softmax of one S^T block (keys kv0 + (i&3) + 8*(i>>2) + 4*h for register i).  Returns P^T as the
// two fp16 B fragments of the block.  `masked`: keys >= N are excluded (wave-uniform flag).
// Returns true (wave-uniform) when the caller has to multiply O by st.alpha.
template <int D>
__device__ __forceinline__ bool softmax_block(f32x16& s, f16x8 (&pf)[2], SoftmaxState& st, float c, int kv0, int N,
                                              bool masked, int lane) {
  const int h = lane >> 5;
  if (masked) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kv = kv0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (kv >= N) s[i] = -INFINITY;
    }
  }
  float mx = fmaxf(s[0], s[1]);
#pragma unroll
  for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(s[i], s[i + 1]));
  const HalfPair mp = both_halves(mx);
  mx = fmaxf(mp.lo, mp.hi);
  // wave-uniform decision: rescale only if some row's max ran ahead of its reference by more than
  // the deferral window (always true for the first block, where m is -huge)
  const bool need = (mx - st.m) * c > kDeferLog2;
  const bool rescale = __builtin_amdgcn_ballot_w64(need) != 0;
  st.alpha = 1.0f;
  if (rescale) {
    const float m_new = fmaxf(st.m, mx);
    st.alpha = fast_exp2((st.m - m_new) * c);
    st.m = m_new;
    st.l *= st.alpha;
  }
  const float mc = st.m * c;
  float rs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float pv = fast_exp2(__builtin_fmaf(s[i], c, -mc));
    rs += pv;
    pf[i >> 3][i & 7] = (f16)pv;
  }
  st.l += rs;
  return rescale;
}

template <int D>
__device__ __forceinline__ void rescale_o(f32x16 (&o)[Geom<D>::kDvBlocks], float alpha) {
#pragma unroll
  for (int b = 0; b < Geom<D>::kDvBlocks; ++b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) o[b][i] *= alpha;
  }
}

// ---- the kernel ---------------------------------------------------------------------------------
// D: compiled head-dim tile (d <= D, missing columns are zero-filled).
// T: element type of Q/K/V/O in HBM (f16 or float; float is rounded to f16 while staging).
// FAST: T == f16 and d % 8 == 0: 128-bit Q loads, direct global->LDS K/V staging, 64-bit O stores.
//       Otherwise everything goes through range-checked element loads/stores.
//
// LDS: a ring of two K tiles followed by a ring of two V tiles (64 rows each).  Block schedule for
// tile t (blocks 2t and 2t+1), with S and P double-buffered in registers as (sA, pA) / (sB, pB):
//   step 2t   : PV(2t-1) [V tile t-1, rows 32..63]  |  S(2t+1) [K tile t, rows 32..63]  |  softmax(2t)
//               wait for outstanding staging, barrier; stage K tile t+2 -> slot t%2 (K tile t is
//               now fully consumed) and V tile t+1 -> slot (t+1)%2 (V tile t-1 fully consumed)
//   step 2t+1 : PV(2t) [V tile t, rows 0..31]       |  S(2t+2) [K tile t+1, rows 0..31] |  softmax(2t+1)
// so every staged tile has a whole tile time to land and there is one barrier per tile.
template <int D, typename T, bool FAST>
__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(const FwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  using G = Geom<D>;
  static_assert(!FAST || sizeof(T) == 2, "direct-to-LDS staging needs fp16 data in HBM");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_char* const smem = (lds_char*)smem_raw;

  // Work item: consecutive items = query tiles of one head; items are dealt to the 8 XCDs in
  // contiguous ranges (workgroup ids that are equal mod 8 share an XCD and hence an L2), so the
  // workgroups that stream the same K/V run on the same L2 at about the same time.
  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot = bid >> 3;
  const uint32_t per = p.total_wg >> 3, rem = p.total_wg & 7u;
  const uint32_t item = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
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
  lds_char* const v_ring = smem + 2 * G::kTileBytes;

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
  auto stage_k = [&](int t) __attribute__((always_inline)) { stage(k_rs, k_srd, t, k_ring + (t & 1) * G::kTileBytes); };
  auto stage_v = [&](int t) __attribute__((always_inline)) { stage(v_rs, v_srd, t, v_ring + (t & 1) * G::kTileBytes); };
  auto staged_barrier = [&]() __attribute__((always_inline)) {
    // every wave waits for its own staging traffic, then all waves meet: tiles issued before this
    // point are readable by everyone, tiles read before this point may be overwritten
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nt = (N + kBc - 1) / kBc;
  stage_k(0);
  stage_v(0);
  if (nt > 1) stage_k(1);

  // ---- Q fragments: B operand of S^T = K Q^T: lane (r, h) holds Q[row r][16*ks + 8*h + 0..7] ----
  const int qrow = (int)qt * kBr + wave * kRowsPerWave + r;
  f16x8 qf[G::kKSteps];
#pragma unroll
  for (int ks = 0; ks < G::kKSteps; ++ks) {
    const int col = 16 * ks + 8 * h;
    if constexpr (FAST) {
      const uint32_t off = (col < d) ? ((uint32_t)qrow * (uint32_t)d + (uint32_t)col) * 2u : kOob;
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(q_rs, off, 0, 0);
      qf[ks] = __builtin_bit_cast(f16x8, raw);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t off =
            (col + j < d) ? ((uint32_t)qrow * (uint32_t)d + (uint32_t)(col + j)) * (uint32_t)sizeof(T) : kOob;
        qf[ks][j] = load_elem<T>(q_rs, off);
      }
    }
  }

  f32x16 o[G::kDvBlocks];
#pragma unroll
  for (int b = 0; b < G::kDvBlocks; ++b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
  }
  SoftmaxState st;
  st.m = -3.0e38f;  // finite, so that the first rescale factor is exp2(-huge) = 0, never NaN
  st.l = 0.f;
  st.alpha = 1.f;
  const float c = p.scale_log2e;
  const bool ragged = (N % kBc) != 0;

  staged_barrier();

  const LaneAddr<D> la = make_lane_addr<D>(lane);
  f32x16 sA, sB;
  f16x8 pA[2], pB[2];
  constexpr int kBlk = 32 * G::kRowBytes;  // bytes of one 32-key block inside a tile

  // S(0)
  matrix_section<D, false, true>(nullptr, pA, o, k_ring, qf, sA, la);

  // One tile = two steps.  VF (wave-uniform): run the vector section before the matrix section.
  // The two waves that share a SIMD (wave w and w + 4) use opposite orders, so that one of them
  // feeds the matrix pipe while the other one runs its softmax.
  auto tile_steps = [&](auto vf_tag, auto first_tag, auto last_tag, int t, int slot) __attribute__((always_inline)) {
    constexpr bool VF = decltype(vf_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;  // t == 0: there is no previous block to multiply
    constexpr bool LAST = decltype(last_tag)::value;    // t == nt-1: there is no next tile to score
    lds_char* const k_cur = k_ring + slot * G::kTileBytes;
    lds_char* const k_nxt = k_ring + (slot ^ 1) * G::kTileBytes;
    lds_char* const v_cur = v_ring + slot * G::kTileBytes;
    lds_char* const v_prv = v_ring + (slot ^ 1) * G::kTileBytes;
    const bool masked = LAST && ragged;
    bool rs;
    // step 2t: PV(2t-1) | QK(2t+1) | softmax(2t)
    if constexpr (VF) {
      rs = softmax_block<D>(sA, pA, st, c, t * kBc, N, masked, lane);
      __builtin_amdgcn_sched_barrier(0);
      matrix_section<D, !FIRST, true>(v_prv + kBlk, pB, o, k_cur + kBlk, qf, sB, la);
    } else {
      matrix_section<D, !FIRST, true>(v_prv + kBlk, pB, o, k_cur + kBlk, qf, sB, la);
      rs = softmax_block<D>(sA, pA, st, c, t * kBc, N, masked, lane);
    }
    if (rs) rescale_o<D>(o, st.alpha);
    staged_barrier();
    if (t + 2 < nt) stage_k(t + 2);
    if (t + 1 < nt) stage_v(t + 1);
    // step 2t+1: PV(2t) | QK(2t+2) | softmax(2t+1)
    if constexpr (VF) {
      rs = softmax_block<D>(sB, pB, st, c, t * kBc + 32, N, masked, lane);
      __builtin_amdgcn_sched_barrier(0);
      matrix_section<D, true, !LAST>(v_cur, pA, o, k_nxt, qf, sA, la);
    } else {
      matrix_section<D, true, !LAST>(v_cur, pA, o, k_nxt, qf, sA, la);
      rs = softmax_block<D>(sB, pB, st, c, t * kBc + 32, N, masked, lane);
    }
    if (rs) rescale_o<D>(o, st.alpha);
  };
  auto sweep = [&](auto vf_tag) __attribute__((always_inline)) {
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    if (nt == 1) {
      tile_steps(vf_tag, yes, yes, 0, 0);
    } else {
      tile_steps(vf_tag, yes, no, 0, 0);
      int t = 1;
      for (; t + 2 < nt; t += 2) {  // slots alternate 1, 0: constant LDS offsets inside the loop
        tile_steps(vf_tag, no, no, t, 1);
        tile_steps(vf_tag, no, no, t + 1, 0);
      }
      if (t + 1 < nt) {
        tile_steps(vf_tag, no, no, t, 1);
        ++t;
      }
      tile_steps(vf_tag, no, yes, t, t & 1);
    }
  };
  if (wave >= kWaves / 2) {
    sweep(std::true_type{});
  } else {
    sweep(std::false_type{});
  }
  // PV of the last block
  matrix_section<D, true, false>(v_ring + ((nt - 1) & 1) * G::kTileBytes + kBlk, pB, o, nullptr, qf, sA, la);

  // ---- epilogue: O = O^T / l, registers 4g..4g+3 of block b are dv = 32b + 8g + 4h + 0..3 ----------
  const HalfPair lp = both_halves(st.l);
  const float inv = 1.0f / (lp.lo + lp.hi);
#pragma unroll
  for (int b = 0; b < G::kDvBlocks; ++b) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dv = 32 * b + 8 * g + 4 * h;
      if constexpr (FAST) {
        f16x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = (f16)(o[b][4 * g + j] * inv);
        const uint32_t off = (dv < d) ? ((uint32_t)qrow * (uint32_t)d + (uint32_t)dv) * 2u : kOob;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), o_rs, off, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t off =
              (dv + j < d) ? ((uint32_t)qrow * (uint32_t)d + (uint32_t)(dv + j)) * (uint32_t)sizeof(T) : kOob;
          store_elem(o_rs, off, (T)(o[b][4 * g + j] * inv));
        }
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace flash_hip
