// fp32 -> fp16 conversion pass for callers whose tensors are fp32.
//
// The reference's callers hand fp32 tensors to flash::forward (reference main.cpp:30-32,
// benchmarks/bench.py:34-36; read with data_ptr<float>() at flash/launch.cu:165-176).  The MFMA
// kernels want fp16 operands, and every K/V row is read by N/256 workgroups, so the rounding is done
// once, in one streaming pass over Q, K and V into a caller-provided workspace, instead of once per
// reader.  Round to nearest even (v_cvt_pk_f16_f32 in the default rounding mode), the same rounding
// torch's .half() applies.
//
// HBM-bound: 4 bytes read + 2 bytes written per element.  A lane converts four 4-element pieces per
// trip; lane i of a workgroup takes piece i, 256 + i, 512 + i, 768 + i of the trip's 4096 elements, so
// every wave instruction reads 1 KiB / writes 512 B of consecutive memory and four loads are in flight
// per lane before the first conversion.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace flash_hip {

constexpr int kCastThreads = 256;
constexpr int kCastTensors = 3;

struct CastParams {
  const float* src[kCastTensors];
  _Float16* dst[kCastTensors];
  uint64_t n4;  // pieces of 4 elements per tensor
};

__global__ __launch_bounds__(kCastThreads) void cast_f32_to_f16_kernel(const CastParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  constexpr int kUnroll = 4;
  const f32x4_t* __restrict__ src = (const f32x4_t*)p.src[blockIdx.y];
  f16x4_t* __restrict__ dst = (f16x4_t*)p.dst[blockIdx.y];
  const uint64_t trip = (uint64_t)kCastThreads * kUnroll;
  for (uint64_t base = (uint64_t)blockIdx.x * trip; base < p.n4; base += (uint64_t)gridDim.x * trip) {
    f32x4_t x[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint64_t i = base + (uint64_t)u * kCastThreads + threadIdx.x;
      if (i < p.n4) x[u] = __builtin_nontemporal_load(src + i);  // read once
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint64_t i = base + (uint64_t)u * kCastThreads + threadIdx.x;
      if (i < p.n4) {
        f16x4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (_Float16)x[u][j];
        dst[i] = h;  // re-read soon by the attention kernel: a normal (cached) store
      }
    }
  }
#endif
}

// Head dims that are not a multiple of 8 (the reference's own test shapes use 50 and 69, reference
// main.cpp:157-193): rows of such tensors have no 16-byte chunks, which the 16x16x32 kernels need for
// their direct-to-LDS loads and 128-bit stores.  With a workspace the rows are copied once into rows of
// d_pad = the next multiple of 8 elements, zero-filled (zero columns add nothing to Q K^T and produce
// zero output columns), the fast kernels run on the padded copies, and one more pass copies the first d
// columns of the padded output into the caller's O.  Both passes are HBM-bound and short next to the
// attention kernel; the element-staging kernel they replace runs at a sixth of the fast kernels' rate.
//
// pad_rows_kernel: one lane per 8-element chunk of a padded row.  Workgroup x takes blocks of 64 rows; a
// chunk index inside the block is split into (row, chunk) with 32-bit arithmetic.  Consecutive lanes read
// consecutive 32-byte spans of the source (except at row ends) and write consecutive 16 or 32 bytes.
// Tensor y of the call is written as fp16 or, when bit y of `f32_mask` is set, as fp32 (Q for the wave
// layouts that read it as fp32).  Measured (rocprofv3 kernel trace): 4.6-4.8 TB/s of reads plus writes at
// (4,96,1024...4096,69), the unpad pass 4.9-5.3 TB/s, against 6.3 TB/s for a plain 128-bit copy on this
// chip.  A wave-per-row form of this kernel (lane l holding elements 2l and 2l + 1, every wave access one
// contiguous span, eight rows in flight) measured no faster: the lane stride of the loads is not what
// bounds the pass.
struct PadParams {
  const float* src[kCastTensors];
  void* dst[kCastTensors];
  uint64_t rows;      // B * H * N
  uint32_t d, d_pad;  // d < d_pad <= 128, d_pad % 8 == 0
  uint32_t f32_mask;
};

constexpr uint32_t kPadRowsPerBlock = 64;

__global__ __launch_bounds__(kCastThreads) void pad_rows_kernel(const PadParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
  const float* __restrict__ src = p.src[blockIdx.y];
  const bool as_f32 = (p.f32_mask >> blockIdx.y) & 1u;
  const uint32_t cpr = p.d_pad / 8;  // chunks per row
  const uint64_t blocks = (p.rows + kPadRowsPerBlock - 1) / kPadRowsPerBlock;
  for (uint64_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    const uint64_t row0 = blk * kPadRowsPerBlock;
    const uint32_t nrows = (uint32_t)(p.rows - row0 < kPadRowsPerBlock ? p.rows - row0 : kPadRowsPerBlock);
    for (uint32_t li = threadIdx.x; li < nrows * cpr; li += kCastThreads) {
      const uint32_t r = li / cpr, c = li - r * cpr;
      const float* s = src + (row0 + r) * p.d + c * 8;
      const uint32_t n = p.d - c * 8;  // elements of the row from this chunk on (>= 1)
      float x[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) x[j] = j < n ? s[j] : 0.f;
      const uint64_t chunk = (row0 + r) * cpr + c;
      if (as_f32) {
        f32x4_t* d4 = (f32x4_t*)p.dst[blockIdx.y] + chunk * 2;
        d4[0] = f32x4_t{x[0], x[1], x[2], x[3]};
        d4[1] = f32x4_t{x[4], x[5], x[6], x[7]};
      } else {
        f16x8_t h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)x[j];  // round to nearest even, as the conversion pass
        ((f16x8_t*)p.dst[blockIdx.y])[chunk] = h;
      }
    }
  }
#endif
}

// unpad_rows_kernel: one wave per row of the padded fp32 output, lanes over its first d columns (d <= 128:
// two columns per lane), four rows in flight per wave.
struct UnpadParams {
  const float* src;  // [rows, d_pad]
  float* dst;        // [rows, d]
  uint64_t rows;
  uint32_t d, d_pad;
};

__global__ __launch_bounds__(kCastThreads) void unpad_rows_kernel(const UnpadParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint32_t kWavesPerBlock = kCastThreads / 64, kRowsInFlight = 4;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock * kRowsInFlight;
  for (uint64_t row0 = ((uint64_t)blockIdx.x * kWavesPerBlock + wave) * kRowsInFlight; row0 < p.rows; row0 += stride) {
    float a[kRowsInFlight], b[kRowsInFlight];
#pragma unroll
    for (uint32_t u = 0; u < kRowsInFlight; ++u) {
      const uint64_t row = row0 + u;
      const float* s = p.src + row * p.d_pad;
      a[u] = (row < p.rows && lane < p.d) ? __builtin_nontemporal_load(s + lane) : 0.f;
      b[u] = (row < p.rows && lane + 64 < p.d) ? __builtin_nontemporal_load(s + lane + 64) : 0.f;
    }
#pragma unroll
    for (uint32_t u = 0; u < kRowsInFlight; ++u) {
      const uint64_t row = row0 + u;
      float* t = p.dst + row * p.d;
      if (row < p.rows && lane < p.d) t[lane] = a[u];
      if (row < p.rows && lane + 64 < p.d) t[lane + 64] = b[u];
    }
  }
#endif
}

}  // namespace flash_hip
