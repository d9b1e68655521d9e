// Scratch microbenchmark: sustained fp16 MFMA rate of the two shapes on random operands, with and
// without VALU fillers, one wave per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int FILL>
__global__ __launch_bounds__(256, 1) void k(const f16x8* in, float* out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    b[i] = in[(tid * 8 + 4 + i) & 0xffff];
  }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = (float)a[0][i];
  float acc_out = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 c[8];
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u & 7]) : "v"(a[u & 3]), "v"(b[(u >> 2) & 3]));
#pragma unroll
        for (int e = 0; e < FILL; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[e & 7]) : "v"(f[(e + 1) & 7]), "v"(f[(e + 2) & 7]));
      }
    }
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 16; ++j) acc_out += c[i][j];
  } else {
    f32x4 c[32];
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 4; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[u & 31]) : "v"(a[u & 3]), "v"(b[(u >> 2) & 3]));
        if (u & 1) {
#pragma unroll
          for (int e = 0; e < FILL; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[e & 7]) : "v"(f[(e + 1) & 7]), "v"(f[(e + 2) & 7]));
        }
      }
    }
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 4; ++j) acc_out += c[i][j];
  }
  for (int i = 0; i < 8; ++i) acc_out += f[i];
  out[tid] = acc_out;
}

template <int SHAPE, int FILL>
void run(const f16x8* in, float* out, const char* name) {
  const int iters = 4000;  // 16 * 32768 flop * 64 lanes... per wave per iter: 16 MFMA(32) or 32 MFMA(16) = 524288 flop
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, in, out, iters);
  std::vector<float> ms;
  for (int r = 0; r < 30; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t;
    hipEventElapsedTime(&t, e0, e1);
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double flop = 256.0 * 4 * iters * 16 * 32768.0;
  const double cyc_ideal = iters * 16 * 32.0;  // per SIMD
  printf("%-28s median %.3f ms  %.1f TFLOP/s   (MFMA-bound clock if no stall: %.2f GHz)\n", name, ms[15],
         flop / (ms[15] * 1e-3) / 1e12, cyc_ideal / (ms[15] * 1e-3) / 1e9);
}

int main() {
  f16x8* in;
  float* out;
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& x : h) {
    float u = 0.f;
    for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX;
    x = (_Float16)(u - 6.f);
  }
  hipMalloc(&in, h.size() * 2);
  hipMalloc(&out, 256 * 256 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 0>(in, out, "32x32x16 bare");
    run<16, 0>(in, out, "16x16x32 bare");
    run<32, 3>(in, out, "32x32x16 + 3 fma/32cyc");
    run<16, 3>(in, out, "16x16x32 + 3 fma/32cyc");
    run<32, 5>(in, out, "32x32x16 + 5 fma/32cyc");
    run<16, 5>(in, out, "16x16x32 + 5 fma/32cyc");
    run<32, 7>(in, out, "32x32x16 + 7 fma/32cyc");
    run<16, 7>(in, out, "16x16x32 + 7 fma/32cyc");
  }
  return 0;
}
