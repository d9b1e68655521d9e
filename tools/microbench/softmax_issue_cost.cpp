// Scratch microbenchmark: issue cost (cycles) of the candidate instructions of the score -> probability
// chain between 16x16x32 MFMAs, one wave per SIMD.  Each variant runs NF fillers per MFMA; if the wave is
// issue-bound, cycles per MFMA = 8 + NF * cost.  Whole per-pair sequences (two scores -> one packed pair of
// probabilities) are measured as one filler.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Kind {
  NONE, MUL_S, EXP32, CVT_PK, MAX3, MIXLO, MIXHI, EXP16, EXP16_SDWA_HI, EXP16_SDWA_LO, PK_MAX3_F16, PK_MAX_F16,
  PK_MUL_F16, MAX_F32, CVT_F16, OR3, PAIR_OLD, PAIR_OLD_MAX, PAIR_MIX, PAIR_MIX_PKMAX, PAIR_CVT_PKMUL, ACC_READ
};

template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void k(const f16x8* in, float* out, long long* cyc, int iters, float cs) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    b[i] = in[(tid * 8 + 4 + i) & 0xffff];
  }
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = (float)a[i & 3][i & 7] * 0.01f;
  unsigned int u[8];
  for (int i = 0; i < 8; ++i) u[i] = 0x38003400u + (unsigned)(tid & 255) + i;
  f32x4 c4[8];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc = {1.f, 2.f, 3.f, 4.f};
  asm volatile("" : "+a"(acc));
  unsigned c = __builtin_amdgcn_readfirstlane(__float_as_uint(cs));
  asm volatile("" : "+s"(c));
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[m]) : "v"(a[m & 3]), "v"(b[(m >> 1) & 3]));
#pragma unroll
      for (int e = 0; e < NF; ++e) {
        const int x = (m * NF + e) & 15, y = (x + 5) & 15, z = (x + 9) & 15;
        unsigned& ux = u[x & 7];
        unsigned& uy = u[y & 7];
        unsigned& uz = u[z & 7];
        if constexpr (KIND == MUL_S) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[x]) : "s"(c), "v"(f[y]));
        if constexpr (KIND == EXP32) asm volatile("v_exp_f32 %0, %1" : "=v"(f[x]) : "v"(f[y]));
        if constexpr (KIND == CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ux) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == MAX3) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]), "v"(f[(x + 3) & 15]));
        if constexpr (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(ux) : "v"(f[y]), "s"(c));
        if constexpr (KIND == MIXHI) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(ux) : "v"(f[y]), "s"(c));
        if constexpr (KIND == EXP16) asm volatile("v_exp_f16 %0, %1" : "+v"(ux) : "v"(uy));
        if constexpr (KIND == EXP16_SDWA_HI)
          asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(ux) : "v"(uy));
        if constexpr (KIND == EXP16_SDWA_LO)
          asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(ux) : "v"(uy));
        if constexpr (KIND == PK_MAX3_F16) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(ux) : "v"(uy), "v"(uz), "v"(u[(x + 3) & 7]));
        if constexpr (KIND == PK_MAX_F16) asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(ux) : "v"(uy), "v"(uz));
        if constexpr (KIND == PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(ux) : "v"(uy), "v"(uz));
        if constexpr (KIND == MAX_F32) asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == CVT_F16) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(ux) : "v"(f[y]));
        if constexpr (KIND == OR3) asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(ux) : "v"(uy), "v"(uz), "v"(u[(x + 3) & 7]));
        if constexpr (KIND == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[x]) : "a"(acc[e & 3]));
        if constexpr (KIND == PAIR_OLD || KIND == PAIR_OLD_MAX) {
          // two scores: multiply, exponential, packed conversion (the shipped chain), optionally the running max
          float t0_, t1_;
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0_) : "s"(c), "v"(f[y]));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1_) : "s"(c), "v"(f[z]));
          asm volatile("v_exp_f32 %0, %0" : "+v"(t0_));
          asm volatile("v_exp_f32 %0, %0" : "+v"(t1_));
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ux) : "v"(t0_), "v"(t1_));
          if constexpr (KIND == PAIR_OLD_MAX)
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(x + 3) & 15]) : "v"(f[y]), "v"(f[z]));
        }
        if constexpr (KIND == PAIR_MIX || KIND == PAIR_MIX_PKMAX) {
          // two scores: fused multiply + conversion into the halves of one register, fp16 exponentials in place
          unsigned t;
          asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(t) : "v"(f[y]), "s"(c));
          asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(t) : "v"(f[z]), "s"(c));
          asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(t));
          asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(t));
          ux = t;
          if constexpr (KIND == PAIR_MIX_PKMAX) {
            if (e & 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(u[(x + 3) & 7]) : "v"(ux), "v"(uy));
          }
        }
        if constexpr (KIND == PAIR_CVT_PKMUL) {
          unsigned t;
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(t) : "v"(f[y]), "v"(f[z]));
          asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(t) : "v"(uy));
          asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(t));
          asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(t));
          ux = t;
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < 16; ++i) sum += f[i];
  for (int i = 0; i < 8; ++i) sum += c4[i][0] + c4[i][3];
  for (int i = 0; i < 8; ++i) sum += (float)u[i];
  out[tid] = sum;
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <int KIND, int NF>
void run(const f16x8* in, float* out, long long* cyc, const char* name) {
  const int iters = 2000;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters, 0.1275f);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("%s: launch failed\n", name);
    exit(1);
  }
  std::vector<long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per = (double)h[512] / (iters * 8.0);
  printf("%-34s NF=%d  cycles/MFMA %.1f   => per filler %.2f\n", name, NF, per, NF ? (per - 8.0) / NF : 0.0);
  fflush(stdout);
}

int main() {
  f16x8* in;
  float* out;
  long long* cyc;
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& x : h) {
    float u = 0.f;
    for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX;
    x = (_Float16)(u - 6.f);
  }
  (void)hipMalloc(&in, h.size() * 2);
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 1024 * 8);
  (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  run<NONE, 0>(in, out, cyc, "none");
  run<MUL_S, 8>(in, out, cyc, "v_mul_f32 (sgpr)");
  run<EXP32, 6>(in, out, cyc, "v_exp_f32");
  run<CVT_PK, 6>(in, out, cyc, "v_cvt_pk_f16_f32");
  run<MAX3, 8>(in, out, cyc, "v_max3_f32");
  run<MAX_F32, 8>(in, out, cyc, "v_max_f32");
  run<MIXLO, 8>(in, out, cyc, "v_fma_mixlo_f16");
  run<MIXHI, 8>(in, out, cyc, "v_fma_mixhi_f16");
  run<EXP16, 6>(in, out, cyc, "v_exp_f16");
  run<EXP16_SDWA_LO, 6>(in, out, cyc, "v_exp_f16_sdwa lo");
  run<EXP16_SDWA_HI, 6>(in, out, cyc, "v_exp_f16_sdwa hi");
  run<PK_MAX3_F16, 8>(in, out, cyc, "v_pk_maximum3_f16");
  run<PK_MAX_F16, 8>(in, out, cyc, "v_pk_max_f16");
  run<PK_MUL_F16, 8>(in, out, cyc, "v_pk_mul_f16");
  run<CVT_F16, 8>(in, out, cyc, "v_cvt_f16_f32");
  run<OR3, 8>(in, out, cyc, "v_or3_b32");
  run<ACC_READ, 8>(in, out, cyc, "v_accvgpr_read_b32");
  run<PAIR_OLD, 1>(in, out, cyc, "pair: 2 mul 2 exp32 cvt_pk");
  run<PAIR_OLD, 2>(in, out, cyc, "pair: 2 mul 2 exp32 cvt_pk");
  run<PAIR_OLD_MAX, 2>(in, out, cyc, "pair: 2 mul 2 exp32 cvt_pk max3");
  run<PAIR_MIX, 1>(in, out, cyc, "pair: mixlo mixhi 2 exp16");
  run<PAIR_MIX, 2>(in, out, cyc, "pair: mixlo mixhi 2 exp16");
  run<PAIR_MIX_PKMAX, 2>(in, out, cyc, "pair: mixlo mixhi 2 exp16 + pkmax3/2");
  run<PAIR_CVT_PKMUL, 2>(in, out, cyc, "pair: cvt_pk pk_mul_f16 2 exp16");
  return 0;
}
