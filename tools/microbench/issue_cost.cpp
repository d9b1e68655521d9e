// Scratch microbenchmark: issue cost (cycles) of filler instructions between 32x32x16 MFMAs, one wave
// per SIMD.  Each variant runs NF fillers per MFMA; cycles per MFMA = max(32, 8 + NF * cost).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL_NONE 0
#define FILL_ADD 1
#define FILL_EXP 2
#define FILL_PKADDH 3
#define FILL_DOT2 4
#define FILL_CVT 5
#define FILL_MAX3 6
#define FILL_WAIT 7
#define FILL_NOP 8
#define FILL_FMA 9
#define FILL_DSB128 10
#define FILL_DSTR 11
#define FILL_PKMULF32 12
#define FILL_MOV 13
#define FILL_PKRTZ 14
#define FILL_EXPH 15
#define FILL_MIX_B128 16   // 4 v_add + NF ds_read_b128 (conflict-free)
#define FILL_MIX_TR 17     // 4 v_add + NF ds_read_b64_tr_b16 (8-byte stride)
#define FILL_CMP 18
#define FILL_PERM 19
#define FILL_MFMA4 20   // v_mfma_f32_4x4x4_16b_f16 on an accumulator of its own
#define FILL_MUL 21

template <int KIND, int NF, int BASE = 0>
__global__ __launch_bounds__(256, 1) void k(const f16x8* in, float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int tid = threadIdx.x + blockIdx.x * 256;
  for (int i = threadIdx.x; i < 32768 / 16; i += 256) ((f16x8*)lds)[i] = in[(i + tid) & 0xffff];
  __syncthreads();
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    b[i] = in[(tid * 8 + 4 + i) & 0xffff];
  }
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = (float)a[i & 3][i & 7] * 0.01f;
  unsigned int u[8];
  for (int i = 0; i < 8; ++i) u[i] = (unsigned)tid * 2654435761u + i;
  f16x8 ld[4] = {a[0], a[1], a[2], a[3]};
  const unsigned lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  const unsigned lds_addr8 = (unsigned)(size_t)lds + (threadIdx.x & 63) * 8;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f32x4 c4[8], r4[4];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) r4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x4 h4[2] = {{a[0][0], a[0][1], a[0][2], a[0][3]}, {a[1][0], a[1][1], a[1][2], a[1][3]}};
  f32x16 c[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if constexpr (BASE == 0) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a[m & 3]), "v"(b[(m >> 1) & 3]));
      } else {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[m]) : "v"(a[m & 3]), "v"(b[(m >> 1) & 3]));
      }
      if constexpr (KIND == FILL_MIX_B128 || KIND == FILL_MIX_TR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int x = (m * 4 + e) & 15, y = (x + 5) & 15, z = (x + 9) & 15;
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]));
        }
      }
#pragma unroll
      for (int e = 0; e < NF; ++e) {
        const int x = (m * NF + e) & 15, y = (x + 5) & 15, z = (x + 9) & 15;
        if constexpr (KIND == FILL_ADD) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == FILL_FMA) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == FILL_EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(f[x]) : "v"(f[y]));
        if constexpr (KIND == FILL_PKADDH) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(u[x & 7]) : "v"(u[y & 7]), "v"(u[z & 7]));
        if constexpr (KIND == FILL_DOT2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(f[x]) : "v"(u[y & 7]), "v"(u[z & 7]), "v"(f[y]));
        if constexpr (KIND == FILL_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[x & 7]) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == FILL_MAX3) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]), "v"(f[(x + 3) & 15]));
        if constexpr (KIND == FILL_WAIT) asm volatile("s_waitcnt lgkmcnt(8)");
        if constexpr (KIND == FILL_NOP) asm volatile("s_nop 0");
        if constexpr (KIND == FILL_MFMA4) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(r4[e & 3]) : "v"(h4[0]), "v"(h4[1]));
        if constexpr (KIND == FILL_MUL) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[x]) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == FILL_PKMULF32) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&f[(2 * x) & 14]) : "v"(*(double*)&f[(2 * y) & 14]), "v"(*(double*)&f[(2 * z) & 14]));
        if constexpr (KIND == FILL_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(f[x]) : "v"(f[y]));
        if constexpr (KIND == FILL_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[x & 7]) : "v"(f[y]), "v"(f[z]));
        if constexpr (KIND == FILL_EXPH) asm volatile("v_exp_f16 %0, %1" : "=v"(u[x & 7]) : "v"(u[y & 7]));
        if constexpr (KIND == FILL_CMP) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[y]), "v"(f[z]) : "vcc");
        if constexpr (KIND == FILL_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[x & 7]) : "v"(u[y & 7]), "v"(u[z & 7]), "v"(u[(x+3)&7]));
        if constexpr (KIND == FILL_MIX_B128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[e & 3]) : "v"(lds_addr), "n"(1024 * ((m * NF + e) & 15)));
        if constexpr (KIND == FILL_MIX_TR) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*(double*)&ld[e & 3]) : "v"(lds_addr8), "n"(512 * ((m * NF + e) & 31)));
        if constexpr (KIND == FILL_DSB128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[e & 3]) : "v"(lds_addr), "n"(1024 * ((m * NF + e) & 15)));
        if constexpr (KIND == FILL_DSTR) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*(double*)&ld[e & 3]) : "v"(lds_addr), "n"(1024 * ((m * NF + e) & 15)));
      }
    }
    if constexpr (KIND == FILL_DSB128 || KIND == FILL_DSTR || KIND == FILL_MIX_B128 || KIND == FILL_MIX_TR) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) acc += c[i][j];
  for (int i = 0; i < 16; ++i) acc += f[i];
  for (int i = 0; i < 8; ++i) acc += c4[i][0] + c4[i][3];
  for (int i = 0; i < 4; ++i) acc += r4[i][0];
  for (int i = 0; i < 8; ++i) acc += (float)u[i];
  for (int i = 0; i < 4; ++i) acc += (float)ld[i][0];
  out[tid] = acc;
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <int KIND, int NF, int BASE = 0>
void run(const f16x8* in, float* out, long long* cyc, const char* name) {
  const int iters = 2000;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, NF, BASE>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per = (double)h[512] / (iters * 8.0);
  printf("%s %-22s NF=%d  cycles/MFMA %.1f   => per filler %.2f (if issue-bound: (cyc-8)/NF)\n", BASE ? "[16x16x32]" : "[32x32x16]", name, NF, per,
         NF ? (per - 8.0) / NF : 0.0);
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
  run<FILL_NONE, 0, 1>(in, out, cyc, "none");
  run<FILL_ADD, 1, 1>(in, out, cyc, "v_add_f32");
  run<FILL_ADD, 2, 1>(in, out, cyc, "v_add_f32");
  run<FILL_ADD, 3, 1>(in, out, cyc, "v_add_f32");
  run<FILL_ADD, 4, 1>(in, out, cyc, "v_add_f32");
  run<FILL_EXP, 2, 1>(in, out, cyc, "v_exp_f32");
  run<FILL_MFMA4, 1, 1>(in, out, cyc, "mfma 4x4x4");
  run<FILL_MFMA4, 2, 1>(in, out, cyc, "mfma 4x4x4");
  run<FILL_MFMA4, 4, 1>(in, out, cyc, "mfma 4x4x4");
  run<FILL_MFMA4, 2, 0>(in, out, cyc, "mfma 4x4x4");
  run<FILL_MFMA4, 6, 0>(in, out, cyc, "mfma 4x4x4");
  run<FILL_MUL, 6, 0>(in, out, cyc, "v_mul_f32");
  run<FILL_NONE, 0>(in, out, cyc, "none");
  run<FILL_ADD, 4>(in, out, cyc, "v_add_f32");
  run<FILL_ADD, 5>(in, out, cyc, "v_add_f32");
  run<FILL_ADD, 6>(in, out, cyc, "v_add_f32");
  run<FILL_PKMULF32, 4>(in, out, cyc, "v_pk_mul_f32");
  run<FILL_PKMULF32, 8>(in, out, cyc, "v_pk_mul_f32");
  run<FILL_FMA, 8>(in, out, cyc, "v_fma_f32");
  run<FILL_NOP, 6>(in, out, cyc, "s_nop 0");
  run<FILL_NOP, 12>(in, out, cyc, "s_nop 0");
  run<FILL_EXP, 4>(in, out, cyc, "v_exp_f32");
  run<FILL_MAX3, 6>(in, out, cyc, "v_max3_f32");
  run<FILL_PKRTZ, 12>(in, out, cyc, "v_cvt_pkrtz_f16_f32");
  run<FILL_CVT, 6>(in, out, cyc, "v_cvt_pk_f16_f32");
  run<FILL_EXPH, 8>(in, out, cyc, "v_exp_f16");
  run<FILL_CMP, 12>(in, out, cyc, "v_cmp_gt_f32");
  run<FILL_PERM, 12>(in, out, cyc, "v_perm_b32");
  run<FILL_MIX_B128, 0>(in, out, cyc, "4 add + 0 ds");
  run<FILL_MIX_B128, 1>(in, out, cyc, "4 add + b128");
  run<FILL_MIX_B128, 2>(in, out, cyc, "4 add + b128");
  run<FILL_MIX_TR, 1>(in, out, cyc, "4 add + tr_b64");
  run<FILL_MIX_TR, 2>(in, out, cyc, "4 add + tr_b64");
  run<FILL_MIX_TR, 3>(in, out, cyc, "4 add + tr_b64");
  run<FILL_MIX_TR, 4>(in, out, cyc, "4 add + tr_b64");
  return 0;
}
