// Scratch test: what a typed buffer load with a 16-bit destination (tbuffer_load_format_d16_xyzw on
// 32_32_32_32 FLOAT data) does to fp32 values on their way into registers: which rounding, what happens
// to values beyond the fp16 range, to subnormals and NaN; and how fast it streams next to plain 128-bit loads.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t srd_t __attribute__((ext_vector_type(4)));

__device__ inline srd_t make_srd(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

__global__ void convert(const float* in, uint16_t* out, uint32_t n4) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  srd_t srd = make_srd(in, n4 * 16u);
  asm volatile("s_nop 4" : "+s"(srd));
  u32x2 h;
  const uint32_t voff = i * 16u;
  asm volatile(
      "tbuffer_load_format_d16_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(h)
      : "v"(voff), "s"(srd)
      : "memory");
  ((u32x2*)out)[i] = h;
}

// streaming rate: every lane reads `reps` chunks of 16 bytes, strided over the buffer
template <bool D16>
__global__ void stream(const float* in, uint32_t* sink, uint32_t n4, int reps) {
  srd_t srd = make_srd(in, n4 * 16u);
  asm volatile("s_nop 4" : "+s"(srd));
  uint32_t acc = 0;
  uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (int r = 0; r < reps; r += 4) {
    u32x4 a[4];
    u32x2 b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t voff = ((idx + (uint32_t)k * stride) % n4) * 16u;
      if constexpr (D16) {
        asm volatile("tbuffer_load_format_d16_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_32_32_32_32,BUF_NUM_FORMAT_FLOAT] offen"
                     : "=&v"(b[k]) : "v"(voff), "s"(srd) : "memory");
      } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(a[k]) : "v"(voff), "s"(srd) : "memory");
      }
    }
    if constexpr (D16) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) acc ^= b[k][0] ^ b[k][1];
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) acc ^= a[k][0] ^ a[k][1] ^ a[k][2] ^ a[k][3];
    }
    idx += 4 * stride;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

static uint16_t rne(float f) {
  _Float16 h = (_Float16)f;  // host conversion: round to nearest even
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static uint16_t rtz(float f) {
  uint16_t a = rne(f);
  _Float16 h;
  memcpy(&h, &a, 2);
  if (std::isnan(f) || std::isinf((float)h) || (float)h == f) return a;
  // step one unit towards zero if rounding went away from zero
  if (std::fabs((float)h) > std::fabs(f)) return (uint16_t)(a - 1);
  return a;
}

int main() {
  std::vector<float> x;
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  for (int i = 0; i < (1 << 20); ++i) {  // normal-range values with random mantissas
    uint32_t u = (rnd() & 0x807fffffu) | ((uint32_t)(112 + rnd() % 30) << 23);
    float f;
    memcpy(&f, &u, 4);
    x.push_back(f);
  }
  // exact ties between two fp16 neighbours (13 low mantissa bits = 0x1000), both parities
  for (int i = 0; i < 4096; ++i) {
    uint32_t u = (uint32_t)(127 << 23) | ((uint32_t)i << 13) | 0x1000u;
    float f;
    memcpy(&f, &u, 4);
    x.push_back(f);
    x.push_back(-f);
  }
  const float special[] = {0.f, -0.f, 65504.f, 65519.9f, 65520.f, 70000.f, 1e10f, -1e10f, INFINITY, -INFINITY, NAN,
                           6.1e-5f, 5.96e-8f, 2.98e-8f, 2.99e-8f, 1e-10f, 3.0e-6f, -3.1e-6f, 1.0f, 1.00048828125f};
  for (float f : special) x.push_back(f);
  while (x.size() % 4) x.push_back(0.f);
  const uint32_t n = (uint32_t)x.size(), n4 = n / 4;
  float* din;
  uint16_t* dout;
  (void)hipMalloc(&din, n * 4);
  (void)hipMalloc(&dout, n * 2);
  (void)hipMemcpy(din, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(convert, dim3((n4 + 255) / 256), dim3(256), 0, 0, din, dout, n4);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  std::vector<uint16_t> h(n);
  (void)hipMemcpy(h.data(), dout, n * 2, hipMemcpyDeviceToHost);
  size_t eq_rne = 0, eq_rtz = 0, neither = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const bool a = h[i] == rne(x[i]), b = h[i] == rtz(x[i]);
    eq_rne += a;
    eq_rtz += b;
    if (!a && !b && neither++ < 10) printf("  x=%g (0x%08x): got 0x%04x, rne 0x%04x, rtz 0x%04x\n", x[i], *(uint32_t*)&x[i], h[i], rne(x[i]), rtz(x[i]));
  }
  printf("d16 typed load of %u fp32 values: equal to round-to-nearest-even %zu, to round-toward-zero %zu, to neither %zu\n", n,
         eq_rne, eq_rtz, neither);
  for (size_t i = n - 24; i < n; ++i) printf("  special %g -> 0x%04x (rne 0x%04x)\n", x[i], h[i], rne(x[i]));

  // streaming rate over 256 MiB
  const uint32_t big4 = 1u << 24;
  float* dbig;
  uint32_t* sink;
  (void)hipMalloc(&dbig, (size_t)big4 * 16);
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(dbig, 0x3c, (size_t)big4 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int d16 = 0; d16 < 2; ++d16) {
    const int reps = 64;
    const dim3 grid(2048), block(256);
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      if (d16) hipLaunchKernelGGL(stream<true>, grid, block, 0, 0, dbig, sink, big4, reps);
      else hipLaunchKernelGGL(stream<false>, grid, block, 0, 0, dbig, sink, big4, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it == 2) printf("%s: %.3f ms for %.1f MiB -> %.2f TB/s of fp32 source\n", d16 ? "tbuffer_load_format_d16_xyzw" : "buffer_load_dwordx4         ", ms,
                          2048.0 * 256 * reps * 16 / 1048576.0, 2048.0 * 256 * reps * 16 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
