// Native parity + timing harness for libflash_hip.so (no Python, no torch).
// Calls the library through its C ABI only and checks the result against the CPU oracle
// (oracle/attention_oracle.c), on the reference runner's shapes (reference main.cpp:157-193) plus
// larger ones.  Usage:
//   selftest                 parity on the built-in shape list (fp16 and fp32 entry points)
//   selftest bench B H N d [reps] [kernel_type]   hipEvent timing of one shape, parity on 4 sample heads
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flash_hip.h"

extern "C" {
int oracle_attn_naive(const float* q, const float* k, const float* v, float* o, int64_t B, int64_t H, int64_t N,
                      int64_t d, float scale);
void oracle_f32_to_f16_bits(const float* in, uint16_t* out, int64_t n);
void oracle_f16_bits_to_f32(const uint16_t* in, float* out, int64_t n);
}

#define HIP_CHECK(x)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static inline double urand() {
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return ((g_state >> 11) + 0.5) / 9007199254740992.0;
}
static void fill_normal(std::vector<float>& x) {
  for (size_t i = 0; i + 1 < x.size(); i += 2) {
    const double u1 = urand(), u2 = urand();
    const double rad = std::sqrt(-2.0 * std::log(u1));
    x[i] = (float)(rad * std::cos(6.283185307179586 * u2));
    x[i + 1] = (float)(rad * std::sin(6.283185307179586 * u2));
  }
  if (x.size() & 1) x.back() = (float)(std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()));
}

struct Cmp {
  double max_abs = 0, max_rel = 0;
  size_t bad = 0, nan = 0;
};
// |res - ref| <= atol + rtol*|ref| (reference main.cpp:49), atol 2e-3, rtol 1e-2
static Cmp compare(const float* res, const float* ref, size_t n) {
  Cmp c;
  for (size_t i = 0; i < n; ++i) {
    if (std::isnan(res[i])) {
      c.nan++;
      c.bad++;
      continue;
    }
    const double e = std::fabs((double)res[i] - (double)ref[i]);
    c.max_abs = std::max(c.max_abs, e);
    c.max_rel = std::max(c.max_rel, e / (std::fabs((double)ref[i]) + 1e-2));
    if (e > 2e-3 + 1e-2 * std::fabs((double)ref[i])) c.bad++;
  }
  return c;
}

// spike > 0: some keys are made parallel to some query rows with a large gain, so that the running
// max of those rows jumps by far more than any deferral threshold at chosen tiles (forces the
// online-softmax rescale, which bounded random data never triggers)
// io: 0 = fp16 tensors, 1 = fp32 tensors (flash_fwd_f32), 2 = fp32 tensors with a workspace (flash_fwd_f32_ws)
static bool run_shape(int64_t B, int64_t H, int64_t N, int64_t d, int kt, int io, int spike = 0) {
  const bool fp32_io = io != 0;
  const size_t n = (size_t)(B * H * N * d);
  std::vector<float> q(n), k(n), v(n), qr(n), kr(n), vr(n), ref(n), out(n);
  fill_normal(q);
  fill_normal(k);
  fill_normal(v);
  if (spike > 0) {
    for (int64_t hd = 0; hd < B * H; ++hd) {
      for (int sidx = 0; sidx < spike; ++sidx) {
        const int64_t row = (int64_t)(urand() * N), key = (int64_t)(urand() * N);
        const float gain = 6.0f + 2.0f * (float)sidx;
        const float* qrow = &q[(size_t)(hd * N + row) * d];
        double nrm = 0;
        for (int64_t x = 0; x < d; ++x) nrm += (double)qrow[x] * qrow[x];
        nrm = std::sqrt(nrm) + 1e-6;
        for (int64_t x = 0; x < d; ++x) k[(size_t)(hd * N + key) * d + x] = (float)(gain * qrow[x] / nrm);
      }
    }
  }
  std::vector<uint16_t> qh(n), kh(n), vh(n), oh(n);
  oracle_f32_to_f16_bits(q.data(), qh.data(), n);
  oracle_f32_to_f16_bits(k.data(), kh.data(), n);
  oracle_f32_to_f16_bits(v.data(), vh.data(), n);
  oracle_f16_bits_to_f32(qh.data(), qr.data(), n);
  oracle_f16_bits_to_f32(kh.data(), kr.data(), n);
  oracle_f16_bits_to_f32(vh.data(), vr.data(), n);
  const float scale = 1.0f / std::sqrt((float)d);
  oracle_attn_naive(qr.data(), kr.data(), vr.data(), ref.data(), B, H, N, d, scale);

  const size_t esz = fp32_io ? 4 : 2;
  void *dq, *dk, *dv, *dout;
  HIP_CHECK(hipMalloc(&dq, n * esz));
  HIP_CHECK(hipMalloc(&dk, n * esz));
  HIP_CHECK(hipMalloc(&dv, n * esz));
  HIP_CHECK(hipMalloc(&dout, n * esz));
  if (fp32_io) {
    // fp32 callers hand over unrounded data; the kernel rounds to fp16 itself
    HIP_CHECK(hipMemcpy(dq, q.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice));
  } else {
    HIP_CHECK(hipMemcpy(dq, qh.data(), n * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dk, kh.data(), n * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv, vh.data(), n * 2, hipMemcpyHostToDevice));
  }
  HIP_CHECK(hipMemset(dout, 0xff, n * esz));  // NaN pattern: unwritten outputs show up
  void* ws = nullptr;
  int64_t ws_bytes = 0;
  if (io == 2) {
    ws_bytes = flash_fwd_f32_workspace_bytes(B, H, N, d, kt);
    if (ws_bytes < 0) {
      printf("  workspace query failed: %s\n", flash_hip_last_error());
      return false;
    }
    if (ws_bytes > 0) {
      HIP_CHECK(hipMalloc(&ws, (size_t)ws_bytes));
      HIP_CHECK(hipMemset(ws, 0xff, (size_t)ws_bytes));
      // a workspace that is one byte short must be refused, and nothing may have been launched
      if (flash_fwd_f32_ws(dq, dk, dv, dout, B, H, N, d, scale, kt, ws, ws_bytes - 1, nullptr) !=
          FLASH_HIP_ERR_INVALID_ARGUMENT) {
        printf("  a short workspace was accepted\n");
        return false;
      }
    }
  }
  int rc = io == 2   ? flash_fwd_f32_ws(dq, dk, dv, dout, B, H, N, d, scale, kt, ws, ws_bytes, nullptr)
           : io == 1 ? flash_fwd_f32(dq, dk, dv, dout, B, H, N, d, scale, kt, nullptr)
                     : flash_fwd_f16(dq, dk, dv, dout, B, H, N, d, scale, kt, nullptr);
  bool ok = true;
  if (rc != FLASH_HIP_OK) {
    printf("  launch failed rc=%d: %s\n", rc, flash_hip_last_error());
    ok = false;
  } else {
    HIP_CHECK(hipDeviceSynchronize());
    if (fp32_io) {
      HIP_CHECK(hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost));
    } else {
      HIP_CHECK(hipMemcpy(oh.data(), dout, n * 2, hipMemcpyDeviceToHost));
      oracle_f16_bits_to_f32(oh.data(), out.data(), n);
    }
    Cmp c = compare(out.data(), ref.data(), n);
    ok = c.bad == 0 && c.max_abs <= 2e-3;
    printf("  (%lld,%lld,%lld,%lld) %-18s %s  max_abs=%.3e max_rel=%.3e bad=%zu nan=%zu  %s\n", (long long)B,
           (long long)H, (long long)N, (long long)d, flash_kernel_type_name(kt), io == 2 ? (ws ? "f32+ws" : "f32+ws(0)") : io == 1 ? "f32" : "f16", c.max_abs,
           c.max_rel, c.bad, c.nan, ok ? "PASS" : "FAIL");
    if (!ok) {
      int shown = 0;
      for (size_t i = 0; i < n && shown < 7; ++i) {
        const double e = std::fabs((double)out[i] - (double)ref[i]);
        if (std::isnan(out[i]) || e > 2e-3 + 1e-2 * std::fabs((double)ref[i])) {
          const size_t x = i % d, row = (i / d) % N, hd = i / (d * N);
          printf("    mismatch[head %zu, row %zu, col %zu] ref=%g res=%g\n", hd, row, x, ref[i], out[i]);
          shown++;
        }
      }
    }
  }
  HIP_CHECK(hipFree(dq));
  HIP_CHECK(hipFree(dk));
  HIP_CHECK(hipFree(dv));
  HIP_CHECK(hipFree(dout));
  if (ws) HIP_CHECK(hipFree(ws));
  return ok;
}

// fp32 callers: the in-kernel conversion path against the workspace path, timed back to back
static int bench_f32(int64_t B, int64_t H, int64_t N, int64_t d, int reps) {
  const size_t n = (size_t)(B * H * N * d);
  std::vector<float> x(n);
  void *dq, *dk, *dv, *dout, *ws = nullptr;
  HIP_CHECK(hipMalloc(&dq, n * 4));
  HIP_CHECK(hipMalloc(&dk, n * 4));
  HIP_CHECK(hipMalloc(&dv, n * 4));
  HIP_CHECK(hipMalloc(&dout, n * 4));
  for (void* p : {dq, dk, dv}) {
    fill_normal(x);
    HIP_CHECK(hipMemcpy(p, x.data(), n * 4, hipMemcpyHostToDevice));
  }
  const int kt = d % 4 == 0 ? FLASH_KT_MMA_QREG_ASYNC : FLASH_KT_NAIVE1D;  // the async type needs d % 4 == 0
  const int64_t ws_bytes = flash_fwd_f32_workspace_bytes(B, H, N, d, kt);
  if (ws_bytes > 0) HIP_CHECK(hipMalloc(&ws, (size_t)ws_bytes));
  const float scale = 1.0f / std::sqrt((float)d);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    auto call = [&]() {
      return mode ? flash_fwd_f32_ws(dq, dk, dv, dout, B, H, N, d, scale, kt, ws, ws_bytes, nullptr)
                  : flash_fwd_f32(dq, dk, dv, dout, B, H, N, d, scale, kt, nullptr);
    };
    for (int i = 0; i < 3; ++i) {
      if (call() != FLASH_HIP_OK) {
        printf("launch failed: %s\n", flash_hip_last_error());
        return 1;
      }
    }
    HIP_CHECK(hipDeviceSynchronize());
    // The chip needs a few tenths of a second under a new load to settle its clock: a burst of that length
    // first (it also tells how many calls the timed burst needs to be as long), then the timed one.
    float ms = 0;
    int n_timed = reps;
    for (int burst = 0; burst < 2; ++burst) {
      HIP_CHECK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < n_timed; ++i) call();
      HIP_CHECK(hipEventRecord(e1, nullptr));
      HIP_CHECK(hipEventSynchronize(e1));
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (burst == 0 && ms < 300.f) {
        n_timed = (int)(n_timed * 300.f / (ms > 0.01f ? ms : 0.01f)) + 1;
        burst = -1;  // once more, long enough this time
        continue;
      }
    }
    ms /= (float)n_timed;
    printf("fp32 tensors (%lld,%lld,%lld,%lld) %-22s %.4f ms  %.1f TFLOP/s\n", (long long)B, (long long)H, (long long)N,
           (long long)d, mode ? "flash_fwd_f32_ws" : "flash_fwd_f32", ms, 4.0 * B * H * N * N * d / (ms * 1e-3) / 1e12);
  }
  return 0;
}

static int bench_shape(int64_t B, int64_t H, int64_t N, int64_t d, int reps, int kt) {
  const size_t per_head = (size_t)(N * d);
  const size_t n = (size_t)(B * H) * per_head;
  // random data for every head: generate 8 distinct heads and tile them (keeps host time bounded)
  const int64_t gen_heads = std::min<int64_t>(8, B * H);
  std::vector<float> q(gen_heads * per_head), k(q.size()), v(q.size());
  fill_normal(q);
  fill_normal(k);
  fill_normal(v);
  std::vector<uint16_t> qh(q.size()), kh(q.size()), vh(q.size());
  oracle_f32_to_f16_bits(q.data(), qh.data(), q.size());
  oracle_f32_to_f16_bits(k.data(), kh.data(), q.size());
  oracle_f32_to_f16_bits(v.data(), vh.data(), q.size());
  uint16_t *dq, *dk, *dv, *dout;
  HIP_CHECK(hipMalloc(&dq, n * 2));
  HIP_CHECK(hipMalloc(&dk, n * 2));
  HIP_CHECK(hipMalloc(&dv, n * 2));
  HIP_CHECK(hipMalloc(&dout, n * 2));
  for (int64_t hd = 0; hd < B * H; ++hd) {
    const size_t src = (size_t)(hd % gen_heads) * per_head;
    HIP_CHECK(hipMemcpy(dq + hd * per_head, qh.data() + src, per_head * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dk + hd * per_head, kh.data() + src, per_head * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv + hd * per_head, vh.data() + src, per_head * 2, hipMemcpyHostToDevice));
  }
  const float scale = 1.0f / std::sqrt((float)d);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) {
    int rc = flash_fwd_f16(dq, dk, dv, dout, B, H, N, d, scale, kt, nullptr);
    if (rc) {
      printf("launch failed: %s\n", flash_hip_last_error());
      return 1;
    }
  }
  HIP_CHECK(hipDeviceSynchronize());
  std::vector<float> times;
  for (int i = 0; i < reps; ++i) {
    HIP_CHECK(hipEventRecord(e0, nullptr));
    flash_fwd_f16(dq, dk, dv, dout, B, H, N, d, scale, kt, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    times.push_back(ms);
  }
  std::sort(times.begin(), times.end());
  const double med = times[times.size() / 2], mn = times[0];
  const double flops = 4.0 * B * H * (double)N * (double)N * d;
  {
    // the same launches back to back (no host synchronisation in between): the rate under sustained load
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) flash_fwd_f16(dq, dk, dv, dout, B, H, N, d, scale, kt, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("bench (%lld,%lld,%lld,%lld) %s: %d back-to-back launches %.4f ms each -> %.1f TFLOP/s (sustained)\n",
           (long long)B, (long long)H, (long long)N, (long long)d, flash_kernel_type_name(kt), reps, ms / reps,
           flops / (ms / reps) * 1e-9);
  }
  printf("bench (%lld,%lld,%lld,%lld) %s: median %.4f ms  min %.4f ms  -> %.1f TFLOP/s (median), %.1f (best)\n",
         (long long)B, (long long)H, (long long)N, (long long)d, flash_kernel_type_name(kt), med, mn,
         flops / med * 1e-9, flops / mn * 1e-9);
  // parity on the first and last generated heads as they sit at the end of the tensor
  bool ok = true;
  if (N <= 4096) {
    const int64_t hd = B * H - 1;
    const size_t src = (size_t)(hd % gen_heads) * per_head;
    std::vector<float> qr(per_head), kr(per_head), vr(per_head), ref(per_head), out(per_head);
    oracle_f16_bits_to_f32(qh.data() + src, qr.data(), per_head);
    oracle_f16_bits_to_f32(kh.data() + src, kr.data(), per_head);
    oracle_f16_bits_to_f32(vh.data() + src, vr.data(), per_head);
    oracle_attn_naive(qr.data(), kr.data(), vr.data(), ref.data(), 1, 1, N, d, scale);
    std::vector<uint16_t> oh(per_head);
    HIP_CHECK(hipMemcpy(oh.data(), dout + hd * per_head, per_head * 2, hipMemcpyDeviceToHost));
    oracle_f16_bits_to_f32(oh.data(), out.data(), per_head);
    Cmp c = compare(out.data(), ref.data(), per_head);
    ok = c.bad == 0;
    printf("  parity (last head): max_abs=%.3e max_rel=%.3e bad=%zu nan=%zu %s\n", c.max_abs, c.max_rel, c.bad, c.nan,
           ok ? "PASS" : "FAIL");
  }
  return ok ? 0 : 1;
}

// A/B of the persistent and the one-item-per-workgroup kernels, interleaved launch by launch in one
// process on one device (the library re-reads FLASH_HIP_PERSISTENT when asked to).
static int ab_shape(int64_t B, int64_t H, int64_t N, int64_t d, int reps) {
  const size_t n = (size_t)(B * H * N * d);
  std::vector<float> x(std::min<size_t>(n, (size_t)8 * N * d));
  fill_normal(x);
  std::vector<uint16_t> xh(x.size());
  oracle_f32_to_f16_bits(x.data(), xh.data(), x.size());
  uint16_t *dq, *dk, *dv, *dout;
  HIP_CHECK(hipMalloc(&dq, n * 2));
  HIP_CHECK(hipMalloc(&dk, n * 2));
  HIP_CHECK(hipMalloc(&dv, n * 2));
  HIP_CHECK(hipMalloc(&dout, n * 2));
  for (size_t off = 0; off < n; off += xh.size()) {
    const size_t len = std::min(xh.size(), n - off);
    HIP_CHECK(hipMemcpy(dq + off, xh.data(), len * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dk + off, xh.data() + (xh.size() - len) / 2, len * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv + off, xh.data(), len * 2, hipMemcpyHostToDevice));
  }
  const float scale = 1.0f / std::sqrt((float)d);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  std::vector<float> t[2];
  setenv("FLASH_HIP_PERSISTENT_MAX_N", "1000000000", 1);  // let the switch below decide alone
  for (int i = 0; i < 2 * (reps + 3); ++i) {
    const int which = i & 1;
    setenv("FLASH_HIP_PERSISTENT", which ? "1" : "0", 1);
    flash_hip_reload_switches();
    HIP_CHECK(hipEventRecord(e0, nullptr));
    flash_fwd_f16(dq, dk, dv, dout, B, H, N, d, scale, FLASH_KT_MMA_QREG_ASYNC, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 6) t[which].push_back(ms);
  }
  const double flops = 4.0 * B * H * (double)N * (double)N * d;
  for (int which = 0; which < 2; ++which) {
    std::sort(t[which].begin(), t[which].end());
    const double med = t[which][t[which].size() / 2];
    printf("ab (%lld,%lld,%lld,%lld) %s: median %.4f ms -> %.1f TFLOP/s, best %.1f\n", (long long)B, (long long)H,
           (long long)N, (long long)d, which ? "persistent      " : "item-per-workgroup", med, flops / med * 1e-9,
           flops / t[which][0] * 1e-9);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 6 && std::string(argv[1]) == "ab") {
    return ab_shape(atoll(argv[2]), atoll(argv[3]), atoll(argv[4]), atoll(argv[5]), argc > 6 ? atoi(argv[6]) : 30);
  }
  if (argc >= 6 && std::string(argv[1]) == "benchf32") {
    return bench_f32(atoll(argv[2]), atoll(argv[3]), atoll(argv[4]), atoll(argv[5]), argc > 6 ? atoi(argv[6]) : 20);
  }
  if (argc >= 6 && std::string(argv[1]) == "bench") {
    const int reps = argc > 6 ? atoi(argv[6]) : 20;
    const int kt = argc > 7 ? atoi(argv[7]) : FLASH_KT_MMA_QREG_ASYNC;
    return bench_shape(atoll(argv[2]), atoll(argv[3]), atoll(argv[4]), atoll(argv[5]), reps, kt);
  }
  struct Shape {
    int64_t B, H, N, d;
  };
  // the reference runner's five shapes (main.cpp:157-193), the tiny plumbing config, and shapes that
  // cross tile boundaries of the native kernel (Br = 256, Bc = 64)
  const Shape shapes[] = {{1, 1, 50, 32},  {1, 1, 100, 32},  {1, 1, 64, 50},   {5, 12, 53, 69}, {1, 1, 30, 128},
                          {2, 4, 256, 64}, {2, 3, 300, 128}, {1, 2, 1024, 128}, {3, 2, 577, 64}, {1, 2, 257, 96},
                          {1, 1, 1, 8},    {1, 3, 129, 40},  {2, 2, 511, 120},
                          // whole-tile shapes with item counts that split unevenly over XCDs and workgroups
                          {3, 7, 512, 64}, {5, 13, 1280, 128}, {1, 5, 256, 128},
                          // rows without 16-byte chunks, enough work for the padded route of flash_fwd_f32_ws
                          {2, 4, 1024, 69}, {1, 12, 1000, 50}};
  bool all = true;
  for (const Shape& s : shapes) {
    printf("shape (%lld,%lld,%lld,%lld)\n", (long long)s.B, (long long)s.H, (long long)s.N, (long long)s.d);
    const int kts[] = {FLASH_KT_NAIVE1D, FLASH_KT_MMA_QREG, FLASH_KT_MMA_QREG_F32X4LOAD, FLASH_KT_MMA_QREG_ASYNC};
    for (int kt : kts) {
      if (kt == FLASH_KT_MMA_QREG && s.d % 2) continue;  // reference guards, main.cpp:114-126
      if ((kt == FLASH_KT_MMA_QREG_F32X4LOAD || kt == FLASH_KT_MMA_QREG_ASYNC) && s.d % 4) continue;
      all &= run_shape(s.B, s.H, s.N, s.d, kt, 0);
    }
    all &= run_shape(s.B, s.H, s.N, s.d, FLASH_KT_NAIVE1D, 1);
    all &= run_shape(s.B, s.H, s.N, s.d, FLASH_KT_NAIVE1D, 2);
  }
  printf("forced rescale (spiked keys)\n");
  const Shape spiked[] = {{1, 2, 640, 128}, {2, 2, 1000, 64}, {1, 1, 2048, 128}, {1, 2, 333, 72}, {2, 4, 1024, 68}};
  for (const Shape& s : spiked) {
    all &= run_shape(s.B, s.H, s.N, s.d, FLASH_KT_MMA_QREG_ASYNC, 0, 6);
    all &= run_shape(s.B, s.H, s.N, s.d, FLASH_KT_NAIVE1D, 1, 6);
    all &= run_shape(s.B, s.H, s.N, s.d, FLASH_KT_NAIVE1D, 2, 6);
  }
  printf(all ? "ALL PASS\n" : "SOME FAILED\n");
  return all ? 0 : 1;
}
