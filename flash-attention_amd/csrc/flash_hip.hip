// C-ABI entry points of libflash_hip.so (declared in include/flash_hip.h): argument checking,
// tile selection, launch.  Host-side counterpart of the reference's flash::forward dispatch and
// launchKernel (reference flash/launch.cu:130-183, 185-324) and of KernelParametersFactory
// (reference flash/KernelParameters.hpp:383-422).
#include "../../include/flash_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <cstdlib>

#include "attn_fwd_mfma.hpp"
#include "attn_fwd_mfma16.hpp"
#include "cast_f16.hpp"
#if defined(FLASH_PHASE_STAMPS)
#include <cstdio>
#include <vector>
#endif

// Built-in scratch of flash_fwd_f32_static: room for the three fp16 copies of Q, K and V.  A device
// array, so every GPU that loads this library holds 48 MiB for it until the process ends.  Host code
// reaches it only through the address hipGetSymbolAddress returns (static_scratch_ptr below).
constexpr int64_t kStaticScratchBytes = FLASH_HIP_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

using namespace flash_hip;

int static_scratch_ptr(void** out);
int forward_f32_scratch(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                        float softmax_scale, int kernel_type, void* workspace, int64_t workspace_bytes,
                        void* hip_stream);

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

const char* const kNames[FLASH_KT_COUNT] = {
    "naive1D",   "scalar2D",         "scalar2D_row_tile", "warp_wmma", "block_wmma", "wmma_row_block",
    "block_wmma_async", "mma", "mma_swizzle",       "mma_qreg",  "mma_qreg_f32x4load", "mma_qreg_async"};

struct Plan {
  int tile_d;      // 64 or 128
  bool fast;       // fp16 data with 16-byte row chunks: direct-to-LDS staging
  bool vec;        // fp32 data with 32-byte row chunks: 128-bit loads, converted in registers
  int staging;     // flash_launch_info.staging
  uint32_t lds;
  uint32_t nqt;
};

// Shape / kernel-type checks shared by the launchers and the geometry query.  The error cases are
// the reference's: head dim above the largest compiled tile (launch.cu:270-280), odd head dim for
// mma_qreg (:267-269), head dim not a multiple of 4 for the vector-load and async kernels
// (:284-286, :301-303), unknown kernel type (:317).
// The kernels take the running maximum over the unscaled scores, which orders them like the scaled ones
// only for a positive scale (the reference has no such argument: it always uses 1 / sqrt(d),
// launch.cu:192).
int check_scale(float softmax_scale) {
  if (!(softmax_scale > 0.f) || !std::isfinite(softmax_scale))
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "softmax_scale must be positive and finite (got %g)", (double)softmax_scale);
  return FLASH_HIP_OK;
}

int make_plan(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type, bool fp16_io, Plan* plan) {
  if (kernel_type < 0 || kernel_type >= FLASH_KT_COUNT)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "Unsupported kernel type %d", kernel_type);
  if (B <= 0 || H <= 0 || N <= 0 || d <= 0)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "B, H, N, d must be positive (got %lld, %lld, %lld, %lld)",
                (long long)B, (long long)H, (long long)N, (long long)d);
  if (kernel_type == FLASH_KT_MMA_QREG && d % 2 != 0)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "Head dim must be even for kernel %s", kNames[kernel_type]);
  if ((kernel_type == FLASH_KT_MMA_QREG_F32X4LOAD || kernel_type == FLASH_KT_MMA_QREG_ASYNC) && d % 4 != 0)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "Head dim must be a multiple of 4 for kernel %s",
                kNames[kernel_type]);
  if (d > 128)
    return fail(FLASH_HIP_ERR_UNSUPPORTED_DIM,
                "Unsupported tile size: head dim %lld exceeds the largest compiled tile (128) for kernel type: %s",
                (long long)d, kNames[kernel_type]);
  if (N * d * 4 >= (int64_t)kOob)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "one head (N*d = %lld elements) exceeds the 1 GiB descriptor range",
                (long long)(N * d));
  const int64_t nqt = (N + kBr - 1) / kBr;
  if (B * H * nqt >= (int64_t)1 << 31)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "grid of %lld workgroups is too large", (long long)(B * H * nqt));
  plan->tile_d = d <= 64 ? 64 : 128;
  plan->fast = fp16_io && (d % 8 == 0);
  plan->vec = !fp16_io && (d % 8 == 0);
  plan->staging = plan->fast ? kStageDma : (plan->vec ? kStageVec : kStageElem);
  plan->lds = plan->tile_d == 64 ? Geom<64>::kLdsBytes : Geom<128>::kLdsBytes;
  plan->nqt = (uint32_t)nqt;
  return FLASH_HIP_OK;
}


// Kernel arguments of one call.  The kernels work in the exp2 domain: the softmax scale carries log2(e)
// and is split into an exact power of two for Q (`scale_q`: the 32x32x16 kernel only, see FwdParams)
// and the factor that multiplies the fp32 scores.
FwdParams make_params(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                      float softmax_scale, const Plan& plan, bool scale_q) {
  FwdParams prm{};
  prm.q = q;
  prm.k = k;
  prm.v = v;
  prm.o = o;
  prm.N = (int)N;
  prm.d = (int)d;
  prm.nqt = (int)plan.nqt;
  prm.total_wg = (uint32_t)(B * H * (int64_t)plan.nqt);
  // softmax_scale * log2(e) = 2^e * c with c in [1, 2); e is kept inside [-14, 0] so that Q * 2^e neither
  // overflows fp16 nor needs more than the format's normal range (outside that window c takes the rest)
  const float x = softmax_scale * 1.44269504088896340736f;
  int e = 0;
  if (scale_q && x > 0.f && std::isfinite(x)) {
    (void)std::frexp(x, &e);  // x = m * 2^e, m in [0.5, 1)
    e -= 1;                   // x = (2m) * 2^(e-1)
  }
  e = e < -14 ? -14 : (e > 0 ? 0 : e);
  prm.q_scale = std::ldexp(1.0f, e);
  prm.s_scale = x / prm.q_scale;
  prm.defer_thr = prm.s_scale > 0.f ? kDeferLog2 / prm.s_scale : kDeferLog2;
  return prm;
}

// Per-device facts the launchers need on every call, kept process-wide (any thread may warm a device up
// for any other: a HIP graph captured on one thread after a plain call on another finds every attribute
// set).  Ordinals beyond the table (partitioned nodes expose up to 64 devices; the table holds that many)
// are served without the cache.
constexpr int kMaxDevices = 64;
std::atomic<int> g_cu_count[kMaxDevices];

int current_device(int* dev) {
  hipError_t e = hipGetDevice(dev);
  if (e != hipSuccess) return fail(FLASH_HIP_ERR_RUNTIME, "hipGetDevice failed: %s", hipGetErrorString(e));
  return FLASH_HIP_OK;
}

// Workgroups of the persistent kernel on device `dev`: one per CU (the kernel needs a CU's whole
// register file and most of its LDS), never more than there are items.
int64_t grid_limit();  // FLASH_HIP_GRID (below, with the other switches)
uint32_t persistent_grid(uint32_t items, int dev) {
  int n = (dev >= 0 && dev < kMaxDevices) ? g_cu_count[dev].load(std::memory_order_relaxed) : 0;
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < kMaxDevices) g_cu_count[dev].store(n, std::memory_order_relaxed);
  }
  const int64_t limit = grid_limit();
  if (limit > 0 && limit < n) n = (int)limit;
  const uint32_t g = (uint32_t)n & ~7u;  // a multiple of the 8 XCDs
  return items < g ? items : (g ? g : 8u);
}

// One bit per device ordinal: the kernel's dynamic-LDS attribute has been set there.
struct AttrDone {
  std::atomic<uint64_t> mask{0};
};

// grid == 0: one workgroup per item; kPersistentGrid: one per CU of the current device.
constexpr uint32_t kPersistentGrid = 0xffffffffu;

template <typename Kernel>
int launch_kernel(Kernel kern, const FwdParams& prm, uint32_t lds, hipStream_t stream, AttrDone& attr_done,
                  uint32_t grid = 0, uint32_t threads = kThreads) {
  int dev = 0;
  int rc = current_device(&dev);
  if (rc != FLASH_HIP_OK) return rc;
  if (grid == 0) grid = prm.total_wg;
  if (grid == kPersistentGrid) grid = persistent_grid(prm.total_wg, dev);
  // Above the 64 KiB default a kernel must opt in to its dynamic LDS size (the reference does the
  // same above 48 KiB, launch.cu:134-148).
  const bool cached = dev >= 0 && dev < kMaxDevices;
  if (!cached || !((attr_done.mask.load(std::memory_order_acquire) >> dev) & 1u)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
      return fail(FLASH_HIP_ERR_LDS, "Failed to set dynamic shared memory size %u: %s", lds, hipGetErrorString(e));
    if (cached) attr_done.mask.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
#if defined(FLASH_PHASE_STAMPS)
  // diagnostic build: collect and print the per-wave phase stamps of this launch (synchronises)
  {
    static unsigned long long* buf = nullptr;
    static size_t cap = 0;
    const size_t n = (size_t)grid * (threads / 64) * 8;
    if (n > cap) {
      (void)hipMalloc(&buf, n * 8);
      cap = n;
    }
    (void)hipMemsetAsync(buf, 0, n * 8, stream);
    FwdParams prm2 = prm;
    prm2.stamps = buf;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, prm2);
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost);
    double ph[4] = {0, 0, 0, 0}, cold = 0, items = 0;
    size_t cnt = 0;
    for (size_t i = 0; i < n; i += 8) {
      if (h[i + 4] == 0) continue;  // kernel without stamps
      for (int k = 0; k < 4; ++k) ph[k] += (double)(h[i + k + 1] - h[i + k]);
      cold += (double)h[i + 5];   // persistent kernel only: cycles before the first item's S(0)
      items += (double)h[i + 6];  // persistent kernel only: items the workgroup went through
      ++cnt;
    }
    if (cnt)
      fprintf(stderr, "[stamps] s_memtime ticks per wave: prologue %.0f  loop %.0f  tail+epilogue issue %.0f  store drain %.0f  "
                      "cold start %.0f  items %.2f  (%zu waves)\n",
              ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, cold / cnt, items / cnt, cnt);
  }
#else
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, prm);
#endif
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FLASH_HIP_ERR_RUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
  return FLASH_HIP_OK;
}

template <int D, typename T, bool FAST>
int launch(const FwdParams& prm, uint32_t lds, hipStream_t stream) {
  static AttrDone attr_done;
  return launch_kernel(attn_fwd_kernel<D, T, FAST>, prm, lds, stream, attr_done);
}

// fp16 rows of a multiple of 8 elements: the 16x16x32-MFMA kernel; TO = element type of O
template <int D, typename TO>
int launch_m16(const FwdParams& prm, uint32_t lds, hipStream_t stream) {
  static AttrDone attr_done;
  return launch_kernel(m16::attn_fwd_kernel_m16<D, TO>, prm, lds, stream, attr_done);
}

// Longest sequence on which fp32 callers that bring a workspace still take the register-staged kernel.
// The conversion pass costs one more trip of K and V through HBM plus a launch that cannot overlap the
// attention kernel; converting in the kernel costs every workgroup that reads a K/V tile 32 conversions
// and twice the L2 -> LDS bytes.  Few readers per tile (N / 256 of them) favour the latter.  Measured
// (tests/native/selftest benchf32, TFLOP/s in-kernel / pass): (4,96,N,128) 478 / 385 at N = 512, 680 / 617 at
// 1024, 860 / 850 at 2048, 965 / 1020 at 4096; (16,32,N,64) with 64 query rows per wave 483 / 349 at 512,
// 518 / 580 at 1024.  Whole 512-row tiles of head dims up to 64 run 128 rows per wave in both routes, and
// there the in-kernel route wins at every length (755 / 596 at N = 1024, 935 / 795 at 2048, 1050 / 952 at 4096,
// 1055 / 1017 at 8192, 1102 / 1096 at 16384): see cast_copies.
constexpr int64_t kVecDefaultMaxN128 = 2048;
constexpr int64_t kVecDefaultMaxN64 = 768;

// Routing switches for A/B measurements, from the environment:
//   FLASH_HIP_PERSISTENT=0         whole-tile shapes go to the one-item-per-workgroup kernel as well
//   FLASH_HIP_PERSISTENT_MAX_N=<n> longest sequence routed to the persistent kernel
//   FLASH_HIP_WIDE_MAX_N=<n>       longest sequence that gets the eight-wave layout (0 = never)
//   FLASH_HIP_TALL=0               keep head dims up to 64 on 256-row query tiles
//   FLASH_HIP_VEC_MAX_N=<n>        longest sequence on which fp32 callers with a workspace still take the
//                                  register-staged kernel instead of the conversion pass (0 = never)
//   FLASH_HIP_GRID=<n>             at most n persistent workgroups (a multiple of 8; default: one per CU):
//                                  long item chains per workgroup on small inputs, for tests
// They are read once per process, on the first call that routes anything, and again only when
// flash_hip_reload_switches() is called: the hot path does no environment lookups, and a warmed-up HIP graph
// capture stays on the kernels it was warmed up with.
struct Switches {
  bool persistent;
  int64_t persistent_max_n;
  int64_t wide_max_n;
  bool tall;
  int64_t vec_max_n;  // < 0: the per-head-dim defaults
  int64_t grid;       // <= 0: one persistent workgroup per CU
};
Switches read_switches() {
  auto flag = [](const char* name) {
    const char* e = getenv(name);
    return !(e && e[0] == '0');
  };
  auto num = [](const char* name, int64_t dflt) {
    const char* e = getenv(name);
    return e ? (int64_t)atoll(e) : dflt;
  };
  Switches s;
  s.persistent = flag("FLASH_HIP_PERSISTENT");
  s.persistent_max_n = num("FLASH_HIP_PERSISTENT_MAX_N", (int64_t)1 << 40);
  s.wide_max_n = num("FLASH_HIP_WIDE_MAX_N", (int64_t)1 << 40);
  s.tall = flag("FLASH_HIP_TALL");
  s.vec_max_n = num("FLASH_HIP_VEC_MAX_N", -1);
  s.grid = num("FLASH_HIP_GRID", 0);
  return s;
}
// The current set: published through one pointer, so a reload is safe against calls in flight on other
// threads (they route by the old set or the new one, never by a mixture).  Old sets are never freed.
std::atomic<const Switches*> g_switches{nullptr};
const Switches* publish_switches() {
  const Switches* fresh = new Switches(read_switches());
  g_switches.store(fresh, std::memory_order_release);
  return fresh;
}
Switches switches() {
  const Switches* s = g_switches.load(std::memory_order_acquire);
  if (!s) s = publish_switches();  // racing first calls read the same environment
  return *s;
}
int64_t grid_limit() { return switches().grid; }

// Whole-tile d = 128 shapes run the persistent kernel in the eight-wave layout (two waves per SIMD, 32
// query rows each).  The two waves of a SIMD fill each other's matrix-pipe gaps: at the item boundaries,
// which is what short items are made of (+9 % at N = 512, +3 % at 1024), and inside the steps, where one
// wave alone is issue-bound.  On long sequences the saved cycles (10 %) come back as a lower clock on
// boxes that sit at the power cap (same time) and as time on the others: measured +0.2 %, +0.9 %, +1.3 %,
// +4.2 % at (8,96,4096,128) on four boxes, +4 % at N = 8192 and 16384 on the slowest, never slower than
// within noise.  d = 64 gains nothing from it at any length.
bool use_wide(const Switches& sw, int tile_d, int64_t N) { return tile_d == 128 && N <= sw.wide_max_n; }

// d = 64 leaves room for 128 query rows per wave (O^T and Q take the registers they take at d = 128 with
// 64 rows): a 512-row query tile per workgroup, i.e. half the K/V staging traffic and half the LDS
// fragment reads per FLOP.  Whole 512-row tiles only.
bool use_tall(const Switches& sw, int tile_d, int64_t N) { return tile_d == 64 && N % m16x::kBr == 0 && sw.tall; }

// fp32 Q, K, V, O with rows of a multiple of 8 elements: the same kernels with register staging (128-bit
// fp32 loads, rounded to fp16 on their way into LDS), four waves x 64 query rows.  Whole query tiles run the
// persistent kernel, ragged N one work item per workgroup.
bool vec_tall(int tile_d, int64_t N);  // (below, with the other routing decisions)
template <int D>
int launch_m16_vec(const FwdParams& prm, uint32_t lds, bool persistent, hipStream_t stream) {
  if constexpr (D == 64) {
    if (persistent && vec_tall(D, prm.N)) {
      // d <= 64, whole 512-row query tiles: 128 query rows per wave, so that every fp32 K/V tile is fetched
      // by half as many workgroups (at d = 64 the in-kernel route is bound by those re-reads)
      FwdParams tall = prm;
      tall.nqt = (int)(prm.N / m16x::kBr);
      tall.total_wg = prm.total_wg / (uint32_t)prm.nqt * (uint32_t)tall.nqt;
      static AttrDone attr_done_t;
      return launch_kernel(m16x::attn_fwd_kernel_m16p<64, float, true>, tall, (uint32_t)m16x::CfgP<64>::kLdsBytes,
                           stream, attr_done_t, kPersistentGrid, (uint32_t)m16x::kThreads);
    }
  }
  if (persistent) {
    // (four waves x 64 query rows also at d = 128: the eight-wave layout measured 10-20 % slower here while
    // the tiles were converted on the vector ALU, and with the converting loads +0.7 % at N = 1024, +0.5 % at
    // 2048 and -5 % at 512)
    static AttrDone attr_done_p;
    return launch_kernel(m16::attn_fwd_kernel_m16p<D, float, true>, prm, (uint32_t)m16::CfgP<D>::kLdsBytes, stream,
                         attr_done_p, kPersistentGrid, (uint32_t)m16::kThreads);
  }
  static AttrDone attr_done;
  return launch_kernel(m16::attn_fwd_kernel_m16<D, float, true>, prm, lds, stream, attr_done);
}

// the eight-wave layout's name for an output type (the bfloat16 tag is declared per layout)
template <typename TO>
struct WideOut {
  using type = TO;
};
template <>
struct WideOut<m16::bf16_out> {
  using type = m16w::bf16_out;
};

template <typename TO>
struct TallOut {
  using type = TO;
};
template <>
struct TallOut<m16::bf16_out> {
  using type = m16x::bf16_out;
};

// whole query tiles (N % 256 == 0): persistent workgroups with uninterrupted K/V staging
template <int D, typename TO>
int launch_m16p(const FwdParams& prm, hipStream_t stream) {
  static AttrDone attr_done;
  return launch_kernel(m16::attn_fwd_kernel_m16p<D, TO>, prm, (uint32_t)m16::CfgP<D>::kLdsBytes, stream, attr_done,
                       kPersistentGrid, (uint32_t)m16::kThreads);
}
template <typename TO>
int launch_m16p_wide(const FwdParams& prm, hipStream_t stream) {
  static AttrDone attr_done;
  static_assert(m16w::CfgP<128>::kLdsBytes == m16::CfgP<128>::kLdsBytes, "both layouts use the same LDS rings");
  return launch_kernel(m16w::attn_fwd_kernel_m16p<128, typename WideOut<TO>::type>, prm,
                       (uint32_t)m16w::CfgP<128>::kLdsBytes, stream, attr_done, kPersistentGrid,
                       (uint32_t)m16w::kThreads);
}

// Which of the 16x16x32 kernels a whole-row-chunk shape goes to.  Decided once per call and handed down,
// so that the conversion pass and the kernel launch of one call agree.
enum Route : int {
  kRouteItem = 0,     // one work item per workgroup (ragged N)
  kRoutePersistent,   // persistent workgroups, four waves x 64 query rows
  kRouteWide,         // persistent, eight waves x 32 query rows
  kRouteTall          // persistent, four waves x 128 query rows (512-row query tiles)
};
// fp32 callers (the fp16 kernels with fp32 output): beyond N = 512 they stay on the four-wave layout, whose
// kernels read Q as fp32 themselves, which is worth more than the eight-wave layout (measured at
// (4,96,N,128): N = 1024 +5..12 %, N = 512 -1 %)
Route pick_route(const Plan& plan, int64_t N, bool fp32_out) {
  const Switches sw = switches();
  if (!(N % kBr == 0 && N <= sw.persistent_max_n && sw.persistent)) return kRouteItem;
  if (use_wide(sw, plan.tile_d, N)) return (fp32_out && N > 512) ? kRoutePersistent : kRouteWide;
  return use_tall(sw, plan.tile_d, N) ? kRouteTall : kRoutePersistent;
}

// fp32 tensors converted inside the attention kernel: persistent workgroups for whole query tiles
bool vec_persistent(int64_t N) {
  const Switches sw = switches();
  return N % kBr == 0 && N <= sw.persistent_max_n && sw.persistent;
}
// ... in the 128-rows-per-wave layout for whole 512-row tiles of head dims up to 64
bool vec_tall(int tile_d, int64_t N) { return vec_persistent(N) && use_tall(switches(), tile_d, N); }

// fp16 Q, K, V with 16-byte row chunks, O as fp16 or fp32
template <typename TO>
int launch_fast(const FwdParams& prm, const Plan& plan, Route route, hipStream_t stream) {
  static_assert(m16::Cfg<128>::kLdsBytes == Geom<128>::kLdsBytes && m16::Cfg<64>::kLdsBytes == Geom<64>::kLdsBytes,
                "both kernel families use the same LDS rings");
  switch (route) {
    case kRouteWide:
      return launch_m16p_wide<TO>(prm, stream);
    case kRouteTall: {
      FwdParams tall = prm;  // the same heads cut into 512-row query tiles
      tall.nqt = (int)(prm.N / m16x::kBr);
      tall.total_wg = prm.total_wg / (uint32_t)prm.nqt * (uint32_t)tall.nqt;
      static AttrDone attr_done;
      return launch_kernel(m16x::attn_fwd_kernel_m16p<64, typename TallOut<TO>::type>, tall,
                           (uint32_t)m16x::CfgP<64>::kLdsBytes, stream, attr_done, kPersistentGrid,
                           (uint32_t)m16x::kThreads);
    }
    case kRoutePersistent:
      // Measured launch by launch against the one-item-per-workgroup kernel (selftest ab), B=4 H=96 d=128:
      // +9 % at N = 512, +8 % at 1024, +4 % at 2048, +2 % at 4096, even at 8192; +5 % at (16,32,2048,64).
      return plan.tile_d == 64 ? launch_m16p<64, TO>(prm, stream) : launch_m16p<128, TO>(prm, stream);
    default:
      return plan.tile_d == 64 ? launch_m16<64, TO>(prm, plan.lds, stream) : launch_m16<128, TO>(prm, plan.lds, stream);
  }
}

// The fast kernels move 16 bytes per lane (128-bit buffer loads and stores, global->LDS DMA): a tensor
// whose base is not 16-byte aligned (a view that starts inside an allocation) goes to the
// element-staging kernel instead, which only needs element alignment.
bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15u) == 0;
}

// Bytes of one fp16 copy of a tensor in the workspace of flash_fwd_f32_ws (256-byte granules)
int64_t ws_tensor_bytes(int64_t B, int64_t H, int64_t N, int64_t d) { return (B * H * N * d * 2 + 255) / 256 * 256; }

template <typename T>
int forward(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
            float softmax_scale, int kernel_type, void* hip_stream) {
  Plan plan;
  int rc = make_plan(B, H, N, d, kernel_type, sizeof(T) == 2, &plan);
  if (rc != FLASH_HIP_OK) return rc;
  if ((rc = check_scale(softmax_scale)) != FLASH_HIP_OK) return rc;
  if (!q || !k || !v || !o) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "null tensor pointer");
  hipStream_t stream = (hipStream_t)hip_stream;
  if constexpr (sizeof(T) == 2) {
    if (plan.fast && aligned16(q, k, v, o)) {
#if defined(FLASH_MFMA32)  // comparison builds: the 32x32x16 kernel on the fast path as well
      const FwdParams prm = make_params(q, k, v, o, B, H, N, d, softmax_scale, plan, true);
      return plan.tile_d == 64 ? launch<64, T, true>(prm, plan.lds, stream) : launch<128, T, true>(prm, plan.lds, stream);
#else
      return launch_fast<f16>(make_params(q, k, v, o, B, H, N, d, softmax_scale, plan, false), plan,
                              pick_route(plan, N, false), stream);
#endif
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (plan.vec && aligned16(q, k, v, o)) {
      const FwdParams prm = make_params(q, k, v, o, B, H, N, d, softmax_scale, plan, false);
      const bool persistent = vec_persistent(N);
      return plan.tile_d == 64 ? launch_m16_vec<64>(prm, plan.lds, persistent, stream)
                               : launch_m16_vec<128>(prm, plan.lds, persistent, stream);
    }
  }
  const FwdParams prm = make_params(q, k, v, o, B, H, N, d, softmax_scale, plan, true);
  return plan.tile_d == 64 ? launch<64, T, false>(prm, plan.lds, stream) : launch<128, T, false>(prm, plan.lds, stream);
}

// Head dims that are not a multiple of 8, fp32 callers with a workspace: the rows are copied into rows of
// the next multiple of 8 elements (zero-filled) and the 16x16x32 kernels run on the copies (cast_f16.hpp:
// pad_rows_kernel, unpad_rows_kernel).  `plan` is the plan of the padded shape.
struct PadPlan {
  int64_t d_pad;
  Plan plan;
  Route route;
  bool q_f32;       // the route's kernel reads Q as fp32 (four-wave layouts), else as fp16
  int64_t each;     // bytes of one padded fp16 tensor (256-byte granules)
  int64_t need;     // workspace bytes: K, V (fp16), Q (room for fp32), O (fp32)
};
// Two more launches and a trip of every tensor through HBM: worth it from a few GFLOP of attention on
// (below that a call is launch overhead either way, and the one-kernel element-staging route stays).
// Measured at 2.3-2.8 GFLOP: (1,8,1024,69) 34 against 72 us, (1,32,512,69) 29 / 44, (1,2,4096,100) 95 / 252,
// (4,24,384,50) 27.6 / 28.9 -- many short heads gain least.
constexpr double kPadMinFlops = 2e9;
bool make_pad_plan(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type, PadPlan* pp) {
  if (d % 8 == 0) return false;
  if (4.0 * (double)B * (double)H * (double)N * (double)N * (double)d < kPadMinFlops) return false;
  pp->d_pad = (d + 7) / 8 * 8;
  char keep[sizeof(g_err)];
  memcpy(keep, g_err, sizeof(g_err));  // a shape the padded plan rejects is no error: it stays on element staging
  const int rc = make_plan(B, H, N, pp->d_pad, kernel_type, true, &pp->plan);
  memcpy(g_err, keep, sizeof(g_err));
  if (rc != FLASH_HIP_OK || !pp->plan.fast) return false;
  // routed like fp16 tensors: Q is copied anyway, and an fp16 copy for the eight-wave layout is cheaper than
  // an fp32 one for the four-wave layout that pick_route prefers for fp32 callers beyond N = 512
  pp->route = pick_route(pp->plan, N, false);
  pp->q_f32 = pp->route == kRouteItem || pp->route == kRoutePersistent;
  pp->each = ws_tensor_bytes(B, H, N, pp->d_pad);
  pp->need = 6 * pp->each;  // K, V: one each; Q: two (fp32); O: two (fp32)
  return true;
}

// fp16 copies the conversion pass of flash_fwd_f32_ws writes for this shape: none when the call converts
// inside the attention kernel (rows without 16-byte chunks, sequences up to the register-staging limit),
// K and V, or Q as well for the wave layouts that cannot read it as fp32.
int cast_copies(const Plan& plan, int64_t N) {
  const int64_t vec_max_n = switches().vec_max_n;
  if (!plan.fast || N <= (vec_max_n >= 0 ? vec_max_n : (plan.tile_d == 64 ? kVecDefaultMaxN64 : kVecDefaultMaxN128))) return 0;
  // head dims up to 64 in whole 512-row query tiles: the 128-rows-per-wave kernel converts in place at every
  // length (unless the switch says otherwise)
  if (vec_max_n < 0 && vec_tall(plan.tile_d, N)) return 0;
  const Route route = pick_route(plan, N, true);
  return (route == kRouteTall || route == kRouteWide) ? 3 : 2;
}

// fp32 tensors through scratch memory: one conversion pass into three fp16 copies, then the fp16
// kernels with fp32 output.  `workspace` is device memory the caller (or the library's built-in
// buffer) provides; nothing is allocated here.
int forward_f32_scratch(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                        float softmax_scale, int kernel_type, void* workspace, int64_t workspace_bytes,
                        void* hip_stream) {
  Plan plan;
  int rc = make_plan(B, H, N, d, kernel_type, true, &plan);
  if (rc != FLASH_HIP_OK) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  PadPlan pp;
  if (!plan.fast && workspace && make_pad_plan(B, H, N, d, kernel_type, &pp)) {
    // rows without 16-byte chunks: pad into the workspace, run the fast kernels there, copy O out
    // (element loads and stores on the caller's side: any alignment of q, k, v, o will do)
    if ((rc = check_scale(softmax_scale)) != FLASH_HIP_OK) return rc;
    if (!q || !k || !v || !o) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "null tensor pointer");
    if (workspace_bytes < pp.need)
      return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "workspace of %lld bytes is too small: %lld needed",
                  (long long)workspace_bytes, (long long)pp.need);
    if ((uintptr_t)workspace % 16 != 0) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "workspace must be 16-byte aligned");
    char* ws = (char*)workspace;
    PadParams pad{};
    pad.src[0] = (const float*)k;
    pad.src[1] = (const float*)v;
    pad.src[2] = (const float*)q;
    pad.dst[0] = ws;
    pad.dst[1] = ws + pp.each;
    pad.dst[2] = ws + 2 * pp.each;
    float* o_pad = (float*)(ws + 4 * pp.each);
    pad.rows = (uint64_t)(B * H * N);
    pad.d = (uint32_t)d;
    pad.d_pad = (uint32_t)pp.d_pad;
    pad.f32_mask = pp.q_f32 ? 4u : 0u;
    const uint64_t blocks = (pad.rows + kPadRowsPerBlock - 1) / kPadRowsPerBlock;
    const uint32_t gx = (uint32_t)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(gx, kCastTensors), dim3(kCastThreads), 0, stream, pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FLASH_HIP_ERR_RUNTIME, "padding kernel launch failed: %s", hipGetErrorString(e));
    const FwdParams prm =
        make_params(pad.dst[2], pad.dst[0], pad.dst[1], o_pad, B, H, N, pp.d_pad, softmax_scale, pp.plan, false);
    if ((rc = launch_fast<float>(prm, pp.plan, pp.route, stream)) != FLASH_HIP_OK) return rc;
    UnpadParams up{o_pad, (float*)o, pad.rows, pad.d, pad.d_pad};
    const uint64_t groups = (pad.rows + 15) / 16;  // 16 rows per workgroup and trip
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((uint32_t)(groups < 8192 ? groups : 8192)), dim3(kCastThreads), 0, stream,
                       up);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(FLASH_HIP_ERR_RUNTIME, "output copy kernel launch failed: %s", hipGetErrorString(e));
    return FLASH_HIP_OK;
  }
  // a sequence short enough for the register-staged kernel, no workspace, or tensors that do not start on a
  // 16-byte boundary: convert while staging
  const int n_cast = cast_copies(plan, N);
  if (n_cast == 0 || !workspace || !aligned16(q, k, v, o))
    return forward<float>(q, k, v, o, B, H, N, d, softmax_scale, kernel_type, hip_stream);
  if ((rc = check_scale(softmax_scale)) != FLASH_HIP_OK) return rc;
  if (!q || !k || !v || !o) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "null tensor pointer");
  const int64_t each = ws_tensor_bytes(B, H, N, d);
  if (workspace_bytes < n_cast * each)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "workspace of %lld bytes is too small: %lld needed",
                (long long)workspace_bytes, (long long)(n_cast * each));
  if ((uintptr_t)workspace % 16 != 0) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "workspace must be 16-byte aligned");
  // K and V, which every query tile of a head re-reads; Q is read by one workgroup per row and the
  // attention kernels take it from the caller's fp32 tensor (m16::kQFromF32), except in the
  // 128-rows-per-wave and eight-wave layouts.  The workspace keeps its three-slot layout (K, V, then Q if converted).
  const Route route = pick_route(plan, N, true);
  const bool cast_q = n_cast == 3;
  CastParams cp{};
  const void* src[kCastTensors] = {k, v, q};
  for (int i = 0; i < n_cast; ++i) {
    cp.src[i] = (const float*)src[i];
    cp.dst[i] = (f16*)((char*)workspace + (int64_t)i * each);
  }
  cp.n4 = (uint64_t)(B * H * N * d / 4);
  // enough workgroups to keep every CU's memory pipeline full, a few trips each at large sizes
  const uint64_t trips = (cp.n4 + kCastThreads * 4 - 1) / (kCastThreads * 4);
  const uint32_t gx = (uint32_t)(trips < 8192 ? (trips ? trips : 1) : 8192);
  hipLaunchKernelGGL(cast_f32_to_f16_kernel, dim3(gx, n_cast), dim3(kCastThreads), 0, stream, cp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FLASH_HIP_ERR_RUNTIME, "conversion kernel launch failed: %s", hipGetErrorString(e));
  const FwdParams prm =
      make_params(cast_q ? (const void*)cp.dst[2] : q, cp.dst[0], cp.dst[1], o, B, H, N, d, softmax_scale, plan, false);
  return launch_fast<float>(prm, plan, route, stream);
}

// Address of this device's copy of the built-in scratch array.  Every GPU the library runs on holds
// its own copy at its own address; the array's name is only a host-side handle for
// hipGetSymbolAddress and is never used as a pointer.  Resolved once per device, with that device
// current, on the first call that needs it there.
std::atomic<void*> g_scratch_ptr[kMaxDevices];

int static_scratch_ptr(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(FLASH_HIP_ERR_RUNTIME, "hipGetDevice failed: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kMaxDevices)
    return fail(FLASH_HIP_ERR_RUNTIME, "device ordinal %d is outside the %d the built-in scratch supports", dev, kMaxDevices);
  void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!ptr) {
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
      return fail(FLASH_HIP_ERR_RUNTIME, "hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev].store(ptr, std::memory_order_release);  // racing first calls resolve the same address
  }
  *out = ptr;
  return FLASH_HIP_OK;
}

}  // namespace

extern "C" {

int flash_fwd_f16(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                  float softmax_scale, int kernel_type, void* hip_stream) {
  return forward<f16>(q, k, v, o, B, H, N, d, softmax_scale, kernel_type, hip_stream);
}

int flash_fwd_f32(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                  float softmax_scale, int kernel_type, void* hip_stream) {
  return forward<float>(q, k, v, o, B, H, N, d, softmax_scale, kernel_type, hip_stream);
}

int flash_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                   float softmax_scale, int kernel_type, void* hip_stream) {
  Plan plan;
  int rc = make_plan(B, H, N, d, kernel_type, true, &plan);
  if (rc != FLASH_HIP_OK) return rc;
  if (!plan.fast)
    return fail(FLASH_HIP_ERR_UNSUPPORTED_DIM, "bfloat16 tensors need a head dim that is a multiple of 8 (got %lld)",
                (long long)d);
  if ((rc = check_scale(softmax_scale)) != FLASH_HIP_OK) return rc;
  if (!q || !k || !v || !o) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "null tensor pointer");
  if (!aligned16(q, k, v, o))
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "bfloat16 tensors must be 16-byte aligned");
  const FwdParams prm = make_params(q, k, v, o, B, H, N, d, softmax_scale, plan, false);
  return launch_fast<m16::bf16_out>(prm, plan, pick_route(plan, N, false), (hipStream_t)hip_stream);
}

int64_t flash_fwd_f32_workspace_bytes(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type) {
  Plan plan;
  if (make_plan(B, H, N, d, kernel_type, true, &plan) != FLASH_HIP_OK) return -1;
  PadPlan pp;
  if (!plan.fast) return make_pad_plan(B, H, N, d, kernel_type, &pp) ? pp.need : 0;
  return (int64_t)cast_copies(plan, N) * ws_tensor_bytes(B, H, N, d);
}

int flash_fwd_f32_ws(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N, int64_t d,
                     float softmax_scale, int kernel_type, void* workspace, int64_t workspace_bytes, void* hip_stream) {
  return forward_f32_scratch(q, k, v, o, B, H, N, d, softmax_scale, kernel_type, workspace, workspace_bytes, hip_stream);
}

int flash_fwd_f32_static(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t N,
                         int64_t d, float softmax_scale, int kernel_type, void* hip_stream) {
  Plan plan;
  int rc = make_plan(B, H, N, d, kernel_type, true, &plan);
  if (rc != FLASH_HIP_OK) return rc;
  // what this call will really put into the buffer: nothing when it converts inside the attention kernel
  // (tensors off a 16-byte boundary included), else the two or three copies of the conversion pass
  // (or the padded copies of rows without 16-byte chunks, whatever the alignment)
  PadPlan pp;
  int64_t need = !plan.fast ? (make_pad_plan(B, H, N, d, kernel_type, &pp) ? pp.need : 0)
                            : (aligned16(q, k, v, o) ? (int64_t)cast_copies(plan, N) * ws_tensor_bytes(B, H, N, d) : 0);
  // such rows have a kernel that needs no scratch (element staging): shapes too large to pad here take it
  if (!plan.fast && need > kStaticScratchBytes) need = 0;
  if (need > kStaticScratchBytes)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT,
                "flash_fwd_f32_static: this shape needs %lld bytes of scratch, the built-in buffer holds %lld; "
                "use flash_fwd_f32_ws with caller memory",
                (long long)need, (long long)kStaticScratchBytes);
  void* ws = nullptr;
  if (need > 0) {
    rc = static_scratch_ptr(&ws);
    if (rc != FLASH_HIP_OK) return rc;
  }
  return forward_f32_scratch(q, k, v, o, B, H, N, d, softmax_scale, kernel_type, ws, need, hip_stream);
}

int flash_fwd_query(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type, flash_launch_info* out) {
  return flash_fwd_query_entry(B, H, N, d, kernel_type, FLASH_ENTRY_F16, out);
}

int flash_fwd_query_entry(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type, int entry,
                          flash_launch_info* out) {
  if (!out) return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "null flash_launch_info");
  if (entry < FLASH_ENTRY_F16 || entry > FLASH_ENTRY_F32_WS)
    return fail(FLASH_HIP_ERR_INVALID_ARGUMENT, "unknown entry point selector %d", entry);
  Plan plan;
  // fp32 callers whose call converts inside the attention kernel (no workspace, or a sequence up to the
  // register-staging limit) are described as flash_fwd_f32 launches them
  bool f32_in_kernel = entry == FLASH_ENTRY_F32;
  PadPlan pp;
  bool padded = false;
  if (entry == FLASH_ENTRY_F32_WS) {
    Plan probe;
    int prc = make_plan(B, H, N, d, kernel_type, true, &probe);
    if (prc != FLASH_HIP_OK) return prc;
    // rows without 16-byte chunks are padded into the workspace and run the fp16 kernels there
    padded = !probe.fast && make_pad_plan(B, H, N, d, kernel_type, &pp);
    f32_in_kernel = !padded && cast_copies(probe, N) == 0;
  }
  int rc = make_plan(B, H, N, d, kernel_type, !f32_in_kernel, &plan);
  if (rc != FLASH_HIP_OK) return rc;
  if (padded) plan = pp.plan;
  if (entry == FLASH_ENTRY_BF16 && !plan.fast)
    return fail(FLASH_HIP_ERR_UNSUPPORTED_DIM, "bfloat16 tensors need a head dim that is a multiple of 8 (got %lld)",
                (long long)d);
  out->tile_kv = kBc;
  out->tile_q = kBr;  // 512 for the 128-rows-per-wave layout, set below
  out->tile_d = (uint32_t)plan.tile_d;
  // whole-tile fp16 shapes run on the persistent kernel with four-slot rings
  const Route route = padded      ? pp.route
                      : plan.fast ? pick_route(plan, N, entry == FLASH_ENTRY_F32_WS)
                                : ((plan.vec && vec_persistent(N)) ? (vec_tall(plan.tile_d, N) ? kRouteTall : kRoutePersistent)
                                                                   : kRouteItem);
  const bool persistent = route != kRouteItem;
  out->lds_bytes = !persistent ? plan.lds
                               : (uint32_t)(plan.tile_d == 64 ? m16::CfgP<64>::kLdsBytes : m16::CfgP<128>::kLdsBytes);
  out->block_x = 64;
  out->block_y = route == kRouteWide ? (uint32_t)m16w::kWaves : (uint32_t)kWaves;
  out->grid_x = (uint32_t)B;
  out->grid_y = (uint32_t)H;
  out->grid_z = plan.nqt;
  out->staging = (uint32_t)plan.staging;
  out->aligned_rows = (d % 8 == 0) ? 1u : 0u;
  out->persistent = persistent ? 1u : 0u;
  if (route == kRouteTall) {
    out->tile_q = (uint32_t)m16x::kBr;
    out->grid_z = (uint32_t)(N / m16x::kBr);
  }
  return FLASH_HIP_OK;
}

const char* flash_kernel_type_name(int kernel_type) {
  if (kernel_type < 0 || kernel_type >= FLASH_KT_COUNT) return nullptr;
  return kNames[kernel_type];
}

const char* flash_hip_last_error(void) { return g_err; }

int flash_hip_version(void) { return 310; }

int flash_hip_reload_switches(void) {
  publish_switches();
  return FLASH_HIP_OK;
}

}  // extern "C"
