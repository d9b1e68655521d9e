/*
 * flash_hip.h — C ABI of the MI355X (gfx950) attention-forward library, libflash_hip.so.
 *
 * This is the drop-in boundary for the one hot path of ishovkun/flash-attention: everything the
 * reference does between `flash::forward` reading the tensor shapes and `cudaDeviceSynchronize`
 * (reference flash/launch.cu:185-324).  Plain pointers and sizes only; no torch types.  The
 * reference's C++/torch surface (flash::forward, KernelParameters, pyflash) is re-created on top of
 * these entry points under flash-attention_amd/flash and flash-attention_amd/pyflash.
 *
 * Conventions
 *   - All tensors are contiguous [B, H, N, d] (reference flash/kernel_mma_qreg.cuh:30-34:
 *     qkv_offset = (b*H + h)*N*d); K and V have the shape of Q; O has the shape of Q.
 *   - Tensor pointers need only the alignment of their element type.  The fast kernels (any of the three
 *     tensor types with d % 8 == 0) move 16 bytes per lane and are used when q, k, v and o all start on a
 *     16-byte boundary (any whole allocation does); otherwise fp16 and fp32 tensors take the slower
 *     element-staging kernel, bfloat16 tensors are rejected (INVALID_ARGUMENT).
 *   - Device = the calling thread's current HIP device.  `hip_stream` is a hipStream_t (NULL = the
 *     default stream).  Calls enqueue work and return; they never synchronise, allocate or free.
 *   - Return value: FLASH_HIP_OK (0) or a negative flash_hip_status.  The text of the last error of
 *     the calling thread is available from flash_hip_last_error().
 *   - kernel_type is the integer value of the reference's `enum class KernelType`
 *     (reference flash/launch.hpp:7-20), 0..11.
 */
#ifndef FLASH_HIP_H_
#define FLASH_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum flash_hip_status {
  FLASH_HIP_OK = 0,
  FLASH_HIP_ERR_INVALID_ARGUMENT = -1, /* maps to std::invalid_argument (launch.cu:79-85,267-269,284-286,301-303,317) */
  FLASH_HIP_ERR_UNSUPPORTED_DIM = -2,  /* head dim has no compiled tile; std::invalid_argument in the reference (launch.cu:270-280) */
  FLASH_HIP_ERR_RUNTIME = -3,          /* HIP runtime error; reference prints and exits (launch.cu:63-77) */
  FLASH_HIP_ERR_LDS = -4               /* dynamic LDS request rejected; std::runtime_error in the reference (launch.cu:141-147) */
} flash_hip_status;

/* Integer values of the reference's KernelType (flash/launch.hpp:7-20). */
typedef enum flash_hip_kernel_type {
  FLASH_KT_NAIVE1D = 0,
  FLASH_KT_SCALAR2D = 1,
  FLASH_KT_SCALAR2D_ROW_TILE = 2,
  FLASH_KT_WARP_WMMA = 3,
  FLASH_KT_BLOCK_WMMA = 4,
  FLASH_KT_WMMA_ROW_BLOCK = 5,
  FLASH_KT_BLOCK_WMMA_ASYNC = 6,
  FLASH_KT_MMA = 7,
  FLASH_KT_MMA_SWIZZLE = 8,
  FLASH_KT_MMA_QREG = 9,
  FLASH_KT_MMA_QREG_F32X4LOAD = 10,
  FLASH_KT_MMA_QREG_ASYNC = 11,
  FLASH_KT_COUNT = 12
} flash_hip_kernel_type;

/*
 * Launch geometry of one call, the data the reference exposes through KernelParametersBase
 * (flash/KernelParameters.hpp:63-71: tileSize / sramSize / blockDim / gridDim).
 */
typedef struct flash_launch_info {
  uint32_t tile_kv;      /* Bc: K/V rows per LDS tile                (tileSize().x) */
  uint32_t tile_q;       /* Br: Q rows per workgroup: 256, or 512 for head dims <= 64 with N % 512 == 0 (tileSize().y) */
  uint32_t tile_d;       /* head dim padded to the compiled tile     (tileSize().z) */
  uint32_t lds_bytes;    /* dynamic LDS per workgroup                (sramSize()); persistent kernels hold
                            4 bytes of static LDS besides (their overflow flag, see `persistent`) */
  uint32_t block_x;      /* lanes per wave = 64                      (blockDim().x) */
  uint32_t block_y;      /* waves per workgroup: 4, or 8 for whole-tile shapes with d > 64 (blockDim().y) */
  uint32_t grid_x;       /* B                                        (gridDim().x)  */
  uint32_t grid_y;       /* H                                        (gridDim().y)  */
  uint32_t grid_z;       /* ceil(N / Br): grid = the work items; when N % Br == 0 a persistent kernel
                            walks them with one workgroup per CU      (gridDim().z)  */
  uint32_t staging;      /* how K/V tiles reach LDS: 0 = element loads through registers (any head dim, any
                            alignment), 1 = fp32 tensors through registers, rounded to fp16 on the way by converting loads
                            (fp32 tensors, d % 8 == 0), 2 = direct global->LDS loads (buffer_load ... lds;
                            fp16 / bfloat16 tensors, and fp32 tensors behind the conversion or padding pass) */
  uint32_t aligned_rows; /* 1 when d % 8 == 0 (128-bit row chunks usable), else 0 */
  uint32_t persistent;   /* 1: the work items (grid_x * grid_y * grid_z of them) are walked by a persistent
                            kernel launched as a 1-D grid of min(items, compute units rounded down to a
                            multiple of 8) workgroups; 0: one workgroup per item, 1-D grid of all items.
                            A persistent kernel takes the softmax's running maximum in a few steps of an item
                            only; an item in which a probability left the fp16 range on the way (a score more
                            than 16 exp2 units above everything those steps saw) is computed a second time by
                            the same workgroup, with the maximum in every step, and so is the rest of that
                            workgroup's list (once).  The result is the same either way; the time of a call
                            therefore depends on its data (8 % at (8,96,4096,128), up to 40 % at N = 512, on
                            inputs built so that every workgroup meets such an item). */
} flash_launch_info;

/* Which entry point a geometry query describes. */
typedef enum flash_hip_entry {
  FLASH_ENTRY_F16 = 0,    /* flash_fwd_f16 */
  FLASH_ENTRY_BF16 = 1,   /* flash_fwd_bf16 */
  FLASH_ENTRY_F32 = 2,    /* flash_fwd_f32, and flash_fwd_f32_ws without a usable workspace: conversion inside the
                             attention kernel, four waves x 64 query rows */
  FLASH_ENTRY_F32_WS = 3  /* flash_fwd_f32_ws with a workspace, flash_fwd_f32_static: as FLASH_ENTRY_F32 for
                             sequences up to 2048 (768 for head dims up to 64), beyond that the conversion pass
                             and the fp16 kernels with fp32 output; for d % 8 != 0 (from about 2 GFLOP on) the
                             launch on the padded copies: tile_d from the padded head dim, staging 2 */
} flash_hip_entry;

/*
 * fp16 attention forward: O = softmax(Q K^T * softmax_scale) V per (b, h), non-causal, no mask.
 * Replaces the body of the reference's flash::forward after tensor unpacking
 * (flash/launch.cu:197-204 KernelArgs, :216-318 dispatch, :130-183 launchKernel) and the kernels
 * kernel_mma_qreg / _f32x4load / _async (flash/kernel_mma_qreg*.cuh).
 * q, k, v, o: device pointers to IEEE binary16 data.  fp16 operands, fp32 accumulation.
 * softmax_scale must be positive and finite (-> INVALID_ARGUMENT otherwise; the reference always uses
 * 1 / sqrt(d), flash/launch.cu:192).
 * Errors mirror the reference: d > 128 -> UNSUPPORTED_DIM; kernel_type 9 needs d % 2 == 0, 10 and
 * 11 need d % 4 == 0 -> INVALID_ARGUMENT; unknown kernel_type -> INVALID_ARGUMENT.
 */
int flash_fwd_f16(const void* q, const void* k, const void* v, void* o,
                  int64_t B, int64_t H, int64_t N, int64_t d,
                  float softmax_scale, int kernel_type, void* hip_stream);

/*
 * bfloat16 attention forward: the same kernels as flash_fwd_f16 with bfloat16 operands on the matrix
 * cores (v_mfma_f32_16x16x32_bf16), fp32 accumulation, bfloat16 output.  No counterpart in the
 * reference (its tensors are fp32, flash/launch.cu:165-176); offered because bfloat16 is the usual
 * storage type of the models this shape comes from.  d must be a multiple of 8 (-> UNSUPPORTED_DIM
 * otherwise); other errors as flash_fwd_f16.  P is rounded to 8 mantissa bits before the P V product,
 * so the error against an fp32 evaluation on the same inputs is several times that of fp16
 * (measured max-abs 1e-3 ... 8.4e-3 on standard-normal data; tested against 1.6e-2).
 */
int flash_fwd_bf16(const void* q, const void* k, const void* v, void* o,
                   int64_t B, int64_t H, int64_t N, int64_t d,
                   float softmax_scale, int kernel_type, void* hip_stream);

/*
 * fp32 attention forward for the reference's fp32 callers (main.cpp:30-32, benchmarks/bench.py:34-36
 * hand fp32 tensors to flash::forward, which reads them with data_ptr<float>(), launch.cu:165-176).
 * Q, K, V are rounded to fp16 (round-to-nearest-even) while they are staged into LDS/registers, the
 * arithmetic is the same fp16-MFMA / fp32-accumulate pipeline, O is written as fp32.  No workspace, one
 * kernel.  With d % 8 == 0 and 16-byte-aligned tensors the 16x16x32 kernels of flash_fwd_f16 do the
 * staging (whole query tiles on the persistent kernel) and K and V are converted by the loads themselves
 * (typed buffer loads with a 16-bit destination: same rounding, but a finite value beyond the fp16 range
 * arrives as +-65504 where a conversion instruction, the conversion pass of flash_fwd_f32_ws and torch's
 * .half() give +-inf; such inputs have no meaningful result on any route); otherwise element by element.
 */
int flash_fwd_f32(const void* q, const void* k, const void* v, void* o,
                  int64_t B, int64_t H, int64_t N, int64_t d,
                  float softmax_scale, int kernel_type, void* hip_stream);

/*
 * fp32 attention forward with a caller-provided workspace: the fast path for the same fp32 callers at
 * every sequence length (reference main.cpp:30-32, benchmarks/bench.py:34-36, flash/launch.cu:165-176; the
 * reference itself allocates per-call scratch for its non-register kernels, launch.cu:150-157).
 * Converting inside the attention kernel costs every workgroup that reads a K/V tile its conversion, so
 * beyond a sequence length where a tile has many readers (N > 2048; N > 768 for head dims up to 64, except
 * N % 512 == 0, which the 128-rows-per-wave kernel converts itself at every length) and when d % 8 == 0, one streaming pass rounds K and V to fp16 (round-to-nearest-even, the rounding
 * flash_fwd_f32 applies) into `workspace` once, and the fp16 kernels of flash_fwd_f16 run on those copies,
 * reading Q from the caller's fp32 tensor (rounded the same way on its way into the registers; a layout
 * that cannot, reachable through the routing switches only, takes Q from the pass as well) and writing O as
 * fp32 straight from the
 * accumulators.  Same results as flash_fwd_f32 up to the order of fp32 additions.  Shorter sequences are
 * handled exactly as by flash_fwd_f32 and touch no workspace.
 * Head dims that are not a multiple of 8 (the reference's runner tests 50 and 69, main.cpp:169-183) have rows
 * without 16-byte chunks, which flash_fwd_f32 stages element by element at about a sixth of the fast kernels'
 * rate.  With a workspace, and from about 2 GFLOP of attention on (4*B*H*N*N*d; below that the one-kernel
 * route stays), one pass copies Q, K and V into rows of the next multiple of 8 elements, zero-filled (K and V
 * rounded to fp16 as above; Q as well, or kept as fp32 for the one wave layout that reads it so), the kernels of
 * flash_fwd_f16 run on the copies (routed as for fp16 tensors) with the caller's softmax_scale, and one more
 * pass copies the first d columns of their fp32 output into `o`.  The caller's tensors are read and written
 * element by element on this route and need no 16-byte alignment.  Measured at (4,96,4096,69): 630 against
 * 187 TFLOP/s.
 * `workspace`: device memory, 16-byte aligned, at least flash_fwd_f32_workspace_bytes(...) bytes;
 * its contents are scratch and must not be touched until the work enqueued by this call has finished.
 * A NULL workspace, or a shape for which flash_fwd_f32_workspace_bytes returns 0, makes the call
 * behave exactly like flash_fwd_f32.  A non-NULL workspace that is too small -> INVALID_ARGUMENT.
 */
int flash_fwd_f32_ws(const void* q, const void* k, const void* v, void* o,
                     int64_t B, int64_t H, int64_t N, int64_t d,
                     float softmax_scale, int kernel_type,
                     void* workspace, int64_t workspace_bytes, void* hip_stream);

/*
 * flash_fwd_f32_ws without the workspace argument, for callers that cannot allocate device memory
 * or hand any over: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  No counterpart in the reference.  The scratch for the fp16 copies of K and V (and of Q where the pass converts it) is a
 * device array built into the library, FLASH_HIP_STATIC_SCRATCH_BYTES (48 MiB) on every GPU the
 * library runs on, held until the process ends.  The call allocates and frees nothing (no hipMalloc,
 * no allocating library routine); it enqueues the same kernels as flash_fwd_f32_ws on `hip_stream`
 * and gives bit-identical results.
 * Limit: shapes whose workspace (flash_fwd_f32_workspace_bytes) exceeds the built-in buffer are rejected
 * with INVALID_ARGUMENT and a text naming both sizes; use flash_fwd_f32_ws for those.  The workspace holds
 * the fp16 copies of K and V (B*H*N*d <= 12 Mi elements fit).  Calls that need no workspace have no limit and
 * behave like flash_fwd_f32: sequences short enough to be converted inside the attention kernel
 * (flash_fwd_f32_workspace_bytes returns 0: N <= 2048, N <= 768 or N % 512 == 0 for head dims up to 64),
 * tensors off a 16-byte boundary.  Head dims that are not a multiple of 8 use the buffer for their padded
 * copies while those fit and take the element-staging kernel (no scratch) beyond, so they are never refused.
 * All calls on one GPU share that GPU's buffer: calls that may overlap in time on one GPU (different
 * streams, different threads) race on it and must be serialised by the caller.  Nothing in the
 * buffer needs clearing between calls: every byte a call reads was written by the same call.
 * The buffer's address is looked up once per device, on the first call made with that device current:
 * make one plain call per device (and per kernel shape class, which sets the kernels' LDS attribute)
 * before recording calls into a HIP graph.
 */
#define FLASH_HIP_STATIC_SCRATCH_BYTES (48ll << 20)
int flash_fwd_f32_static(const void* q, const void* k, const void* v, void* o,
                         int64_t B, int64_t H, int64_t N, int64_t d,
                         float softmax_scale, int kernel_type, void* hip_stream);

/* Workspace flash_fwd_f32_ws uses for this shape: the fp16 copies its conversion pass writes (two of
 * B*H*N*d elements; three where a routing switch puts a layout that cannot read fp32 Q behind the pass), 0
 * when the call converts inside the attention kernel (short sequences, whole 512-row tiles of head dims up
 * to 64, small shapes with d % 8 != 0), -1 (and an error text) when the shape or kernel type is rejected.
 * For d % 8 != 0 from about 2 GFLOP on: six fp16 tensors of the shape padded to d_pad = the next multiple
 * of 8 (K, V as fp16, room for Q as fp32, the padded fp32 output), each rounded up to 256 bytes.
 * Does not touch the GPU. */
int64_t flash_fwd_f32_workspace_bytes(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type);

/*
 * Geometry query; does not touch the GPU.  Replaces KernelParametersFactory::create
 * (flash/KernelParameters.hpp:383-422) for the native kernel.  Same error codes as flash_fwd_f16.
 * Describes the launch flash_fwd_f16 makes for 16-byte-aligned tensors (the reference's factory takes
 * no dtype either); flash_fwd_query_entry does the same for a named entry point.
 */
int flash_fwd_query(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type,
                    flash_launch_info* out);

/* As flash_fwd_query, for the entry point `entry` (a flash_hip_entry), assuming 16-byte-aligned tensors. */
int flash_fwd_query_entry(int64_t B, int64_t H, int64_t N, int64_t d, int kernel_type, int entry,
                          flash_launch_info* out);

/* Name of a kernel type as the reference's to_string prints it (flash/launch.hpp:28-46); NULL if
 * out of range. */
const char* flash_kernel_type_name(int kernel_type);

/* Last error text of the calling thread ("" if none).  The pointer stays valid until the next
 * failing call on the same thread. */
const char* flash_hip_last_error(void);

/* Library version, major*10000 + minor*100 + patch.  310 (padded route for d % 8 != 0 behind
 * flash_fwd_f32_ws; no interface change); flash_launch_info unchanged since 200 (which
 * appended `persistent`: callers built against 100 must be recompiled, the query functions write the
 * whole structure). */
int flash_hip_version(void);

/*
 * Diagnostics; no counterpart in the reference.  The library routes a call by a few switches that it
 * reads from the environment ONCE, on the first call of the process (FLASH_HIP_PERSISTENT,
 * FLASH_HIP_PERSISTENT_MAX_N, FLASH_HIP_WIDE_MAX_N, FLASH_HIP_TALL, FLASH_HIP_VEC_MAX_N: wave layout and
 * kernel variant, for A/B measurements; every setting computes the same function).  This call reads them
 * again, for harnesses that compare two settings inside one process.  Safe against calls in flight on
 * other threads (each routes by the old or the new set as a whole).  Always returns FLASH_HIP_OK.
 */
int flash_hip_reload_switches(void);

#ifdef __cplusplus
}
#endif
#endif /* FLASH_HIP_H_ */
