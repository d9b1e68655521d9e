// flash::forward on MI355X: tensor checks, output allocation, one call into the C ABI
// (include/flash_hip.h), stream synchronisation.  Replaces the body of the reference's
// flash::forward (reference flash/launch.cu:185-324); the template dispatch and launchKernel of the
// reference (:130-183, :216-318) live behind flash_fwd_f16 / flash_fwd_f32_ws.
#include "launch.hpp"

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <iostream>

#include "../../include/flash_hip.h"
#include "KernelParameters.hpp"

namespace flash {

namespace {

void check_inputs(const torch::Tensor& Q, const torch::Tensor& K, const torch::Tensor& V) {
  TORCH_CHECK(Q.dim() == 4, "flash::forward: Q must be [B, H, N, d], got ", Q.dim(), " dims");
  TORCH_CHECK(K.sizes() == Q.sizes() && V.sizes() == Q.sizes(), "flash::forward: K and V must have the shape of Q");
  TORCH_CHECK(Q.is_cuda() && K.is_cuda() && V.is_cuda(), "flash::forward: tensors must be on a HIP device");
  TORCH_CHECK(K.device() == Q.device() && V.device() == Q.device(), "flash::forward: tensors on different devices");
  TORCH_CHECK(Q.scalar_type() == at::kHalf || Q.scalar_type() == at::kBFloat16 || Q.scalar_type() == at::kFloat,
              "flash::forward: dtype must be float16, bfloat16 or float32");
  TORCH_CHECK(K.scalar_type() == Q.scalar_type() && V.scalar_type() == Q.scalar_type(),
              "flash::forward: Q, K, V must share one dtype");
}

torch::Tensor run(torch::Tensor Q, torch::Tensor K, torch::Tensor V, KernelType kernel_type, bool wait) {
  check_inputs(Q, K, V);
  Q = Q.contiguous();
  K = K.contiguous();
  V = V.contiguous();
  const int64_t B = Q.size(0);   // batch size
  const int64_t H = Q.size(1);   // number of heads
  const int64_t N = Q.size(2);   // sequence length
  const int64_t d = Q.size(3);   // head dimension
  const float softmax_scale = 1.0f / std::sqrt(static_cast<float>(d));

  // The launch geometry is not needed here (the library works it out itself), but asking for it keeps the
  // reference's behaviour of rejecting unsupported shapes and kernel types before anything is allocated
  // (the reference builds its KernelParameters object first, flash/launch.cu:206-207).  On the stack: this
  // function is the per-call path of short sequences, where microseconds of host time show.
  const int kt = static_cast<int>(kernel_type);
  if (kt < 0 || kt >= kNumKernelTypes) throw std::invalid_argument("flash::forward: unknown kernel type");
  flash_launch_info geometry;
  if (flash_fwd_query(B, H, N, d, kt, &geometry) != FLASH_HIP_OK) throw std::invalid_argument(flash_hip_last_error());

  // PyTorch-ROCm tensors carry the device type "cuda"; these are the guard and stream types that go
  // with it.
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  auto O = torch::empty_like(Q);  // every element is written by the kernel
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(Q.device().index());
  int rc;
  if (Q.scalar_type() == at::kHalf) {
    rc = flash_fwd_f16(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), B, H, N, d, softmax_scale, kt,
                       stream.stream());
  } else if (Q.scalar_type() == at::kBFloat16) {
    rc = flash_fwd_bf16(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), B, H, N, d, softmax_scale, kt,
                        stream.stream());
  } else {
    // fp32 tensors, the reference's own dtype (flash/launch.cu:165-176).  Short sequences are converted
    // inside the attention kernel and need no scratch (ws_bytes == 0); longer ones get per-call scratch
    // for the fp16 copies of K and V, as the reference allocates l and m per call (launch.cu:150-157).  The
    // block goes back to the caching allocator when this function returns; the allocator hands it out
    // again in stream order, so the kernels still running on this stream keep exclusive use of it.
    const int64_t ws_bytes = flash_fwd_f32_workspace_bytes(B, H, N, d, kt);
    torch::Tensor workspace;
    if (ws_bytes > 0) workspace = torch::empty({ws_bytes}, Q.options().dtype(at::kByte));
    rc = flash_fwd_f32_ws(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), B, H, N, d, softmax_scale, kt,
                          ws_bytes > 0 ? workspace.data_ptr() : nullptr, ws_bytes > 0 ? ws_bytes : 0, stream.stream());
  }
  if (rc == FLASH_HIP_ERR_INVALID_ARGUMENT || rc == FLASH_HIP_ERR_UNSUPPORTED_DIM)
    throw std::invalid_argument(flash_hip_last_error());
  if (rc != FLASH_HIP_OK) throw std::runtime_error(flash_hip_last_error());
  if (wait) {
    // (polling hipStreamQuery instead was measured and is no faster: 73.7 against 70.7 us per blocking call
    // at (4,96,512,128); the runtime's own wait already spins before it parks the thread)
    const hipError_t e = hipStreamSynchronize(stream.stream());
    if (e != hipSuccess) {
      // the reference prints and exits on a device error (flash/launch.cu:68-77)
      std::cerr << "GPUassert: " << hipGetErrorString(e) << " " << __FILE__ << " " << __LINE__ << std::endl;
      std::exit(static_cast<int>(e));
    }
  }
  return O;
}

}  // namespace

torch::Tensor forward(torch::Tensor Q, torch::Tensor K, torch::Tensor V, KernelType kernel_type) {
  return run(std::move(Q), std::move(K), std::move(V), kernel_type, /*wait=*/true);
}

torch::Tensor forward_async(torch::Tensor Q, torch::Tensor K, torch::Tensor V, KernelType kernel_type) {
  return run(std::move(Q), std::move(K), std::move(V), kernel_type, /*wait=*/false);
}

}  // namespace flash
