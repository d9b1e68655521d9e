// Public C++ entry of the attention forward for MI355X.  Same declarations as the reference's
// flash/launch.hpp:7-24 (KernelType, flash::forward) and :28-46 (to_string), so callers such as
// the runner and the pyflash extension compile unchanged; the implementation (launch.cpp) forwards
// to the C ABI of libflash_hip.so (include/flash_hip.h).
#pragma once
#include <torch/torch.h>

#include <stdexcept>
#include <string>

namespace flash {

// Values and order are part of the interface: the integer value is what crosses the C ABI.
enum class KernelType {
  naive1D,
  scalar2D,
  scalar2D_row_tile,
  warp_wmma,
  block_wmma,
  wmma_row_block,
  block_wmma_async,
  mma,
  mma_swizzle,
  mma_qreg,
  mma_qreg_f32x4load,
  mma_qreg_async,
};

constexpr int kNumKernelTypes = 12;

// O = softmax(Q K^T / sqrt(d)) V for contiguous [B, H, N, d] tensors on a HIP device.
// fp16 tensors are consumed as they are; fp32 tensors (what the reference's callers pass) are
// rounded to fp16 inside the kernel and O comes back as fp32.  Blocks until the kernel has
// finished, like the reference (flash/launch.cu:320-321).  Every KernelType runs the one native
// MFMA kernel; the type only selects the argument checks the reference applies to it.
// Throws std::invalid_argument for the shapes the reference rejects (launch.cu:79-85,267-269,
// 284-286,301-303,317) and std::runtime_error for launch failures.
torch::Tensor forward(torch::Tensor Q, torch::Tensor K, torch::Tensor V, KernelType kernel_type);

// Same, without the final synchronisation: work is enqueued on the current HIP stream of Q's
// device.  Used by multi-GPU drivers that launch on several devices before waiting.
torch::Tensor forward_async(torch::Tensor Q, torch::Tensor K, torch::Tensor V, KernelType kernel_type);

}  // namespace flash

inline std::string to_string(flash::KernelType type) {
  static const char* const names[flash::kNumKernelTypes] = {
      "naive1D", "scalar2D",    "scalar2D_row_tile", "warp_wmma",          "block_wmma",    "wmma_row_block",
      "block_wmma_async", "mma", "mma_swizzle",      "mma_qreg",           "mma_qreg_f32x4load", "mma_qreg_async"};
  const int i = static_cast<int>(type);
  if (i < 0 || i >= flash::kNumKernelTypes) throw std::invalid_argument("wrong kernel type");
  return names[i];
}
