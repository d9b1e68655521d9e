// torch.ops.pyflash.* for PyTorch-ROCm: the reference's twelve operator names with the schema
// "(Tensor Q, Tensor K, Tensor V) -> Tensor" (reference pyflash/pyflash.cpp:10-35), each bound to
// flash::forward with the KernelType the reference binds it to.  Loaded the same way:
//   torch.ops.load_library("build/pyflash/libpyflash.so")   (reference benchmarks/bench.py:12-14)
#include <torch/extension.h>

#include <string>

#include "../flash/launch.hpp"

namespace {

template <flash::KernelType KT>
torch::Tensor attention_op(torch::Tensor Q, torch::Tensor K, torch::Tensor V) {
  return flash::forward(std::move(Q), std::move(K), std::move(V), KT);
}

template <flash::KernelType KT>
void bind(torch::Library& m, const char* op_name) {
  const std::string schema = std::string(op_name) + "(Tensor Q, Tensor K, Tensor V) -> Tensor";
  // A plain function pointer: C++ exceptions travel through the dispatcher to torch's Python
  // binding, which translates them with the GIL held.  (The reference wraps the function with
  // torch::wrap_pybind_function, whose error path touches the interpreter without the GIL when
  // the op is invoked through torch.ops.)
  m.def(schema.c_str(), &attention_op<KT>);
}

}  // namespace

TORCH_LIBRARY(pyflash, m) {
  using KT = flash::KernelType;
  bind<KT::naive1D>(m, "naive");
  bind<KT::scalar2D>(m, "scalar2d");
  bind<KT::scalar2D_row_tile>(m, "scalar2d_row_tile");
  bind<KT::warp_wmma>(m, "warp_wmma");
  bind<KT::block_wmma>(m, "block_wmma");
  bind<KT::wmma_row_block>(m, "block_wmma_row_block");
  bind<KT::block_wmma_async>(m, "block_wmma_async");
  bind<KT::mma>(m, "mma");
  bind<KT::mma_swizzle>(m, "mma_swizzle");
  bind<KT::mma_qreg>(m, "mma_qreg");
  bind<KT::mma_qreg_f32x4load>(m, "mma_qreg_vld");
  bind<KT::mma_qreg_async>(m, "mma_qreg_async");
}
