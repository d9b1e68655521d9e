// Launch geometry of the attention forward, with the interface of the reference's
// flash/KernelParameters.hpp:63-71 (KernelParametersBase: tileSize / sramSize / blockDim / gridDim)
// and :383-422 (KernelParametersFactory::create).  The reference has one parameter class per rung
// of its CUDA kernel ladder; here every KernelType is served by the single MFMA kernel, so there is
// one concrete class, MFMAQregParameters, whose numbers come from the library itself
// (flash_fwd_query in include/flash_hip.h) and therefore cannot drift from what is launched.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "../../include/flash_hip.h"
#include "launch.hpp"

namespace flash {

class KernelParametersBase {
 public:
  KernelParametersBase() = default;
  virtual dim3 tileSize() = 0;  // (Bc, Br, padded head dim)
  virtual uint32_t sramSize() = 0;
  virtual dim3 blockDim() = 0;
  virtual dim3 gridDim() = 0;
  virtual ~KernelParametersBase() = default;
};

// Geometry of the wave64 MFMA kernel: 256 query rows per workgroup (Br), worked by 4 waves (one per
// SIMD) x 64 rows or, for whole-tile d = 128 shapes, by 8 waves x 32 rows (512 rows, 4 waves x 128, for
// whole 512-row tiles of d <= 64); 64-row K/V
// tiles (Bc), head dim padded to 64 or 128, rings of three (persistent kernel: four) K and as many V
// tiles in LDS.  The numbers come from the library (flash_fwd_query), not from this header.
// Replaces MMAQregParameters / MMAQregAsyncParameters (reference flash/KernelParameters.hpp:332-381).
class MFMAQregParameters : public KernelParametersBase {
  flash_launch_info info_{};

 public:
  MFMAQregParameters(int batchSize, int numHeads, int seqLen, int headDim, KernelType type) {
    const int rc = flash_fwd_query(batchSize, numHeads, seqLen, headDim, static_cast<int>(type), &info_);
    if (rc != FLASH_HIP_OK) throw std::invalid_argument(flash_hip_last_error());
  }
  static constexpr uint32_t rowsPerTileQ() { return 256; }
  static constexpr uint32_t rowsPerTileK() { return 64; }
  static constexpr uint32_t wavesPerBlock() { return 4; }
  dim3 tileSize() override { return dim3(info_.tile_kv, info_.tile_q, info_.tile_d); }
  uint32_t sramSize() override { return info_.lds_bytes; }
  dim3 blockDim() override { return dim3(info_.block_x, info_.block_y); }
  dim3 gridDim() override { return dim3(info_.grid_x, info_.grid_y, info_.grid_z); }
  // 0 element loads, 1 128-bit register staging, 2 direct global->LDS
  uint32_t staging() const { return info_.staging; }
};

class KernelParametersFactory {
 public:
  static std::unique_ptr<KernelParametersBase> create(int batchSize, int numHeads, int seqLen, int headDim,
                                                      KernelType kernelType) {
    const int kt = static_cast<int>(kernelType);
    if (kt < 0 || kt >= kNumKernelTypes) {
      std::ostringstream err;
      err << __FILE__ << "(" << __LINE__ << ") error: unknown kernel type";
      throw std::invalid_argument(err.str());
    }
    return std::make_unique<MFMAQregParameters>(batchSize, numHeads, seqLen, headDim, kernelType);
  }
};

}  // namespace flash
