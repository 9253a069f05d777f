// Legacy tile variants 0-11 (single / double stage kernel and the first ring kernel): kept as the measured steps of
// the optimisation ladder and for A/B runs; nothing in the forward selects them.  Own translation
// unit so the library builds in parallel.
#include "gemm_kernels.h"

namespace vdr {

hipError_t launch_gemm_legacy(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  switch (variant) {
    case 0:
    case 1:
      return launch_cfg<2, 2, 2, 2, 0>(a, epilogue, s);  // 128x128, 4 waves, single stage
    case 2:
      return launch_cfg<2, 2, 4, 2, 0>(a, epilogue, s);  // 256x128, 4 waves (wave 128x64)
    case 3:
      return launch_cfg<4, 2, 2, 2, 0>(a, epilogue, s);  // 256x128, 8 waves (wave 64x64)
    case 4:
      return launch_cfg<2, 4, 4, 2, 1>(a, epilogue, s);  // 256x256, 8 waves (wave 128x64), 2 stages
    case 5:
      return launch_cfg<4, 2, 2, 4, 1>(a, epilogue, s);  // 256x256, 8 waves (wave 64x128), 2 stages
    case 6:
      return launch_cfg<2, 4, 2, 2, 1>(a, epilogue, s);  // 128x256, 8 waves (wave 64x64), 2 stages
    case 7:
      return launch_cfg<2, 2, 2, 2, 1>(a, epilogue, s);  // 128x128, 4 waves, 2 stages
    case 8:
      return launch_cfg<2, 2, 4, 2, 13>(a, epilogue, s);  // ring: 256x128, 4 waves (wave 128x64), 3 x 24 KB, 2 WG/CU
    case 9:
      return launch_cfg<4, 2, 2, 4, 14>(a, epilogue, s);  // ring: 256x256, 8 waves (wave 64x128), 4 x 32 KB
    case 10:
      return launch_cfg<2, 2, 2, 4, 13>(a, epilogue, s);  // ring: 128x256, 4 waves (wave 64x128), 3 x 24 KB, 2 WG/CU
    case 11:
      return launch_cfg<2, 4, 4, 2, 14>(a, epilogue, s);  // ring: 256x256, 8 waves (wave 128x64), 4 x 32 KB
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace vdr
