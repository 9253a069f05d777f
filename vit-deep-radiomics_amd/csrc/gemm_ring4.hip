// ring4 tile variants 26-28 (gemm_kernels.h: both operands staged in whole 128-B lines).  Own translation unit so the
// library builds in parallel.
#include "gemm_kernels.h"

namespace vdr {

hipError_t launch_gemm_ring4(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  switch (variant) {
    case 26:
      return launch_cfg<2, 4, 50>(a, epilogue, s);  // ring4: 128x256, 8 waves, 2 x 16 KB (A) + 3 x 16 KB (W), 2 WG/CU
    case 27:
      return launch_cfg<4, 4, 50>(a, epilogue, s);  // ring4: 256x256, 16 waves, 2 x 32 KB + 3 x 16 KB
    case 28:
      return launch_cfg<2, 2, 50>(a, epilogue, s);  // ring4: 128x128, 4 waves, 2 x 16 KB + 3 x 8 KB, 2 WG/CU
    case 29:
      return launch_cfg<1, 2, 50>(a, epilogue, s);  // ring4: 64x128, 2 waves, 2 x 8 KB + 3 x 8 KB: launches of a few hundred rows x 768 columns
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace vdr
