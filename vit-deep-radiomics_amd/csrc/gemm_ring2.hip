// ring2 tile variants 12-21: the mid-step-barrier ring pipeline on the 32x32x16 MFMA.  Variant 16 /
// 19 / 20 / 21 were the defaults before ring3; kept for A/B runs and as the reference point of the MFMA-shape result.
#include "gemm_kernels.h"

namespace vdr {

hipError_t launch_gemm_ring2(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  switch (variant) {
    case 12:
      return launch_cfg<4, 2, 2, 4, 24>(a, epilogue, s);  // ring2: 256x256, 8 waves (wave 64x128), 4 x 32 KB
    case 13:
      return launch_cfg<2, 4, 4, 2, 24>(a, epilogue, s);  // ring2: 256x256, 8 waves (wave 128x64), 4 x 32 KB
    case 14:
      return launch_cfg<2, 2, 4, 2, 23>(a, epilogue, s);  // ring2: 256x128, 4 waves (wave 128x64), 3 x 24 KB, 2 WG/CU
    case 15:
      return launch_cfg<2, 2, 2, 4, 23>(a, epilogue, s);  // ring2: 128x256, 4 waves (wave 64x128), 3 x 24 KB, 2 WG/CU
    case 16:
      return launch_cfg<2, 2, 2, 2, 23>(a, epilogue, s);  // ring2: 128x128, 4 waves (wave 64x64), 3 x 16 KB, 3 WG/CU
    case 17:
      return launch_cfg<2, 2, 2, 2, 24>(a, epilogue, s);  // ring2: 128x128, 4 waves (wave 64x64), 4 x 16 KB, 2 WG/CU
    case 18:
      return launch_cfg<4, 4, 2, 2, 24>(a, epilogue, s);  // ring2: 256x256, 16 waves (wave 64x64), 4 x 32 KB
    case 19:
      return launch_cfg<2, 4, 2, 2, 23>(a, epilogue, s);  // ring2: 128x256, 8 waves (wave 64x64), 3 x 24 KB, 2 WG/CU
    case 20:
      return launch_cfg<4, 2, 2, 2, 23>(a, epilogue, s);  // ring2: 256x128, 8 waves (wave 64x64), 3 x 24 KB, 2 WG/CU
    case 21:
      return launch_cfg<4, 4, 2, 2, 23>(a, epilogue, s);  // ring2: 256x256, 16 waves (wave 64x64), 3 x 32 KB
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace vdr
