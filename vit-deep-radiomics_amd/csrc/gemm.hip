// bf16 MFMA GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)   -- entry point and the
// default (ring3, 16x16x32 MFMA) tile variants.  Kernels live in gemm_kernels.h; the older variants are instantiated
// in gemm_ring2.hip / gemm_legacy.hip.
//
// Replaces the nn.Linear calls under nn.MultiheadAttention / nn.TransformerEncoderLayer
// (reference src/models_archs.py:130-135) and attn.qkv / attn.proj / mlp.fc1 / mlp.fc2 of the
// frozen ViTs called at src/tfds_dense_descriptor.py:123, plus the patchify conv as an im2col GEMM
// (src/tfds_dense_descriptor.py:128).
#include "gemm_kernels.h"

namespace vdr {

hipError_t launch_gemm_legacy(const GemmArgs& a, int epilogue, int variant, hipStream_t s);  // gemm_legacy.hip
hipError_t launch_gemm_ring2(const GemmArgs& a, int epilogue, int variant, hipStream_t s);   // gemm_ring2.hip

int gemm_num_variants() { return 26; }

hipError_t launch_gemm(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  if (a.K <= 0 || (a.K & 63) || (a.N & 7) || a.M <= 0) return hipErrorInvalidValue;
  if (epilogue == EPI_SWIGLU && (a.N & 63)) return hipErrorInvalidValue;
  g_gemm_ablation = variant / 100;
  variant %= 100;
  if ((a.ln_stats || a.ln_cpart || a.ln_part || a.win_ws || a.a_rpg || a.out_f32) && variant < 12) return hipErrorInvalidValue;
  if (a.out_f32 && epilogue != EPI_BIAS) return hipErrorInvalidValue;  // needs epilogue_lds (ring2 / ring3)
  if (a.ln_part && (a.N & 63)) return hipErrorInvalidValue;
  if (variant < 12) return launch_gemm_legacy(a, epilogue, variant, s);
  if (variant < 22) return launch_gemm_ring2(a, epilogue, variant, s);
  switch (variant) {
    case 22:
      return launch_cfg<2, 4, 2, 2, 33>(a, epilogue, s);  // ring3 (16x16x32 MFMA): 128x256, 8 waves, 3 x 24 KB, 2 WG/CU
    case 23:
      return launch_cfg<4, 4, 2, 2, 33>(a, epilogue, s);  // ring3: 256x256, 16 waves, 3 x 32 KB
    case 24:
      return launch_cfg<2, 2, 2, 2, 33>(a, epilogue, s);  // ring3: 128x128, 4 waves, 3 x 16 KB, 3 WG/CU
    case 25:
      return launch_cfg<2, 2, 2, 2, 43>(a, epilogue, s);  // ring3k: 128x128 tile, 8 waves = 2 K-groups x (2x2), 3 x 32 KB
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace vdr
