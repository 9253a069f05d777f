// bf16 MFMA GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)   -- entry point, the ring3 tile
// variants and the weight packer.  Kernels live in gemm_kernels.h; the ring4 variants are instantiated in gemm_ring4.hip.
//
// Replaces the nn.Linear calls under nn.MultiheadAttention / nn.TransformerEncoderLayer
// (reference src/models_archs.py:130-135) and attn.qkv / attn.proj / mlp.fc1 / mlp.fc2 of the
// frozen ViTs called at src/tfds_dense_descriptor.py:123, plus the patchify conv as a GEMM
// (src/tfds_dense_descriptor.py:128).
#include "gemm_kernels.h"

namespace vdr {

hipError_t launch_gemm_ring4(const GemmArgs& a, int epilogue, int variant, hipStream_t s);  // gemm_ring4.hip
hipError_t launch_gemm_8p(const GemmArgs& a, int epilogue, hipStream_t s);                   // gemm_8p.hip

// [N][K] (row stride ld) -> pair-interleaved [N/2][K/32][2][32]; one 16-byte chunk per thread
__global__ __launch_bounds__(256) void w_interleave_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K,
                                                           int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // destination chunk
  const int64_t total = (int64_t)N * (K >> 3);
  if (i >= total) return;
  const int c = (int)(i & 3), r1 = (int)((i >> 2) & 1);
  const int64_t t = i >> 3;
  const int kb = (int)(t % (K >> 5));
  const int64_t pair = t / (K >> 5);
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (2 * pair + r1) * ld + kb * 32 + c * 8);
  *reinterpret_cast<bf16x8*>(dst + i * 8) = v;
}

hipError_t launch_w_interleave(const void* src, void* dst, int N, int K, int64_t ld, hipStream_t s) {
  if (N <= 0 || (N & 1) || K <= 0 || (K & 31)) return hipErrorInvalidValue;
  const int64_t total = (int64_t)N * (K >> 3);
  hipLaunchKernelGGL(w_interleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, N, K, ld);
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  if (a.K <= 0 || (a.K & 63) || (a.N & 7) || a.M <= 0) return hipErrorInvalidValue;
  if (epilogue == EPI_SWIGLU && (a.N & 63)) return hipErrorInvalidValue;
#ifdef VDR_TUNING
  g_gemm_gn = variant >= 1000 ? variant / 1000 - 1 : -1;  // tools/: (gn + 1) * 1000 + v forces the column-group width gn
  variant %= 1000;
  g_gemm_ablation = variant / 100;  // tools/: 1xx no epilogue, 4xx no global loads after the ring fill, 8xx no stores
  variant %= 100;
#endif
  if (a.out_f32 && epilogue != EPI_BIAS && epilogue != EPI_PATCH) return hipErrorInvalidValue;
  if (a.ln_part && (a.N & 63)) return hipErrorInvalidValue;
  switch (variant) {
    case 22:
      return launch_cfg<2, 4, 33>(a, epilogue, s);  // ring3: 128x256, 8 waves, 3 x 24 KB, 2 WG/CU
    case 23:
      return launch_cfg<4, 4, 33>(a, epilogue, s);  // ring3: 256x256, 16 waves, 3 x 32 KB
    case 24:
      return launch_cfg<2, 2, 33>(a, epilogue, s);  // ring3: 128x128, 4 waves, 3 x 16 KB, 3 WG/CU
    case 25:
      return launch_cfg<2, 2, 43>(a, epilogue, s);  // ring3k: 128x128 tile, 8 waves = 2 K-groups x (2x2), 3 x 32 KB
    case 26:
    case 27:
    case 28:
    case 29:
      return launch_gemm_ring4(a, epilogue, variant, s);
    case 31:
      return launch_gemm_8p(a, epilogue, s);  // 8-phase: 256x256 tile, 8 waves, one persistent workgroup per CU, plain W layout
    default:
      return hipErrorInvalidValue;  // (variants 0-21, the earlier rungs of the ladder in DESIGN.md, are no longer built; 30, the persistent stream kernel of round 3, lives in tools/micro/)
  }
}

}  // namespace vdr
