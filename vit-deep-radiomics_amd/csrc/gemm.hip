// bf16 MFMA GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)
//
// Replaces the nn.Linear calls under nn.MultiheadAttention / nn.TransformerEncoderLayer
// (reference src/models_archs.py:130-135) and attn.qkv / attn.proj / mlp.fc1 / mlp.fc2 of the
// frozen ViTs called at src/tfds_dense_descriptor.py:123, plus the patchify conv as an im2col GEMM
// (src/tfds_dense_descriptor.py:128).
//
// Structure (variant 0): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA
// 32x32x16 tiles.  A and W tiles go global -> LDS with 16-byte global_load_lds; the LDS image is
// [row][64 k] bf16 = 128-B rows with the 16-B chunk index XOR-swizzled by (row>>1)&7 (applied on
// the per-lane SOURCE address and on the ds_read_b128 address), which makes the fragment reads
// bank-conflict free.  The MFMA is issued transposed (W fragment as the A operand, activation
// fragment as the B operand) so that a lane owns one output ROW and 4 consecutive output columns
// per register group: the epilogue then reads bias/residual and writes bf16 8 bytes at a time.
#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

struct GemmK {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  const bf16_t* resid;
  const float* gamma;
  const float* pos;
  bf16_t* C;
  int64_t M;
  int N, K;
  int64_t lda, ldw, ldc, ldr;
  int rpg;
  int64_t gstride;
  int off;
  int tiles_n;
  int nwg;
};

template <int EPI>
VDR_DEV void epilogue_store(const GemmK& p, const f32x16& acc, const f32x16& acc2, int64_t m, int n_base,
                            int h) {
  // acc: 32 (n) x 32 (m) tile, this lane owns row m, columns n_base + 8g + 4h + e.
  if (m >= p.M) return;
  int64_t orow = m;
  int prow = 0;
  if (EPI == EPI_PATCH) {
    const int64_t g = m / p.rpg;
    const int i = (int)(m - g * p.rpg);
    orow = g * p.gstride + p.off + i;
    prow = p.off + i;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n_base + 8 * g + 4 * h;
    if (n >= p.N) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e];
    if (p.bias) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += b[e];
    }
    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    }
    if (EPI == EPI_SWIGLU) {
      // acc2 is the gate partner tile (x2); bias for it sits 32 packed rows further
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = acc2[4 * g + e];
      if (p.bias) {
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.bias + n + 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] += b2[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = silu(v[e]) * u[e];
    }
    if (EPI == EPI_BIAS_RESID) {
      if (p.gamma) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gm[e];
      }
      const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.resid + orow * p.ldr + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    }
    if (EPI == EPI_PATCH) {
      if (p.pos) {
        const f32x4 ps = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += ps[e];
      }
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
    int oc = n;
    if (EPI == EPI_SWIGLU) {
      // packed column n = 64*blk + t (t < 32)  ->  output feature 32*blk + t
      oc = (n >> 6) * 32 + (n & 31);
    }
    *reinterpret_cast<bf16x4*>(p.C + orow * p.ldc + oc) = o;
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(GemmK p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int NA = BM / 8 / NW;  // global_load_lds instructions per wave for the A tile
  constexpr int NB = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + BM * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int wg = xcd_remap(blockIdx.x, p.nwg);
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  // ---- staging addresses: instruction q of this wave covers tile rows (wave*NA + q)*8 .. +7
  const int srow = lane >> 3;  // row within the 8-row piece
  const int spc = lane & 7;    // physical 16-B chunk within the 128-B row
  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int r = (wave * NA + q) * 8 + srow;
    const int c = spc ^ ((r >> 1) & 7);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    a_src[q] = p.A + gr * p.lda + c * 8;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 8 + srow;
    const int c = spc ^ ((r >> 1) & 7);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    b_src[q] = p.W + (int64_t)gr * p.ldw + c * 8;
  }

  // ---- fragment read addresses
  const int swz = (lane >> 1) & 7;  // == (row >> 1) & 7 because tile row bases are multiples of 32
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (wm * TM * 32 + i * 32 + l31) * 128;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = (wn * TN * 32 + j * 32 + l31) * 128;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  const int nk = p.K >> 6;
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      glds16(a_src[q], sA + (wave * NA + q) * 1024);
      a_src[q] += 64;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      glds16(b_src[q], sB + (wave * NB + q) * 1024);
      b_src[q] += 64;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = ((2 * ks + h) ^ swz) * 16;
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sA + a_off[i] + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sB + b_off[j] + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[j][i], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns output row m (per i) and 4-column groups (per j, g)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t m = m0 + wm * TM * 32 + i * 32 + l31;
    if (EPI == EPI_SWIGLU) {
#pragma unroll
      for (int j = 0; j + 1 < TN; j += 2)
        epilogue_store<EPI>(p, acc[j][i], acc[j + 1][i], m, n0 + wn * TN * 32 + j * 32, h);
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        epilogue_store<EPI>(p, acc[j][i], acc[j][i], m, n0 + wn * TN * 32 + j * 32, h);
    }
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static hipError_t launch_cfg(const GemmArgs& a, int epi, hipStream_t s) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  GemmK k;
  k.A = (const bf16_t*)a.A;
  k.W = (const bf16_t*)a.W;
  k.bias = a.bias;
  k.resid = (const bf16_t*)a.resid;
  k.gamma = a.gamma;
  k.pos = a.pos;
  k.C = (bf16_t*)a.C;
  k.M = a.M;
  k.N = a.N;
  k.K = a.K;
  k.lda = a.lda;
  k.ldw = a.ldw;
  k.ldc = a.ldc;
  k.ldr = a.ldr;
  k.rpg = a.omap.rpg;
  k.gstride = a.omap.gstride;
  k.off = a.omap.off;
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  k.tiles_n = (a.N + BN - 1) / BN;
  const int64_t nwg = tiles_m * k.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffff) return hipErrorInvalidValue;
  k.nwg = (int)nwg;
  const dim3 grid((unsigned)nwg), block(WAVES_M * WAVES_N * 64);
  const size_t lds = (size_t)(BM + BN) * 128;
#define VDR_LAUNCH(E)                                                                             \
  case E: {                                                                                       \
    auto fn = gemm_kernel<WAVES_M, WAVES_N, TM, TN, E>;                                           \
    if (lds > 65536) {                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds);                                               \
      if (e != hipSuccess) return e;                                                              \
    }                                                                                             \
    hipLaunchKernelGGL(fn, grid, block, lds, s, k);                                               \
    break;                                                                                        \
  }
  switch (epi) {
    VDR_LAUNCH(EPI_BIAS)
    VDR_LAUNCH(EPI_BIAS_GELU)
    VDR_LAUNCH(EPI_BIAS_RESID)
    VDR_LAUNCH(EPI_SWIGLU)
    VDR_LAUNCH(EPI_PATCH)
    default:
      return hipErrorInvalidValue;
  }
#undef VDR_LAUNCH
  return hipGetLastError();
}

int gemm_num_variants() { return 3; }

hipError_t launch_gemm(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  if (a.K <= 0 || (a.K & 63) || (a.N & 7) || a.M <= 0) return hipErrorInvalidValue;
  if (epilogue == EPI_SWIGLU && (a.N & 63)) return hipErrorInvalidValue;
  switch (variant) {
    case 0:
    case 1:
      return launch_cfg<2, 2, 2, 2>(a, epilogue, s);  // 128x128, 4 waves
    case 2:
      return launch_cfg<2, 2, 4, 2>(a, epilogue, s);  // 256x128, 4 waves (wave 128x64)
    case 3:
      return launch_cfg<4, 2, 2, 2>(a, epilogue, s);  // 256x128, 8 waves (wave 64x64)
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace vdr
