// MX-fp8 GEMM for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T),  A and W as e4m3 payload + e8m0 block scales
// (mx.hip layout), fp32 accumulation on  v_mfma_scale_f32_32x32x64_f8f6f4  (2x the bf16 MFMA rate, half the
// operand bytes).  BASELINE config 5 (DINOv2 ViT-g/14, fp8 weights): qkv / w12 / w3 (fc1 / fc2) run here,
// with the activations re-quantised per 32-element block by their producers (LayerNorm, fc1 epilogue).
//
// Same skeleton as the bf16 ring kernels (gemm_kernels.h): a ring of NST LDS slots filled by 16-byte global_load_lds,
// counted vmcnt, raw s_barrier, LDS-staged fused epilogue.  A 64-byte LDS row is now 64 K elements = ONE
// scaled MFMA per 32x32 tile pair: measured operand layout (tools/micro/mxfp8_test.hip) is
//   lane (r, h): registers 0-3 = K bytes [16h, 16h+16) of scale block 0, registers 4-7 = the same of block 1,
//   block b's scale byte is supplied by lane r + 32 b, opsel picks the byte of the scale VGPR,
// i.e. exactly the two ds_read_b128 the bf16 kernel issues for its two k-steps, concatenated.  The scales of a
// unit (2 blocks x 64 rows of A and of W per wave) arrive by ONE extra 4-byte global_load_lds per wave into a
// wave-private 256-byte region of the slot; a lane fetches the pair for its two row tiles with one ds_read_u16.
#include <cstdlib>

#include "gemm_epi.h"

namespace vdr {

typedef __attribute__((ext_vector_type(8))) int i32x8;

VDR_DEV void glds4(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

VDR_DEV i32x8 cat_frag(const u32x4& lo, const u32x4& hi) {
  i32x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r[e] = (int)lo[e];
    r[4 + e] = (int)hi[e];
  }
  return r;
}

// (the body takes the LDS base as a plain pointer argument, like the bf16 ring bodies: with the __shared__ array
// referenced directly hipcc puts an s_waitcnt vmcnt(0) -- "LDS-DMA may alias" -- in front of every ds_read)
template <int WAVES_M, int WAVES_N, int NST, int EPI>
VDR_DEV void gemm_mx_body(const GemmK& p, char* smem) {
  constexpr int TM = 2, TN = 2;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  constexpr int UNIT = (BM + BN) * 64;
  constexpr int NA = BM / 16 / NW, NB = BN / 16 / NW;
  constexpr int G = NA + NB + 1;  // + the wave's scale fetch
  constexpr int SC = NST * UNIT;  // scale regions [slot][wave][256 B] behind the ring
  static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile/wave mismatch");
  static_assert((NST - 1) * G <= 63, "vmcnt range");

  const int wg = xcd_remap(blockIdx.x, p.nwg);
  int tm, tn;
  tile_of(p, wg, tm, tn);  // column groups of gn tile columns (launch_mx_cfg), as in the bf16 kernels
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int srow = lane >> 2, spc = lane & 3;
  const char* a_src[NA];
  const char* b_src[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int r = (wave * NA + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    a_src[q] = reinterpret_cast<const char*>(p.A) + gr * p.lda + c * 16;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    // packed weights (GemmArgs::w_interleaved, gemm_kernels.h): [N/2][K/64][2][64 B] -- a 128-B line holds one 64-element
    // K unit of rows 2i and 2i+1, so this wave-instruction (16 rows x 64 B) touches 8 whole lines instead of 16 halves
    b_src[q] = p.w_il ? reinterpret_cast<const char*>(p.W) + (int64_t)(gr >> 1) * ((int64_t)(p.K >> 6) * 128) + (gr & 1) * 64 + c * 16
                      : reinterpret_cast<const char*>(p.W) + (int64_t)gr * p.ldw + c * 16;
  }
  const int bstep = p.w_il ? 128 : 64;  // bytes between consecutive units of one W row
  // scale fetch: lanes 0-31 the activation scales, 32-63 the weight scales; 16 lanes per block plane, 4 bytes
  // = two (r, r+32) row pairs each.  Rows past M / N fall inside the 256-row padding of the scale arrays.
  const uint8_t* s_src;
  int64_t s_step;
  {
    const int pl = (lane >> 4) & 1, q4 = (lane & 15) * 4;
    if (lane < 32) {
      s_src = p.sA + (int64_t)pl * p.sa_rows + (m0 + wm * 64) + q4;
      s_step = 2 * p.sa_rows;
    } else {
      s_src = p.sW + (int64_t)pl * p.sw_rows + (n0 + wn * 64) + q4;
      s_step = 2 * p.sw_rows;
    }
  }

  const int swz = (lane >> 2) & 3;
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (wm * 64 + i * 32 + l31) * 64;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = BM * 64 + (wn * 64 + j * 32 + l31) * 64;
  const int ch0 = ((0 + h) ^ swz) * 16;  // block 0: chunks 0,1
  const int ch1 = ((2 + h) ^ swz) * 16;  // block 1: chunks 2,3
  const int sc_a = SC + wave * 256 + h * 64 + 2 * l31;  // (i = 0, 1) pair of this lane's activation rows
  const int sc_w = sc_a + 128;                          // (j = 0, 1) pair of its weight rows

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  const int nsteps = p.K >> 6;
  auto stage = [&](int slot) {
    char* d = smem + slot * UNIT;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      glds16(a_src[q], d + (wave * NA + q) * 1024);
      a_src[q] += 64;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      glds16(b_src[q], d + BM * 64 + (wave * NB + q) * 1024);
      b_src[q] += bstep;
    }
    glds4(s_src, smem + SC + slot * (NW * 256) + wave * 256);
    s_src += s_step;
  };
  // NOTE the LDS reads are typed bf16x8 on purpose: with an integer-typed ds_read hipcc's waitcnt pass assumes it
  // may alias the LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front of every read (type-based alias
  // info is all that separates them); same for the scale read below (read as bf16, bits reinterpreted)
  i32x8 fa[TM], fb[TN];  // 32-byte operands: registers 0-3 = block 0 bytes, 4-7 = block 1 bytes
  int sa2 = 0, sw2 = 0;
  auto read_unit = [&](int slot) {
    const char* c0 = smem + slot * UNIT;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[j] = cat_frag(__builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(c0 + b_off[j] + ch0)),
                       __builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(c0 + b_off[j] + ch1)));
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[i] = cat_frag(__builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(c0 + a_off[i] + ch0)),
                       __builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(c0 + a_off[i] + ch1)));
    sa2 = __builtin_bit_cast(uint16_t, *reinterpret_cast<const bf16_t*>(smem + slot * (NW * 256) + sc_a));
    sw2 = __builtin_bit_cast(uint16_t, *reinterpret_cast<const bf16_t*>(smem + slot * (NW * 256) + sc_w));
  };
  auto retire = [&](int u, int issued_upto) {
    const int younger = issued_upto - u;
    if (younger >= NST - 1) {
      wait_vmcnt<(NST - 1) * G>();
    } else if (younger == NST - 2 && NST >= 3) {
      wait_vmcnt<(NST - 2) * G>();
    } else if (younger == 1 && NST >= 4) {
      wait_vmcnt<G>();
    } else {
      wait_vmcnt<0>();
    }
  };

  int issued = -1;
#pragma unroll
  for (int u = 0; u < NST; ++u)
    if (u < nsteps) {
      stage(u);
      issued = u;
    }
  retire(0, issued);
  __builtin_amdgcn_s_barrier();
  read_unit(0);
  int slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int nslot = slot + 1 == NST ? 0 : slot + 1;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): fragments and scales of unit s are in registers
    // first operand = weight tile (output column block j), second = activation tile (row block i): a lane owns an
    // output row (token) and 4 consecutive columns per register group, as in the bf16 kernels
    acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[0], fa[0], acc[0][0],
                                                                0, 0, 0, sw2, 0, sa2);
    acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[0], fa[1], acc[0][1],
                                                                0, 0, 0, sw2, 1, sa2);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nsteps) {
      retire(s + 1, issued);
      __builtin_amdgcn_s_barrier();  // every wave has unit s in registers: its slot is free, unit s+1 has landed
      if (s + NST < nsteps) {
        stage(slot);
        issued = s + NST;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[1], fa[0], acc[1][0],
                                                                0, 0, 1, sw2, 0, sa2);
    acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[1], fa[1], acc[1][1],
                                                                0, 0, 1, sw2, 1, sa2);
    __builtin_amdgcn_sched_barrier(0);
    read_unit(nslot);  // lands under the MFMAs just issued (unconditional: after the last unit it fetches
                       // stale, unused bytes -- a conditional read costs a second fragment register set)
    slot = nslot;
  }

  __syncthreads();  // every wave is done with the ring: its memory becomes the epilogue staging area
  epilogue_lds<EPI, TM, TN>(p, acc, smem + wave * (32 * 272), m0 + wm * 64, n0 + wn * 64, lane);
}

template <int WAVES_M, int WAVES_N, int NST, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 4) void gemm_mx_kernel(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_mx_body<WAVES_M, WAVES_N, NST, EPI>(p, smem);
}

template <int WAVES_M, int WAVES_N, int NST>
static hipError_t launch_mx_cfg(const GemmArgs& a, int epi, hipStream_t s) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64, NW = WAVES_M * WAVES_N;
  GemmK k{};
  k.A = (const bf16_t*)a.A;
  k.W = (const bf16_t*)a.W;
  k.w_il = a.w_interleaved;
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
  k.tiles_m = (int)tiles_m;
  {
    // tile order: about 1.7 MB of W payload per column group (gemm_kernels.h, launch_cfg; an e4m3 panel is BN x K bytes)
    int gn = (int)((1700u << 10) / ((size_t)BN * a.K));
    if (gn < 2 || gn >= k.tiles_n) gn = 0;
    k.gn = gn;
#ifdef VDR_TUNING
    if (const char* e = getenv("VDR_MX_GN")) k.gn = atoi(e);
#endif
  }
  const int64_t nwg = tiles_m * k.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffff) return hipErrorInvalidValue;
  k.nwg = (int)nwg;
  k.ln_part = a.ln_part;
  k.part_stride = a.part_stride;
  k.sA = (const uint8_t*)a.a_scale;
  k.sW = (const uint8_t*)a.w_scale;
  k.sa_rows = mx_rows_pad(a.M);
  k.sw_rows = mx_rows_pad(a.N);
  k.sC = (uint8_t*)a.c_scale;
  k.sc_rows = mx_rows_pad(a.M);
  const size_t ring = (size_t)NST * ((size_t)(BM + BN) * 64 + (size_t)NW * 256);
  const size_t stg = (size_t)NW * 32 * 272;
  const size_t lds = ring > stg ? ring : stg;
  const dim3 grid((unsigned)nwg), block(NW * 64);
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
#define VDR_LAUNCH_MX(E)                                                                                          \
  case E: {                                                                                                       \
    auto fn = gemm_mx_kernel<WAVES_M, WAVES_N, NST, E>;                                                           \
    static PerDeviceFlag attr; /* per instantiation and device: lds is a compile-time constant of it */           \
    if (lds > 65536 && !attr.done[dev]) {                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                              \
      attr.done[dev] = true;                                                                                      \
    }                                                                                                             \
    hipLaunchKernelGGL(fn, grid, block, lds, s, k);                                                               \
    break;                                                                                                        \
  }
  switch (epi) {
    VDR_LAUNCH_MX(EPI_BIAS)
    VDR_LAUNCH_MX(EPI_BIAS_GELU)
    VDR_LAUNCH_MX(EPI_BIAS_RESID)
    VDR_LAUNCH_MX(EPI_SWIGLU)
    VDR_LAUNCH_MX(EPI_BIAS_GELU_MX)
    VDR_LAUNCH_MX(EPI_SWIGLU_MX)
    default:
      return hipErrorInvalidValue;
  }
#undef VDR_LAUNCH_MX
  return hipGetLastError();
}

hipError_t launch_gemm_mx(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
  if (a.K <= 0 || (a.K & 63) || (a.N & 63) || a.M <= 0 || !a.a_scale || !a.w_scale) return hipErrorInvalidValue;
  if (a.ln_stats || a.win_ws || a.a_rpg || a.out_f32 || a.M >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
  if (a.c_scale) {  // MX output
    if (epilogue == EPI_BIAS_GELU) epilogue = EPI_BIAS_GELU_MX;
    else if (epilogue == EPI_SWIGLU) epilogue = EPI_SWIGLU_MX;
    else return hipErrorInvalidValue;
  }
  switch (variant) {
    case 0:
      return launch_mx_cfg<2, 4, 3>(a, epilogue, s);  // 128x256, 8 waves, 3 x 26 KB, 2 workgroups per CU
    case 1:
      return launch_mx_cfg<4, 4, 3>(a, epilogue, s);  // 256x256, 16 waves, 3 x 36 KB
    case 2:
      return launch_mx_cfg<2, 2, 3>(a, epilogue, s);  // 128x128, 4 waves, 3 x 17 KB, 3 workgroups per CU
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace vdr
