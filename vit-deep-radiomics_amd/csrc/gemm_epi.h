// Shared by gemm.hip (bf16) and gemm_mx.hip (MX-fp8): the kernel argument block and the fused epilogues
// (bias, LayerNorm fold, erf-GELU, SwiGLU, residual + LayerScale, position add, window un-partition,
// LayerNorm partial sums, MX-fp8 re-quantisation of the output) applied to fp32 MFMA accumulators.
#pragma once
#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

// diagnostic ablation bits exist in tuning builds only (-DVDR_TUNING, `make tuning`, used by tools/): the shipped
// library compiles them out
#ifdef VDR_TUNING
#define VDR_ABL(p, bit) (__builtin_amdgcn_readfirstlane((p).abl & (bit)) != 0)
#else
#define VDR_ABL(p, bit) false
#endif

struct GemmK {
  const bf16_t* A;
  const bf16_t* W;
  int w_il = 0;  // W is in the pair-interleaved layout [N/2][K/32][2][32] (gemm_kernels.h)
  const float* bias;
  const bf16_t* resid;
  const float* resid32 = nullptr;  // EPI_BIAS_RESID32: the fp32 residual stream (read) ...
  float* C32 = nullptr;            // ... and (written) next to its bf16 copy C
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
  int tiles_m = 0, gn = 0;  // gn > 0: tiles are walked in column groups of gn tile columns (see tile_of)
  int nwg;
  const float* ln_stats;
  const float* colsum;
  // LayerNorm fold with the (mean, rstd) finalisation inside the GEMM (ring3 kernels): the producers' (sum, sumsq)
  // partials [ln_groups][ln_cstride][2]; every workgroup turns those of its BM rows into (mean, rstd) in LDS at
  // stats_off while its ring fills (same arithmetic as ln_finalize_kernel: double, groups in order)
  const float* ln_cpart = nullptr;
  int ln_groups = 0;
  int64_t ln_cstride = 0;
  float ln_inv_d = 0.0f, ln_eps = 0.0f;
  int stats_off = 0;
  int ln_fold = 0;  // the consumer-side fold is on (statistics from ln_stats or from ln_cpart)
  float* ln_part;
  int64_t part_stride;
  // the last workgroup to add its partial sums to a block of tile rows finalises that block's (mean, rstd) (ring4 kernels,
  // residual epilogue; finalize_rows_if_last in gemm_kernels.h); null = off
  float* fin_stats = nullptr;
  uint32_t* fin_cnt = nullptr;
  int fin_groups = 0;
  float fin_inv_d = 0.0f, fin_eps = 0.0f;
  int a_rpg;
  int64_t a_gs, a_is;
  int out_f32;
  int nt_store = 0;  // large write-once outputs (fc1's activation, qkv): non-temporal stores keep them from evicting the
                     // A / W panels the main loops of the resident workgroups stream from L2
  // MX-fp8 operands (gemm_mx.hip): A / W point at e4m3 payloads, lda / ldw are in bytes
  const uint8_t* sA = nullptr;
  const uint8_t* sW = nullptr;
  int64_t sa_rows = 0, sw_rows = 0;  // padded row counts of the scale arrays
  uint8_t* sC = nullptr;             // MX output: C is an e4m3 payload [M][ldc], sC its scales
  int64_t sc_rows = 0;
  int win_ws, win_g;  // SAM window un-partition of the output rows (0 = off)
  // im2col-free patchify (ring4, EPI_PATCH): A is a bf16 NCHW image batch and row m / column k of the GEMM operand are
  // token (b, py, px) / (channel, ky, kx) of Conv2d(C, D, p, stride p); pg_ps = log2(p) (p in {8, 16, 32}), 0 = off
  int pg_ps = 0, pg_g = 0, pg_C = 0;
#ifdef VDR_GEMM_STAMPS
  unsigned long long* stamps = nullptr;  // tools/micro/gemm_stamps.hip: [workgroup][wave][8] s_memtime / s_memrealtime stamps
#endif
  int abl = 0;  // tuning builds: ablation bits, 1 = skip the epilogue, 4 = skip global loads after the first units, 8 = skip the output stores
};

// value of lane i + N inside the 16-lane DPP row (0 past the row end)
template <int N>
VDR_DEV float dpp_row_shl(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

// Epilogue math for 8 consecutive output columns n..n+7 of output row m (u = SwiGLU gate partner): residual read and
// output store are 16 bytes per lane
// (a store wave-instruction costs ~80-100 cycles of the CU's store path whatever its width, so the
// epilogue is written with the widest ones).
typedef __attribute__((ext_vector_type(8))) float f32x8;

// Per-column epilogue constants of one octet (bias, LayerNorm-fold column sums, LayerScale): they depend on the
// column only, so epilogue_lds fetches them once per 64-column block instead of once per row step.
struct EpiCols {
  f32x4 b0, b1, c0, c1, g0, g1;
  bool have = false;
};
VDR_DEV EpiCols load_epi_cols(const GemmK& p, int n) {
  EpiCols e;
  e.have = true;
  const int nn = n < p.N ? n : 0;  // out-of-range octets are never stored; keep the address valid
  const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
  e.b0 = e.b1 = e.c0 = e.c1 = z;
  e.g0 = e.g1 = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
  if (p.bias) {
    e.b0 = *reinterpret_cast<const f32x4*>(p.bias + nn);
    e.b1 = *reinterpret_cast<const f32x4*>(p.bias + nn + 4);
  }
  if (p.ln_fold) {
    e.c0 = *reinterpret_cast<const f32x4*>(p.colsum + nn);
    e.c1 = *reinterpret_cast<const f32x4*>(p.colsum + nn + 4);
  }
  if (p.gamma) {
    e.g0 = *reinterpret_cast<const f32x4*>(p.gamma + nn);
    e.g1 = *reinterpret_cast<const f32x4*>(p.gamma + nn + 4);
  }
  return e;
}
// EPI_*_MX (gemm_mx.hip only): the same epilogue math, output re-quantised to MX-fp8
constexpr int epi_base(int e) {
  return e == EPI_BIAS_GELU_MX ? EPI_BIAS_GELU : e == EPI_SWIGLU_MX ? EPI_SWIGLU : e == EPI_BIAS_RESID32 ? EPI_BIAS_RESID : e;
}
constexpr bool epi_mx_out(int e) { return e == EPI_BIAS_GELU_MX || e == EPI_SWIGLU_MX; }
template <int EPI>
VDR_DEV int64_t epi_oct(const GemmK& p, float (&v)[8], float (&u)[8], int64_t m, int n, float& sum1, float& sum2,
                        float mu = 0.0f, float rs = 1.0f, const EpiCols& ec = EpiCols()) {
  constexpr int E = epi_base(EPI);
  constexpr bool MXO = epi_mx_out(EPI);
  (void)MXO;
  sum1 = 0.0f;
  sum2 = 0.0f;
  if (m >= p.M || n >= p.N) return -1;
  int64_t orow = m;
  int prow = 0;
  if (E == EPI_PATCH) {
    const int64_t g = m / p.rpg;
    const int i = (int)(m - g * p.rpg);
    orow = g * p.gstride + p.off + i;
    prow = p.off + i;
  }
  if (p.win_ws > 0) {
    // window un-partition (segment_anything window_unpartition): row m of the windowed order -> token (y, x)
    const int ws = p.win_ws, g = p.win_g, nw = (g + ws - 1) / ws;
    const int64_t widx = m / (ws * ws);
    const int wtok = (int)(m - widx * (ws * ws));
    const int wx = (int)(widx % nw);
    const int64_t t2 = widx / nw;
    const int wy = (int)(t2 % nw);
    const int64_t b = t2 / nw;
    const int y = wy * ws + wtok / ws, x = wx * ws + wtok % ws;
    if (y >= g || x >= g) return -1;  // zero padding of the border windows: dropped
    orow = (b * g + y) * g + x;
  }
  if (E != EPI_BIAS_RESID && p.ln_fold) {
    // LayerNorm folded into this GEMM: acc = x.W'^T with W' = W.diag(gamma); the row statistics and
    // the column sums of W' turn it into LN(x).W^T; the beta term is already inside p.bias
    f32x4 c0 = ec.c0, c1 = ec.c1;
    if (!ec.have) {
      c0 = *reinterpret_cast<const f32x4*>(p.colsum + n);
      c1 = *reinterpret_cast<const f32x4*>(p.colsum + n + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = rs * (v[e] - mu * c0[e]);
      v[4 + e] = rs * (v[4 + e] - mu * c1[e]);
    }
    if (E == EPI_SWIGLU) {
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(p.colsum + n + 32);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(p.colsum + n + 36);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u[e] = rs * (u[e] - mu * d0[e]);
        u[4 + e] = rs * (u[4 + e] - mu * d1[e]);
      }
    }
  }
  // (with the per-block constants at hand the bias is added unconditionally -- zeros when there is none: as
  // `p.bias ? v + b : v` hipcc emits the add AND a v_cndmask per value)
  if (ec.have || p.bias) {
    f32x4 b0 = ec.b0, b1 = ec.b1;
    if (!ec.have) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] += b0[e];
      v[4 + e] += b1[e];
    }
  }
  if (E == EPI_BIAS_GELU) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x2 g = gelu_erf2(f32x2{v[e], v[e + 1]});
      v[e] = g[0];
      v[e + 1] = g[1];
    }
  }
  if (E == EPI_SWIGLU) {
    if (p.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n + 32);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 36);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u[e] += b0[e];
        u[4 + e] += b1[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu(v[e]) * u[e];
  }
  if (E == EPI_BIAS_RESID) {
    if (ec.have || p.gamma) {  // (LayerScale: ones when there is none, see the bias above)
      f32x4 g0 = ec.g0, g1 = ec.g1;
      if (!ec.have) {
        g0 = *reinterpret_cast<const f32x4*>(p.gamma + n);
        g1 = *reinterpret_cast<const f32x4*>(p.gamma + n + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] *= g0[e];
        v[4 + e] *= g1[e];
      }
    }
    const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.resid + orow * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
  }
  if (E == EPI_PATCH) {
    if (p.pos) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + n);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += p0[e];
        v[4 + e] += p1[e];
      }
    }
  }
  if (p.out_f32) {
    float* dst = reinterpret_cast<float*>(p.C) + orow * p.ldc + n;
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = v[e];
      hi[e] = v[4 + e];
    }
    *reinterpret_cast<f32x4*>(dst) = lo;
    *reinterpret_cast<f32x4*>(dst + 4) = hi;
    return orow;
  }
  if constexpr (MXO) {
    // MX-fp8 output (the operand of the next fp8 linear): the 4 lanes that hold 32 consecutive output columns of
    // this row agree on one e8m0 scale (validity is uniform inside such a quad: N % 64 == 0), each stores 8 bytes
    int oc8 = n;
    if (E == EPI_SWIGLU) oc8 = (n >> 6) * 32 + (n & 31);
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const float t = amax * (1.0f / 448.0f);
    const uint32_t tb = __float_as_uint(t);
    int ex = (int)((tb >> 23) & 255) - 127 + ((tb & 0x7fffff) ? 1 : 0);
    ex = ex < -126 ? -126 : (ex > 126 ? 126 : ex);
    const float inv = __uint_as_float((uint32_t)(127 - ex) << 23);
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, w1, true);
    u32x2 o8;
    o8[0] = (uint32_t)w0;
    o8[1] = (uint32_t)w1;
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(p.C) + orow * p.ldc + oc8) = o8;
    if ((oc8 & 31) == 0)
      p.sC[(int64_t)(oc8 >> 5) * p.sc_rows + (orow & ~(int64_t)63) + 2 * (orow & 31) + ((orow >> 5) & 1)] =
          (uint8_t)(ex + 127);
    return orow;
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o[e] = (bf16_t)v[e];
    const float r = (float)o[e];  // statistics of what the consumer will actually read
    sum1 += r;
    sum2 = fmaf(r, r, sum2);
  }
  int oc = n;
  if (E == EPI_SWIGLU) oc = (n >> 6) * 32 + (n & 31);
  if (VDR_ABL(p, 8)) {  // diagnostic (tuning builds): the whole epilogue except the global store (the value stays live through a never-true test)
    if (o[0] == (bf16_t)12345.0f && o[7] == (bf16_t)-54321.0f) *reinterpret_cast<bf16x8*>(p.C + orow * p.ldc + oc) = o;
    return orow;
  }
  if (p.nt_store) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p.C + orow * p.ldc + oc));
  else *reinterpret_cast<bf16x8*>(p.C + orow * p.ldc + oc) = o;
  return orow;
}

// LDS-staged epilogue: the accumulators of a 32 (m) x 64 (n) block go through a wave-private fp32
// staging image (row stride 272 B: conflict-free ds_write_b128 / ds_read_b128) and come back with
// 16 lanes per output row, so every global access of the epilogue (bias, residual, store) touches
// whole 128-B lines instead of 32 rows x 16 B per instruction.
// staging of one 32 (m) x 64 (n) block (row block i, column-pair block jp) of a wave's accumulators as fp32
// [32][64] with row stride 272 B, for the two accumulator layouts:
//   32x32x16 MFMA: acc[j][i][4g + e] = D[n = 32j + 8g + 4h + e][m = 32i + (lane & 31)]
template <int TM, int TN>
VDR_DEV void stage_acc_block(const f32x16 (&acc)[TN][TM], char* stg, int i, int jp, int lane) {
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = acc[2 * jp + jj][i][4 * g + e];
      *reinterpret_cast<f32x4*>(stg + l31 * 272 + (jj * 32 + 8 * g + 4 * h) * 4) = t;
    }
}
//   16x16x32 MFMA (64 x 64 wave tile = 4 x 4 tiles): t[jt][it][e] = D[n = 16jt + 4(lane >> 4) + e][m = 16it + (lane & 15)]
struct Acc16 {
  f32x4 t[4][4];
};
template <int TM, int TN>
VDR_DEV void stage_acc_block(const Acc16& acc, char* stg, int i, int jp, int lane) {
  static_assert(TM == 2 && TN == 2, "Acc16 is a 64 x 64 wave tile");
  (void)jp;
#pragma unroll
  for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      *reinterpret_cast<f32x4*>(stg + (it2 * 16 + (lane & 15)) * 272 + (jt * 16 + 4 * (lane >> 4)) * 4) = acc.t[jt][2 * i + it2];
}

template <int TM, int TN, bool F32, typename AccT>
VDR_DEV void epilogue_resid(const GemmK& p, const AccT& acc, char* stg, int64_t m_base, int n_base, int lane);
VDR_DEV bool epilogue_resid_ok(const GemmK& p);

template <int EPI, int TM, int TN, typename AccT>
VDR_DEV void epilogue_lds(const GemmK& p, const AccT& acc, char* stg, int64_t m_base, int n_base, int lane,
                          const float2* tile_stats = nullptr) {  // tile_stats: LDS (mean, rstd) of row m_base onwards
  constexpr int E = epi_base(EPI);
  constexpr bool MXO = epi_mx_out(EPI);
  (void)MXO;
  constexpr int RS = 272;
  static_assert(TN % 2 == 0, "column tiles are staged in pairs");
  constexpr bool FOLD = E != EPI_BIAS_RESID;  // (a residual GEMM never consumes a LayerNorm: refused at launch)
  if constexpr (E == EPI_BIAS_RESID) {
    // the residual form has its own straight-line epilogue; launches it cannot take (fp32 output, a consumer-side
    // LayerNorm fold, 2^31 rows) are refused on the host (resid_launch_ok), so the general code below is not even
    // compiled into the residual kernels -- except for the no-store diagnostic of the tuning builds
    if (epilogue_resid_ok(p)) {
      epilogue_resid<TM, TN, EPI == EPI_BIAS_RESID32>(p, acc, stg, m_base, n_base, lane);
      return;
    }
#ifndef VDR_TUNING
    return;
#endif
  }
  // LayerNorm fold: (mean, rstd) of the 4*TM rows this lane owns in the read-back phase, fetched up front
  float st_mu[TM][4], st_rs[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      st_mu[i][rr] = 0.0f;
      st_rs[i][rr] = 1.0f;
      if (FOLD && tile_stats) {
        const float2 t = tile_stats[i * 32 + rr * 8 + (lane >> 3)];
        st_mu[i][rr] = t.x;
        st_rs[i][rr] = t.y;
      } else if (FOLD && p.ln_stats) {
        int64_t m = m_base + i * 32 + rr * 8 + (lane >> 3);
        m = m < p.M ? m : p.M - 1;
        const float2 t = *reinterpret_cast<const float2*>(p.ln_stats + 2 * m);
        st_mu[i][rr] = t.x;
        st_rs[i][rr] = t.y;
      }
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int jp = 0; jp < TN / 2; ++jp) {
      stage_acc_block<TM, TN>(acc, stg, i, jp, lane);
      if (E != EPI_SWIGLU) {
        // 8 lanes per row (8 columns each), 8 rows per instruction: whole 128-B lines, 16-B accesses
        const EpiCols ec = load_epi_cols(p, n_base + jp * 64 + (lane & 7) * 8);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = rr * 8 + (lane >> 3), c8 = lane & 7;
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 32);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 32 + 16);
          float v[8], u[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = u[e] = t0[e];
            v[4 + e] = u[4 + e] = t1[e];
          }
          float s1, s2;
          const int64_t orow = epi_oct<EPI>(p, v, u, m_base + i * 32 + row, n_base + jp * 64 + c8 * 8, s1, s2,
                                            st_mu[i][rr], st_rs[i][rr], ec);
          if (p.ln_part) {
            // the 8 lanes of a row hold its 64 columns of this block: (sum, sumsq) -> one slot per
            // (row, 64-column group), written exactly once: no atomics, no zeroing, deterministic
            // lane c8 == 0 of each 8-lane group collects the group: three DPP row-shift adds (lane i += lane i+4, +2,
            // +1 inside its 16-lane row) instead of three ds_bpermute round trips per value
            s1 += dpp_row_shl<4>(s1);
            s2 += dpp_row_shl<4>(s2);
            s1 += dpp_row_shl<2>(s1);
            s2 += dpp_row_shl<2>(s2);
            s1 += dpp_row_shl<1>(s1);
            s2 += dpp_row_shl<1>(s2);
            const int grp = (n_base + jp * 64) >> 6;
            if (c8 == 0 && orow >= 0 && n_base + jp * 64 < p.N) {
              float* dst = p.ln_part + ((int64_t)grp * p.part_stride + orow) * 2;
              dst[0] = s1;
              dst[1] = s2;
            }
          }
        }
      } else {
        // gate pairs: columns 0..31 of the block are x1, 32..63 the matching x2 -> 32 outputs per row
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int row = rr * 16 + (lane >> 2), c8 = lane & 3;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 32);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 32 + 16);
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(stg + row * RS + 128 + c8 * 32);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(stg + row * RS + 128 + c8 * 32 + 16);
          float v[8], u[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = a0[e];
            v[4 + e] = a1[e];
            u[e] = g0[e];
            u[4 + e] = g1[e];
          }
          float s1, s2, mu = 0.0f, rs = 1.0f;
          if (tile_stats) {
            const float2 t = tile_stats[i * 32 + row];
            mu = t.x;
            rs = t.y;
          } else if (p.ln_stats) {
            int64_t m = m_base + i * 32 + row;
            m = m < p.M ? m : p.M - 1;
            const float2 t = *reinterpret_cast<const float2*>(p.ln_stats + 2 * m);
            mu = t.x;
            rs = t.y;
          }
          epi_oct<EPI>(p, v, u, m_base + i * 32 + row, n_base + jp * 64 + c8 * 8, s1, s2, mu, rs);
        }
      }
    }
  }
}

// bf16-staged epilogue for the write-once outputs (EPI_BIAS = qkv, EPI_BIAS_GELU = fc1: 232 and 310 MB per launch at
// the headline shapes).  The whole epilogue arithmetic (LayerNorm fold, bias, erf-GELU) runs in the ACCUMULATOR layout
// -- lane (r = lane & 15, q = lane >> 4) owns rows 16 it + r and columns 16 jt + 4 q + e of the 64 x 64 wave tile, so the
// per-column constants are 4 x f32x4 and the per-row statistics 4 pairs per lane -- and only the final bf16 values go
// through LDS: a wave-private [64 rows][128 B] image, 8-byte writes, 16-byte read-back with 8 lanes per row, whole
// 128-B lines to memory.  Half the LDS bytes of the fp32 staging of epilogue_lds (the LDS write port, ~80 B/clk per
// CU, is what that staging waits for).  16-B slot s of row r sits at slot s ^ ((r >> 1) & 7): conflict-free
// read-back, 2-way on the writes (rows 2k / 2k+1), which an 8-byte write hides.
// Epilogues that add a residual keep the fp32 staging: the sum has to be rounded once, after the add.
// The per-column constants (bias, column sums of the folded weight) and per-row statistics of epilogue_bf16, in the
// accumulator layout.  Fetched by the ring4 body right after its last MFMA is issued -- into the registers the operand
// fragments have just left -- so that their L2 round trip runs under the drain of the matrix pipe and the barrier that
// frees the ring instead of at the head of the epilogue.
struct EpiPre {
  f32x4 bias[4], csum[4];
  float2 stats[4];
};
VDR_DEV EpiPre epilogue_bf16_prefetch(const GemmK& p, int64_t m_base, int n_base, int lane) {
  EpiPre e;
  const int r15 = lane & 15, q4 = lane >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    int n = n_base + jt * 16 + 4 * q4;
    n = n < p.N ? n : 0;  // out-of-range columns are never stored; keep the address valid
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    e.bias[jt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : z;
    e.csum[jt] = p.ln_fold ? *reinterpret_cast<const f32x4*>(p.colsum + n) : z;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    e.stats[it] = float2{0.0f, 1.0f};
    if (p.ln_stats) {
      int64_t m = m_base + it * 16 + r15;
      m = m < p.M ? m : p.M - 1;
      e.stats[it] = *reinterpret_cast<const float2*>(p.ln_stats + 2 * m);
    }
  }
  return e;
}

template <int EPI>
VDR_DEV void epilogue_bf16(const GemmK& p, const Acc16& acc, char* stg, int64_t m_base, int n_base, int lane,
                           const float2* tile_stats, const EpiPre& here) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU, "write-once outputs only");
  const int r15 = lane & 15, q4 = lane >> 4;
  const f32x4 (&bias)[4] = here.bias;
  const f32x4 (&csum)[4] = here.csum;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    float mu = here.stats[it].x, rs = here.stats[it].y;
    if (tile_stats) {
      const float2 t = tile_stats[it * 16 + r15];
      mu = t.x;
      rs = t.y;
    }
    const int row = it * 16 + r15;
    // rs (acc - mu c) + b  =  rs acc + (b - rs mu c): two fused multiply-adds per value.  Without the fold the same two
    // with rs = 1, mu = 0, c = 0 give acc + b exactly (1 * acc is exact, 0 * 0 + b = b): written as one formula so that
    // hipcc does not evaluate both forms and select per element (an add and a v_cndmask per value in a VALU-bound epilogue)
    const float nrm = -rs * mu;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      bf16x4 o;
      // (on pairs of neighbouring columns: v_pk_fma_f32 is two values per issue slot, bit for bit the scalar operations)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 a2 = {acc.t[jt][it][e], acc.t[jt][it][e + 1]};
        const f32x2 c2 = {csum[jt][e], csum[jt][e + 1]}, b2 = {bias[jt][e], bias[jt][e + 1]};
        f32x2 v = __builtin_elementwise_fma(f32x2{rs, rs}, a2, __builtin_elementwise_fma(f32x2{nrm, nrm}, c2, b2));
        if (EPI == EPI_BIAS_GELU) v = gelu_erf2(v);
        o[e] = (bf16_t)v[0];
        o[e + 1] = (bf16_t)v[1];
      }
      const int slot = (2 * jt + (q4 >> 1)) ^ ((row >> 1) & 7);
      *reinterpret_cast<bf16x4*>(stg + row * 128 + slot * 16 + (q4 & 1) * 8) = o;
    }
  }
  // read-back: 8 lanes per row, 8 rows per instruction
  const int c8 = lane & 7;
  const int n = n_base + c8 * 8;
  if (VDR_ABL(p, 8)) {  // diagnostic (tuning builds): everything but the stores
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = rr * 8 + (lane >> 3);
      const bf16x8 o = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((c8 ^ ((row >> 1) & 7)) * 16));
      const int64_t m = m_base + row;
      if (m < p.M && n < p.N && o[0] == (bf16_t)12345.0f && o[7] == (bf16_t)-54321.0f)
        *reinterpret_cast<bf16x8*>(p.C + m * p.ldc + n) = o;
    }
    return;
  }
  // The stores are unconditional buffer stores relative to the wave tile's first row: the resource ends after the
  // tile's last valid row (rows past M are dropped by the hardware's range check), lanes past N carry an out-of-range
  // offset, and the 8-row step is a scalar offset -- per step one ds_read_b128 and one buffer_store, no address
  // arithmetic, no exec-masked branch (as `if (m < M && n < N) *dst = o` a step was ~25 instructions: two 64-bit
  // multiplies, compare, saveexec, branch, and a wait for its own LDS read that kept the 8 steps in a row).
  int64_t mb;
  {
    const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)m_base);
    const int hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)m_base >> 32));
    mb = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  }
  const int nb = __builtin_amdgcn_readfirstlane(n_base);
  const int64_t left = p.M - mb;
  const int valid = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u;  // (ldc < 2^24: checked at launch)
  const __amdgpu_buffer_rsrc_t rc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + mb * p.ldc + nb), 0, (int)((uint32_t)valid * ldc2), 0x00020000);
  const uint32_t voff = n < p.N ? (uint32_t)(lane >> 3) * ldc2 + (uint32_t)c8 * 16u : 0x7fffffffu;
  bf16x8 o[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int row = rr * 8 + (lane >> 3);
    o[rr] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((c8 ^ ((row >> 1) & 7)) * 16));
  }
  if (p.nt_store) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[rr]), rc, voff, (uint32_t)(rr * 8) * ldc2, 2 /* nt */);
  } else {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[rr]), rc, voff, (uint32_t)(rr * 8) * ldc2, 0);
  }
}

// Straight-line residual epilogue: EPI_BIAS_RESID with a bf16 output, x = resid + gamma * (acc + bias), optionally the
// (sum, sumsq) partials of the rounded outputs for the next LayerNorm and SAM's window un-partition of the rows.  Same
// arithmetic, same order, same bits as epi_oct through epilogue_lds.
// Why it exists: in the general path every 8-column group is [residual load -> vmcnt(0) -> arithmetic -> store ->
// vmcnt(0)] -- the output usually IS the residual buffer, the row / column guards are control flow, and hipcc ends
// every group by waiting for its stores -- so a lane's 8 groups are 16 memory round trips in a row (stamped: 35 % of a
// proj tile's life).  Here the residual chunks of a 32 x 64 block are requested ahead of the last store of the block
// before it, nothing is loaded between a block's first use and its stores, and the guards only predicate the stores.
// Two build-time shapes of it, measured on ViT-B batch 256 (bench.py in alternating processes, one box; the general
// epilogue it replaces: 25.40 k img/s, proj 1.127 ms, fc2 2.708 ms per step):
//   GROUP = 8-row steps computed before their stores are issued together, LATE = the next block's residual is
//   requested just before this block's last stores (one buffer) instead of while this block is staged (two buffers)
//   GROUP 1, LATE 1: 25.87 k, proj 0.986, fc2 2.640 (no spills)      <- built
//   GROUP 2, LATE 1: 25.77 k, proj 0.990, fc2 2.646 (2 VGPRs spilt)
//   GROUP 1, LATE 0: 25.28 k, proj 1.118, fc2 2.741 (12 spilt: every reload is a vmcnt(0) in front of a store)
//   GROUP 4, LATE 1: 40 spilt, not run
// Also measured against GROUP 1 / LATE 1 (another box): each 8-row step requesting the same step of the NEXT block into
// the registers it has just finished with (a whole block of distance for every chunk), with every store an
// unconditional buffer store (masked-off lanes carry an out-of-range offset; hipcc's vmcnt then counts exactly instead
// of assuming the exec-masked stores were skipped): no spills, waits of vmcnt(11) instead of vmcnt(1..3) -- proj 0.997
// -> 1.009 ms, fc2 2.684 -> 2.686.  With one block of distance the residual is no longer what the steps wait for.
// And: the first block's residual / bias / LayerScale requested by ring4 before the barrier that ends its main loop (as
// EpiPre does for the write-once outputs): proj 0.892 -> 0.914 ms, fc2 2.296 -> 2.338 (3 VGPRs spilt in the persistent
// kernel at its 128-register budget) -- not kept.
#ifndef VDR_RESID_GROUP
#define VDR_RESID_GROUP 1
#endif
#ifndef VDR_RESID_LATE
#define VDR_RESID_LATE 1
#endif
// F32 (vdr_config.resid_fp32, EPI_BIAS_RESID32): the residual stream has an fp32 master copy -- read from p.resid32, the
// fp32 sum written back to p.C32 and, rounded ONCE from it, to p.C (the operand of the next GEMM; the LayerNorm partials
// stay statistics of that bf16 copy).  The rounding of the stream no longer accumulates over the blocks
// (tools/resid_precision.py: rel-L2 to the fp32 reference arithmetic 1.24e-2 -> 6.6e-3 at 24 blocks, 1.72e-2 -> 1.0e-2 at 40).
template <int TM, int TN, bool WIN, bool F32, typename AccT>
VDR_DEV void epilogue_resid_impl(const GemmK& p, const AccT& acc, char* stg, int64_t m_base, int n_base, int lane) {
  constexpr int RS = 272;
  constexpr int NJ = TN / 2, NB = TM * NJ;  // 32 x 64 blocks of the wave tile
  constexpr int G = VDR_RESID_GROUP;
  constexpr bool LATE = VDR_RESID_LATE != 0;
  constexpr int NR = LATE ? 1 : 2;
  asm volatile("" : "+v"(lane));  // (opaque: keeps the lane's address terms out of the persistent kernel's tile loop)
  const int c8 = lane & 7, r8 = lane >> 3;
  const bool stats = p.ln_part != nullptr;  // wave-uniform
  const int mrow = (int)m_base + r8, M = (int)p.M;  // rows are < 2^31 (checked at launch): 32-bit row arithmetic
  // output row of tile row m (-1: not stored)
  auto out_row = [&](int m) -> int {
    if (m >= M) return -1;
    if constexpr (WIN) {  // segment_anything window_unpartition, as in epi_oct
      const uint32_t ws = p.win_ws, g = p.win_g, nw = (g + ws - 1) / ws, mm = (uint32_t)m;
      const uint32_t widx = mm / (ws * ws), wtok = mm - widx * (ws * ws);
      const uint32_t wx = widx % nw, t2 = widx / nw, wy = t2 % nw, b = t2 / nw;
      const uint32_t y = wy * ws + wtok / ws, x = wx * ws + wtok % ws;
      if (y >= g || x >= g) return -1;  // zero padding of the border windows: dropped
      return (int)((b * g + y) * g + x);
    }
    return m;
  };
  constexpr int NBG = NJ > 1 ? 2 : 1;  // (one column block per wave tile: its bias / gamma are fetched once)
  int orow[WIN ? 2 : 1][4];            // (identity rows are recomputed where they are used)
  bf16x8 rsd[F32 ? 1 : NR][F32 ? 1 : 4];
  f32x4 rsf[F32 ? NR : 1][F32 ? 4 : 1][2];
  f32x4 b0[NBG], b1[NBG], g0[NBG], g1[NBG];
  auto fetch = [&](int blk) {
    const int i = blk / NJ, jp = blk % NJ, s = blk & 1, sb = NJ > 1 ? s : 0;
    const int n = n_base + jp * 64 + c8 * 8;
    const int nn = n < p.N ? n : 0;  // (columns / rows past the end: a valid address, the value is never stored)
    if (NJ > 1 || blk == 0) {
      const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f}, one = {1.0f, 1.0f, 1.0f, 1.0f};
      b0[sb] = b1[sb] = z;
      g0[sb] = g1[sb] = one;
      if (p.bias) {
        b0[sb] = *reinterpret_cast<const f32x4*>(p.bias + nn);
        b1[sb] = *reinterpret_cast<const f32x4*>(p.bias + nn + 4);
      }
      if (p.gamma) {
        g0[sb] = *reinterpret_cast<const f32x4*>(p.gamma + nn);
        g1[sb] = *reinterpret_cast<const f32x4*>(p.gamma + nn + 4);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int o = out_row(mrow + i * 32 + rr * 8);
      if constexpr (WIN) orow[s][rr] = o;
      if constexpr (F32) {
        const float* src = p.resid32 + (int64_t)(o < 0 ? 0 : o) * p.ldr + nn;
        rsf[LATE ? 0 : s][rr][0] = *reinterpret_cast<const f32x4*>(src);
        rsf[LATE ? 0 : s][rr][1] = *reinterpret_cast<const f32x4*>(src + 4);
      } else {
        rsd[LATE ? 0 : s][rr] = *reinterpret_cast<const bf16x8*>(p.resid + (int64_t)(o < 0 ? 0 : o) * p.ldr + nn);
      }
    }
  };
  fetch(0);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int i = blk / NJ, jp = blk % NJ, s = blk & 1, sb = NJ > 1 ? s : 0;
    const int n = n_base + jp * 64 + c8 * 8;
    stage_acc_block<TM, TN>(acc, stg, i, jp, lane);
    if (!LATE && blk + 1 < NB) fetch(blk + 1);  // (this block's accumulators are dead by now; everything up front spills)
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += G) {
      // G 8-row steps are computed before their stores go out together: with a store per step hipcc reuses the
      // data registers and each step waits for the store before it
      bf16x8 o[G];
      f32x4 vf[F32 ? G : 1][2];
      float s1[G], s2[G];
      int om[G];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int rr = r0 + k, row = rr * 8 + r8;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 32);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 32 + 16);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (F32) {
            v[e] = (t0[e] + b0[sb][e]) * g0[sb][e] + rsf[LATE ? 0 : s][rr][0][e];
            v[4 + e] = (t1[e] + b1[sb][e]) * g1[sb][e] + rsf[LATE ? 0 : s][rr][1][e];
            vf[k][0][e] = v[e];
            vf[k][1][e] = v[4 + e];
          } else {
            v[e] = (t0[e] + b0[sb][e]) * g0[sb][e] + (float)rsd[LATE ? 0 : s][rr][e];
            v[4 + e] = (t1[e] + b1[sb][e]) * g1[sb][e] + (float)rsd[LATE ? 0 : s][rr][4 + e];
          }
        }
        s1[k] = s2[k] = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[k][e] = (bf16_t)v[e];
          const float r = (float)o[k][e];  // statistics of what the consumer will actually read
          s1[k] += r;
          s2[k] = fmaf(r, r, s2[k]);
        }
        om[k] = WIN ? orow[WIN ? s : 0][rr] : out_row(mrow + i * 32 + rr * 8);
        if (n >= p.N) om[k] = -1;
      }
      // in place (the usual case) hipcc keeps a load behind every store ahead of it: the one-buffer form requests the
      // next block's residual here, after the last use of this block's and before its last stores
      if (LATE && r0 + G >= 4 && blk + 1 < NB) fetch(blk + 1);
#pragma unroll
      for (int k = 0; k < G; ++k)
        if (om[k] >= 0) {
          *reinterpret_cast<bf16x8*>(p.C + (int64_t)om[k] * p.ldc + n) = o[k];
          if constexpr (F32) {
            float* d32 = p.C32 + (int64_t)om[k] * p.ldc + n;
            *reinterpret_cast<f32x4*>(d32) = vf[k][0];
            *reinterpret_cast<f32x4*>(d32 + 4) = vf[k][1];
          }
        }
      if (stats) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
          float a = om[k] < 0 ? 0.0f : s1[k], b = om[k] < 0 ? 0.0f : s2[k];  // (as epi_oct: a dropped group adds nothing)
          a += dpp_row_shl<4>(a);
          b += dpp_row_shl<4>(b);
          a += dpp_row_shl<2>(a);
          b += dpp_row_shl<2>(b);
          a += dpp_row_shl<1>(a);
          b += dpp_row_shl<1>(b);
          if (c8 == 0 && om[k] >= 0) {
            // (agent-scope stores: written through to where a workgroup on another XCD reads them -- the one that finalises
            // this block of rows, finalize_rows_if_last -- instead of staying in this XCD's L2 until the kernel ends)
            float* dst = p.ln_part + ((int64_t)((n_base + jp * 64) >> 6) * p.part_stride + om[k]) * 2;
            __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  }
}
template <int TM, int TN, bool F32, typename AccT>
VDR_DEV void epilogue_resid(const GemmK& p, const AccT& acc, char* stg, int64_t m_base, int n_base, int lane) {
  if constexpr (F32) {  // (refused at launch with the SAM window map: pre-LN image models only)
    epilogue_resid_impl<TM, TN, false, true>(p, acc, stg, m_base, n_base, lane);
  } else {
    if (p.win_ws > 0) epilogue_resid_impl<TM, TN, true, false>(p, acc, stg, m_base, n_base, lane);
    else epilogue_resid_impl<TM, TN, false, false>(p, acc, stg, m_base, n_base, lane);
  }
}
VDR_DEV bool epilogue_resid_ok(const GemmK& p) { return !VDR_ABL(p, 8); }

// which epilogue a ring3 / ring4 tile takes
template <int EPI>
VDR_DEV void epilogue_tile(const GemmK& p, const Acc16& acc, char* smem, int wave, int64_t m_base, int n_base, int lane,
                           const float2* tile_stats = nullptr) {
  if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
    if (!p.out_f32 && p.win_ws == 0 && !p.ln_part) {
      epilogue_bf16<EPI>(p, acc, smem + wave * 8192, m_base, n_base, lane, tile_stats,
                         epilogue_bf16_prefetch(p, m_base, n_base, lane));
      return;
    }
  }
  epilogue_lds<EPI, 2, 2>(p, acc, smem + wave * (32 * 272), m_base, n_base, lane, tile_stats);
}

// logical tile id -> (tile row, tile column).  gn == 0: row-major (the column tiles of a tile row are neighbours and
// share the A panel; right when the whole W fits the XCD's L2 or K is long).  gn > 0: column groups of gn tile
// columns are swept over all tile rows before the next group starts: the gn W panels of a group stay resident in
// the 4 MB L2 of the XCD while A streams through once per group instead of once per L2 eviction.
VDR_DEV void tile_of(const GemmK& p, int wg, int& tm, int& tn) {
  if (p.gn <= 0 || p.gn >= p.tiles_n) {
    tm = wg / p.tiles_n;
    tn = wg - tm * p.tiles_n;
    return;
  }
  const int per_group = p.tiles_m * p.gn;
  const int g = wg / per_group;
  const int r = wg - g * per_group;
  const int width = min(p.gn, p.tiles_n - g * p.gn);  // the last group may be narrower
  tm = r / width;
  tn = g * p.gn + (r - tm * width);
}

template <int N>
VDR_DEV void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace vdr
