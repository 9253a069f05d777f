// Attention with decomposed relative position bias for the SAM / MedSAM image encoder (the backbone the
// reference loads by default: src/tfds_dense_descriptor.py:93-107, called at :123).
//
//   attn[q, k] = (q . k) * dh^-0.5 + q . Rh[qh - kh] + q . Rw[qw - kw]      (third-party ImageEncoderViT,
//   softmax over k, times v                                                   add_decomposed_rel_pos)
//
// for S x S token grids: the 14 x 14 windows (196 tokens, zero-padded border windows included) and the
// four global blocks over the whole 64 x 64 grid (4096 tokens).
//
// relpos_pack_kernel: both rel-pos tables as one [Npad][64] bf16 GEMM operand; T = q . table^T (fp32, all
//                   2(2S-1) relative offsets per token and head) is a GEMM in gemm.hip
// attn_relpos_kernel<NT, S, MULTI>: the fused attention of attention.hip (K in swizzled LDS, V^T in LDS,
//                   S^T = K.Q^T so a softmax row is lane-local, P fed back as an MFMA operand) with the bias
//                   added to the logits from registers: the key of accumulator element (t, e, half) is a
//                   compile-time constant, so kh = key / S and kw = key % S index register arrays statically.
//                   MULTI (S = 64): online softmax over 128-key chunks = two grid rows per chunk.
#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

// ---------------------------------------------------------------------------------------------------
// rel-pos table packing: [Npad][64] bf16, rel_pos_h rows at 0, rel_pos_w rows at Npad/2, zero elsewhere.
// The products T[(token, head)][j] = q . table[j] are then ONE MFMA GEMM over (token, head) rows
// (launch_gemm with the two-stride A row map), and the attention kernel picks
//   rel_h[kh] = T[qh - kh + S-1],   rel_w[kw] = T[Npad/2 + qw - kw + S-1]
// with per-lane base offsets and compile-time kh / kw.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relpos_pack_kernel(const float* __restrict__ rel_h,
                                                          const float* __restrict__ rel_w,
                                                          bf16_t* __restrict__ table, int S, int npad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npad * 64) return;
  const int row = i >> 6, c = i & 63;
  const int half = npad >> 1, n = 2 * S - 1;
  float v = 0.0f;
  if (row < n) v = rel_h[row * 64 + c];
  else if (row >= half && row < half + n) v = rel_w[(row - half) * 64 + c];
  table[i] = (bf16_t)v;
}

hipError_t launch_relpos_pack(const float* rel_h, const float* rel_w, void* table, int S, hipStream_t s) {
  if (S <= 0) return hipErrorInvalidValue;
  const int npad = relpos_npad(S);
  hipLaunchKernelGGL(relpos_pack_kernel, dim3((unsigned)((npad * 64 + 255) / 256)), dim3(256), 0, s, rel_h, rel_w,
                     (bf16_t*)table, S, npad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
struct AttnRK {
  const bf16_t* qkv;
  const float* T;  // [batch*seq][heads][relpos_npad(S)]
  bf16_t* out;
  int seq, heads;
  int64_t ld_qkv, ld_out;
  int qt_per_block;
  int n_chunks;
};

// Workgroup: 4 waves.  (MULTI with 8 waves = 8 query tiles per workgroup, half the LDS-DMA pieces per wave and half the
// K / V bytes per query, measured SLOWER: 8.05 -> 8.71 ms per MedSAM step at B = 16 -- one 8-wave workgroup per CU loses
// the second workgroup that runs while the first sits at its chunk barrier.)
template <int NT, int S, bool MULTI>
__global__ __launch_bounds__(256, 2) void attn_relpos_kernel(AttnRK p) {
  constexpr int KEYS = NT * 32;
  constexpr int NW = 4;                   // waves
  constexpr int PPW = NT * 4 / NW;        // K (and V) pieces of 8 rows per wave and image
  static_assert(NT * 4 % NW == 0 || !MULTI, "pieces must divide among the waves");
  static_assert(!MULTI || (S == 64 && NT == 4), "the chunked form assumes 128-key chunks = two rows of a 64-wide grid");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF = 2 * KEYS * 128;  // K image + V image; the chunked (MULTI) form ping-pongs between two of them

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int l31 = lane & 31;
  const int swz = (lane >> 1) & 7;
  // transposed V read (see attention.hip): lane 4q+p of a 16-lane group addresses key row q, d columns 4p..4p+3
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const int tq = (lane & 15) >> 2, tp = lane & 3, dg = (lane >> 4) & 1;
  const int vkey = 4 * hh + tq;
  const int vbase = KEYS * 128 + vkey * 128 + 8 * (tp & 1);  // byte offset of this lane's V address inside a buffer
  int vch[2];
#pragma unroll
  for (int nd = 0; nd < 2; ++nd) vch[nd] = ((4 * nd + 2 * dg + (tp >> 1)) ^ (((vkey >> 1) & 1) << 2)) * 16;

  // 1-D grid of (window or grid, head) x query blocks in XCD-contiguous order (see attention.hip: attn_kernel).  The
  // 32 query blocks of one global-attention head now share their XCD's L2 for its 1 MB of K / V instead of each
  // streaming it from HBM: 6.4 GB per launch at B = 16 before.
  const int nqt = (p.seq + 31) >> 5;
  const int nyb = (nqt + p.qt_per_block - 1) / p.qt_per_block;
  const int vid = nyb > 1 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bh = vid / nyb;
  const int yb = vid - bh * nyb;
  const int b = bh / p.heads;
  const int hd = bh - b * p.heads;
  const int HD = p.heads * 64;
  const bf16_t* qb = p.qkv + (int64_t)b * p.seq * p.ld_qkv + hd * 64;
  const bf16_t* kb = qb + HD;
  const bf16_t* vb = qb + 2 * HD;
  bf16_t* ob = p.out + (int64_t)b * p.seq * p.ld_out + hd * 64;

  const int qt_begin = yb * p.qt_per_block;
  const int qt_end = min(nqt, qt_begin + p.qt_per_block);
  constexpr float LOG2E = 1.44269504088896341f;

  f32x16 o[2];
  float m_run = -INFINITY, l_run = 0.0f;
  bf16x8 qf[4];
  // rel_w values kept in registers: the whole row, or (MULTI) the 32 columns this lane's accumulator elements ever meet
  // -- element e of an even / odd 32-key tile is column (tile & 1) * 32 + (e & 3) + 8 (e >> 2) + 4 hh, so a lane needs
  // 2 x 16 of the 64 and the upper lane half simply loads its own, shifted, set: no per-element select between halves
  constexpr int NRW = MULTI ? 32 : S;
  constexpr int NRH = MULTI ? 2 : S;     // rel_h values: the whole column, or the chunk's two grid rows
  constexpr int NPAD = 2 * ((2 * S - 1 + 31) / 32 * 32);  // relpos_npad(S)
  float relw[MULTI ? 1 : NRW], relh[NRH];
  f32x16 rw8[2];  // MULTI: 8 x rel_w of the lane's 16 columns in an even / odd 32-key tile -- the QK^T accumulators START from these
  const float* relrow = nullptr;

  auto load_q = [&](int qt) {
    int q = qt * 32 + l31;
    q = q < p.seq ? q : p.seq - 1;
    const bf16_t* src = qb + (int64_t)q * p.ld_qkv + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
    const int qh = q / S, qw = q - qh * S;
    const float* trow = p.T + (((int64_t)b * p.seq + q) * p.heads + hd) * NPAD;
    relrow = trow + qh + (S - 1);  // rel_h[kh] = relrow[-kh]
    const float* wrow = trow + NPAD / 2 + qw + (S - 1);
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      if constexpr (MULTI) rw8[j >> 4][j & 15] = 8.0f * wrow[-((j >> 4) * 32 + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh)];  // (x 8: see process)
      else relw[j] = wrow[-j];
    }
    if (!MULTI) {
#pragma unroll
      for (int j = 0; j < NRH; ++j) relh[j] = relrow[-j];
    }
#pragma unroll
    for (int nd = 0; nd < 2; ++nd)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[nd][e] = 0.0f;
    m_run = -INFINITY;
    l_run = 0.0f;
  };

  // MULTI: the lane part of a piece's source address does not depend on the chunk (seq is a whole number of chunks:
  // no clamp); kept as 32-bit offsets beside a wave-uniform base that moves by one chunk
  uint32_t koff[MULTI ? PPW : 1], voff[MULTI ? PPW : 1];
  if (MULTI) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int r = (wave * PPW + q) * 8 + (lane >> 3);
      koff[q] = (uint32_t)r * (uint32_t)(p.ld_qkv * 2) + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
      voff[q] = (uint32_t)r * (uint32_t)(p.ld_qkv * 2) + (uint32_t)(((lane & 7) ^ (((r >> 1) & 1) << 2)) * 16);
    }
  }
  auto stage_issue = [&](int kc0, char* buf) {
    char* sK = buf;
    char* sVt = buf + KEYS * 128;
    if (MULTI) {
      const bf16_t* kc = kb + (int64_t)kc0 * p.ld_qkv;
#pragma unroll
      for (int q = 0; q < PPW; ++q) glds16_raw(kc, koff[q], sK + (wave * PPW + q) * 1024);
#pragma unroll
      for (int q = 0; q < PPW; ++q) glds16_raw(kc + HD, voff[q], sVt + (wave * PPW + q) * 1024);
      return;
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int piece = wave * NT + q;
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int key = kc0 + r;
      key = key < p.seq ? key : p.seq - 1;
      glds16_raw(kb + (int64_t)key * p.ld_qkv + c * 8, sK + piece * 1024);
    }
    // V rows the same way (row-major, chunk ^ (((key >> 1) & 1) << 2)): consumed by ds_read_b64_tr_b16 below, so no
    // register-staged transpose; rows past the sequence repeat the last key (their P is exp(-inf) = 0)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int piece = wave * NT + q;
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (((r >> 1) & 1) << 2);
      int key = kc0 + r;
      key = key < p.seq ? key : p.seq - 1;
      glds16_raw(vb + (int64_t)key * p.ld_qkv + c * 8, sVt + piece * 1024);
    }
  };
  auto stage_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  auto process = [&](int kc0, bool rescale, const char* buf) {
    const char* sK = buf;
    const __attribute__((address_space(3))) char* sVtr = (const __attribute__((address_space(3))) char*)buf + vbase;
    f32x16 s[NT];
    __builtin_amdgcn_s_setprio(0);  // (low for the K.Q^T MFMAs, high for the vector-heavy rest: attention.hip, attn_persist_kernel)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // MULTI: the first MFMA takes 8 rel_w[kw] as its C operand (registers that never change: no per-element add, no
      // copy) and the rest add q.k on top; 8 rel_h[kh] is constant over a tile (grid row t >> 1 of the chunk) and enters
      // the row maximum and the exp2 offset below.  logits = (acc + 8 rel_h) / 8, the 1/8 folded into the exp2 scale.
#pragma unroll
      for (int e = 0; e < 16; ++e) s[t][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf =
            *reinterpret_cast<const bf16x8*>(sK + (t * 32 + l31) * 128 + (((2 * ks + hh) ^ swz) * 16));
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], (MULTI && ks == 0) ? rw8[t & 1] : s[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(2);
    // windows: logits = s * dh^-0.5 + rel_h[kh] + rel_w[kw]; the key of element (t, e, half) is static
    if constexpr (!MULTI) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k0 = t * 32 + (e & 3) + 8 * (e >> 2);  // half 0; half 1 is k0 + 4
          const int k1 = k0 + 4;
          const int kh0 = k0 / S < S ? k0 / S : S - 1;
          const int kh1 = k1 / S < S ? k1 / S : S - 1;
          const float b0 = relh[kh0] + relw[k0 % S];
          const float b1 = relh[kh1] + relw[k1 % S];
          const float bias = hh ? b1 : b0;
          s[t][e] = fmaf(s[t][e], 0.125f, bias);
        }
        if (kc0 + t * 32 + 32 > p.seq) mask_keys(s[t], kc0 + t * 32, hh, p.seq);
      }
    }
    constexpr float SC = MULTI ? 0.125f * LOG2E : LOG2E;  // exp2 scale of the values held in s[]
    float mx = -INFINITY;
    if constexpr (MULTI) {
      float mrow[2] = {-INFINITY, -INFINITY};
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) mrow[t >> 1] = fmaxf(mrow[t >> 1], s[t][e]);
      mx = fmaxf(mrow[0] + relh[0], mrow[1] + relh[1]);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[t][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (rescale) {
      const float alpha = fast_exp2((m_run - m_new) * SC);
      l_run *= alpha;
#pragma unroll
      for (int nd = 0; nd < 2; ++nd)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[nd][e] *= alpha;
    }
    m_run = m_new;
    const float mb = m_new * SC;
    float crow[2];  // exp2 offset of a tile: -max (+ 8 rel_h of its grid row), scaled
    crow[0] = MULTI ? fmaf(relh[0], SC, -mb) : -mb;
    crow[1] = MULTI ? fmaf(relh[1], SC, -mb) : -mb;
    // (The row sum through the matrix pipe instead -- one more MFMA per 16-key slice with an all-ones A operand, 8 issue
    // cycles against 32 for the 8 adds it replaces -- measured the same within noise: 7.16 vs 7.34 ms per MedSAM step.)
    f32x2 lsum2 = {0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
        softmax_slice8(s[t], s2, SC, crow[MULTI ? t >> 1 : 0], lsum2, pf);  // packed pairs: vdr_dev.h
#pragma unroll
        for (int nd = 0; nd < 2; ++nd) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sVtr + (t * 32 + s2 * 16) * 128 + vch[nd]));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sVtr + (t * 32 + s2 * 16 + 8) * 128 + vch[nd]));
          bf16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = lo[j];
            vf[4 + j] = hi[j];
          }
          o[nd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[nd], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l_run += lsum2[0] + lsum2[1];
  };

  auto store = [&](int qt) {
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    const int q = qt * 32 + l31;
    store_row64_bf16(ob + (int64_t)(q < p.seq ? q : 0) * p.ld_out, o, inv, hh, q < p.seq);
  };

  if (!MULTI) {
    stage_issue(0, smem);
    stage_wait();
    for (int qt = qt_begin + wave; qt < qt_end; qt += 4) {
      load_q(qt);
      process(0, false, smem);
      store(qt);
    }
  } else {
    const int qt = qt_begin + wave;
    const bool valid = qt < qt_end;
    load_q(valid ? qt : qt_begin);
    // double-buffered: chunk c+1 streams into the other buffer while chunk c is processed; one barrier per chunk.
    // The LDS-DMA is issued as an opaque instruction (glds16_raw) so that hipcc orders nothing after it; what it
    // still does is meet the first use of every ordinary load with a `vmcnt(n)` counted over the loads IT knows,
    // which -- vmcnt retires in issue order -- would wait for the DMA pieces issued before them as well.  So every
    // register filled by a load is "used" (empty asm) right after the explicit vmcnt(0) of stage_wait and nowhere
    // else first: q / rel_w after the first one, the two rel_h values of a chunk (grid rows kc0/64, kc0/64 + 1)
    // one chunk ahead, after the wait that ends the chunk before.  (With the builtin DMA and the rel_h loads inside process() every chunk sat out the whole
    // latency of the prefetch it had just issued.)
    float rh_next[2];
    auto load_relh = [&](int kc0) {
      rh_next[0] = relrow[-(kc0 >> 6)];  // (scaled by 8 where they are retired: arithmetic on them here would be a use)
      rh_next[1] = relrow[-min((kc0 >> 6) + 1, S - 1)];
    };
    load_relh(0);
    stage_issue(0, smem);
    stage_wait();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(rw8[j]));
    for (int c = 0; c < p.n_chunks; ++c) {
      asm volatile("" : "+v"(rh_next[0]), "+v"(rh_next[1]));
      relh[0] = 8.0f * rh_next[0];
      relh[1] = 8.0f * rh_next[1];
      const bool more = c + 1 < p.n_chunks;
      if (more) {
        load_relh((c + 1) * KEYS);
        stage_issue((c + 1) * KEYS, smem + ((c + 1) & 1) * BUF);
      }
      if (valid) process(c * KEYS, c > 0, smem + (c & 1) * BUF);
      if (more) stage_wait();
    }
    if (valid) store(qt);
  }
}

template <int NT, int S, bool MULTI>
static hipError_t launch_rp(const AttnRK& k, int batch, hipStream_t s) {
  constexpr size_t lds = (MULTI ? 2 : 1) * 2 * (size_t)NT * 32 * 128;  // (K image + V image) x 1 or 2 buffers
  auto fn = attn_relpos_kernel<NT, S, MULTI>;
  static PerDeviceFlag attr;  // per instantiation and device: raised once, not per launch
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  if (lds > 65536 && !attr.done[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr.done[dev] = true;
  }
  const int nqt = (k.seq + 31) / 32;
  const dim3 grid((unsigned)(batch * k.heads * ((nqt + k.qt_per_block - 1) / k.qt_per_block)));
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, k);
  return hipGetLastError();
}

hipError_t launch_attention_relpos(const void* qkv, const float* T, void* out, int batch, int S, int heads,
                                   hipStream_t s) {
  if (batch <= 0 || S <= 0 || heads <= 0) return hipErrorInvalidValue;
  AttnRK k;
  k.qkv = (const bf16_t*)qkv;
  k.T = T;
  k.out = (bf16_t*)out;
  k.seq = S * S;
  k.heads = heads;
  k.ld_qkv = (int64_t)3 * heads * 64;
  k.ld_out = (int64_t)heads * 64;
  const int nqt = (k.seq + 31) / 32;
  k.qt_per_block = nqt;
  k.n_chunks = 1;
  switch (S) {
    case 4:
      return launch_rp<2, 4, false>(k, batch, s);
    case 7:
      return launch_rp<2, 7, false>(k, batch, s);
    case 10:
      return launch_rp<4, 10, false>(k, batch, s);
    case 14:
      return launch_rp<7, 14, false>(k, batch, s);
    case 64:
      k.qt_per_block = 4;
      k.n_chunks = (k.seq + 127) / 128;
      return launch_rp<4, 64, true>(k, batch, s);
    default:
      return hipErrorInvalidValue;  // grid sides outside {4, 7, 10, 14, 64} are not instantiated
  }
}

}  // namespace vdr
