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

template <int NT, int S, bool MULTI>
__global__ __launch_bounds__(256, 2) void attn_relpos_kernel(AttnRK p) {
  constexpr int KEYS = NT * 32;
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

  const int b = blockIdx.x / p.heads;
  const int hd = blockIdx.x - b * p.heads;
  const int HD = p.heads * 64;
  const bf16_t* qb = p.qkv + (int64_t)b * p.seq * p.ld_qkv + hd * 64;
  const bf16_t* kb = qb + HD;
  const bf16_t* vb = qb + 2 * HD;
  bf16_t* ob = p.out + (int64_t)b * p.seq * p.ld_out + hd * 64;

  const int nqt = (p.seq + 31) >> 5;
  const int qt_begin = blockIdx.y * p.qt_per_block;
  const int qt_end = min(nqt, qt_begin + p.qt_per_block);
  constexpr float LOG2E = 1.44269504088896341f;

  f32x16 o[2];
  float m_run = -INFINITY, l_run = 0.0f;
  bf16x8 qf[4];
  constexpr int NRW = S;                 // rel_w values kept in registers
  constexpr int NRH = MULTI ? 2 : S;     // rel_h values: the whole column, or the chunk's two grid rows
  constexpr int NPAD = 2 * ((2 * S - 1 + 31) / 32 * 32);  // relpos_npad(S)
  float relw[NRW], relh[NRH];
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
    for (int j = 0; j < NRW; ++j) relw[j] = wrow[-j];
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

  auto stage_issue = [&](int kc0, char* buf) {
    char* sK = buf;
    char* sVt = buf + KEYS * 128;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int piece = wave * NT + q;
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int key = kc0 + r;
      key = key < p.seq ? key : p.seq - 1;
      glds16(kb + (int64_t)key * p.ld_qkv + c * 8, sK + piece * 1024);
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
      glds16(vb + (int64_t)key * p.ld_qkv + c * 8, sVt + piece * 1024);
    }
  };
  auto stage_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  auto process = [&](int kc0, bool rescale, const char* buf) {
    const char* sK = buf;
    const __attribute__((address_space(3))) char* sVtr = (const __attribute__((address_space(3))) char*)buf + vbase;
    if (MULTI) {
      // this chunk covers grid rows kc0/64 and kc0/64 + 1
      relh[0] = relrow[-(kc0 >> 6)];
      relh[1] = relrow[-min((kc0 >> 6) + 1, S - 1)];
    }
    f32x16 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[t][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf =
            *reinterpret_cast<const bf16x8*>(sK + (t * 32 + l31) * 128 + (((2 * ks + hh) ^ swz) * 16));
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
      }
    }
    // logits = s * dh^-0.5 + rel_h[kh] + rel_w[kw]; the (chunk-local) key of element (t, e, half) is static
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        constexpr int dummy = 0;
        (void)dummy;
        const int k0 = t * 32 + (e & 3) + 8 * (e >> 2);  // half 0; half 1 is k0 + 4
        const int k1 = k0 + 4;
        const int kh0 = MULTI ? (k0 >> 6) : (k0 / S < S ? k0 / S : S - 1);
        const int kh1 = MULTI ? (k1 >> 6) : (k1 / S < S ? k1 / S : S - 1);
        const int kw0 = MULTI ? (k0 & 63) : k0 % S;
        const int kw1 = MULTI ? (k1 & 63) : k1 % S;
        const float b0 = relh[kh0] + relw[kw0];
        const float b1 = relh[kh1] + relw[kw1];
        const float bias = hh ? b1 : b0;
        s[t][e] = fmaf(s[t][e], 0.125f, bias);
      }
      if (kc0 + t * 32 + 32 > p.seq) mask_keys(s[t], kc0 + t * 32, hh, p.seq);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[t][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (rescale) {
      const float alpha = fast_exp2((m_run - m_new) * LOG2E);
      l_run *= alpha;
#pragma unroll
      for (int nd = 0; nd < 2; ++nd)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[nd][e] *= alpha;
    }
    m_run = m_new;
    const float mb = m_new * LOG2E;
    float lsum = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = fast_exp2(fmaf(s[t][8 * s2 + j], LOG2E, -mb));
          lsum += pv;
          pf[j] = (bf16_t)pv;
        }
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
    l_run += lsum;
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
    // double-buffered: chunk c+1 streams into the other buffer while chunk c is processed; one barrier per chunk
    stage_issue(0, smem);
    for (int c = 0; c < p.n_chunks; ++c) {
      stage_wait();
      if (c + 1 < p.n_chunks) stage_issue((c + 1) * KEYS, smem + ((c + 1) & 1) * BUF);
      if (valid) process(c * KEYS, c > 0, smem + (c & 1) * BUF);
    }
    if (valid) store(qt);
  }
}

template <int NT, int S, bool MULTI>
static hipError_t launch_rp(const AttnRK& k, int batch, hipStream_t s) {
  constexpr size_t lds = (MULTI ? 2 : 1) * 2 * (size_t)NT * 32 * 128;  // (K image + V image) x 1 or 2 buffers
  auto fn = attn_relpos_kernel<NT, S, MULTI>;
  static bool attr_set = false;  // per instantiation: raised once, not per launch
  if (lds > 65536 && !attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nqt = (k.seq + 31) / 32;
  const dim3 grid((unsigned)(batch * k.heads), (unsigned)((nqt + k.qt_per_block - 1) / k.qt_per_block));
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
