// Fused multi-head self-attention for gfx950, head dim 64:  out = softmax(q k^T / 8) v per (image, head)
//
// Replaces F.scaled_dot_product_attention inside nn.MultiheadAttention (reference
// src/models_archs.py:130-135) and Attention.forward of the frozen ViTs called at
// src/tfds_dense_descriptor.py:123.  No mask, no dropout (eval).
//
// One workgroup (4 waves) per (image, head[, q-block]).  K of the current key chunk is staged in
// LDS by global_load_lds as [key][64] bf16 (128-B rows, 16-B chunks XOR-swizzled by (key>>1)&7);
// V is staged TRANSPOSED as Vt[d][key] (row stride NT*64+8 bytes, odd multiple of 8 -> the 8-byte
// fragment reads are bank-conflict free).  Each wave owns 32-query tiles and computes
//     S^T = K . Q^T   (MFMA 32x32x16, key on the accumulator row, query on the lane)
// so a whole softmax row lives in one lane pair (lane, lane^32): row max / sum are register
// reductions plus one cross-half shuffle.  exp2((s - max) * 0.125*log2 e) in fp32, P rounded to
// bf16 IN the accumulator registers and fed straight back as the B operand of
//     O^T = V^T . P^T   (the accumulator-as-operand form; the V fragment is read in the same
//                        permuted key order the accumulator registers hold)
// so P never touches LDS.  seq <= 288 runs as one chunk (no rescale); longer sequences run an
// online softmax over 128-key chunks with one query tile per wave.
#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

struct AttnK {
  const bf16_t* qkv;
  bf16_t* out;
  int seq, heads;
  int64_t ld_qkv, ld_out;
  int qt_per_block;  // query tiles handled by one workgroup
  int n_chunks;      // key chunks of NT*32 keys
  int abl;           // tuning builds (-DVDR_TUNING): 1 = no compute, 2 = no staging after the first item
  const int* lens;   // non-null: per-batch-entry valid length = lens[b] + len_add (<= seq); keys past it are masked
  int len_add;       //           (variable-length token sequences padded to seq)
  uint8_t* o_scale;  // non-null: `out` is an MX-fp8 payload [tokens][heads*64] and this its e8m0 scale array
  int64_t os_rows;   //           (mx.hip layout: [heads*2 blocks][os_rows], rows paired inside 64-row groups)
#ifdef VDR_ATTN_STAMPS
  unsigned long long* stamps;  // tools/micro/attn_stamps.hip: [workgroup][8 waves][16 items][16] cycle stamps of the phases
#endif
};

#ifdef VDR_ATTN_STAMPS
#define VDR_STAMP(i) st[i] = __builtin_readcyclecounter()
#else
#define VDR_STAMP(i)
#endif

// One output row (token `grow`, head `hd`): this lane holds dims nd*32 + 8g + 4hh + e of it, its partner lane
// (xor 32) the other half of each 32-dim block.  bf16 store, or -- fp8 path, operand of the MX out-projection --
// one e8m0 scale per 32-dim block agreed with the partner lane and 4-byte e4m3 stores.
VDR_DEV void attn_store_row(const AttnK& p, const f32x16 (&o)[2], float inv, bool ok, int64_t grow, int hd, int hh) {
  if (p.o_scale) {
    uint8_t* dst = reinterpret_cast<uint8_t*>(p.out) + grow * p.ld_out + hd * 64;
#pragma unroll
    for (int nd = 0; nd < 2; ++nd) {
      float amax = 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(o[nd][e] * inv));
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const float t = amax * (1.0f / 448.0f);
      const uint32_t tb = __float_as_uint(t);
      int ex = (int)((tb >> 23) & 255) - 127 + ((tb & 0x7fffff) ? 1 : 0);
      ex = ex < -126 ? -126 : (ex > 126 ? 126 : ex);
      const float qs = inv * __uint_as_float((uint32_t)(127 - ex) << 23);
      if (ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(o[nd][4 * g] * qs, o[nd][4 * g + 1] * qs, w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(o[nd][4 * g + 2] * qs, o[nd][4 * g + 3] * qs, w, true);
          *reinterpret_cast<int*>(dst + nd * 32 + 8 * g + 4 * hh) = w;
        }
        if (hh == nd)
          p.o_scale[(int64_t)(hd * 2 + nd) * p.os_rows + (grow & ~(int64_t)63) + 2 * (grow & 31) + ((grow >> 5) & 1)] =
              (uint8_t)(ex + 127);
      }
    }
    return;
  }
  store_row64_bf16(p.out + grow * p.ld_out + hd * 64, o, inv, hh, ok);
}

template <int NT>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnK p) {
  constexpr int KEYS = NT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                 // KEYS * 128 B
  char* sVt = smem + KEYS * 128;   // V image, KEYS * 128 B, row-major like K

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int l31 = lane & 31;
  const int swz = (lane >> 1) & 7;
  // transposed V read: lane 4q+p of a 16-lane group addresses key row q, d columns 4p..4p+3 of a 4 x 16 block and
  // receives d column (lane & 15) of the 4 keys (groups: d half (lane >> 4) & 1, key offset 4 hh)
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const int tq = (lane & 15) >> 2, tp = lane & 3, dg = (lane >> 4) & 1;
  const int vkey = 4 * hh + tq;
  const __attribute__((address_space(3))) char* sVtr =
      (const __attribute__((address_space(3))) char*)sVt + vkey * 128 + 8 * (tp & 1);
  int vch[2];
#pragma unroll
  for (int nd = 0; nd < 2; ++nd) vch[nd] = ((4 * nd + 2 * dg + (tp >> 1)) ^ (((vkey >> 1) & 1) << 2)) * 16;

  // 1-D grid of (image, head) x query blocks, query block fastest, walked in XCD-contiguous order: the query blocks of
  // one (image, head) run next to each other on ONE XCD and find its K / V in that L2.  (As a 2-D grid with the query
  // block on y they were a whole grid row apart: every block re-read K / V from HBM -- ViT-L/14@336, 5 query blocks
  // per head: 0.83 GB per launch at 5.2 TB/s.)
  const int nqt = (p.seq + 31) >> 5;
  const int nyb = (nqt + p.qt_per_block - 1) / p.qt_per_block;
  const int vid = nyb > 1 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bh = vid / nyb;
  const int yb = vid - bh * nyb;
  const int b = bh / p.heads;
  const int hd = bh - b * p.heads;
  const int HD = p.heads * 64;
  const bf16_t* qb = p.qkv + (int64_t)b * p.seq * p.ld_qkv + hd * 64;
  const int len = p.lens ? min(p.seq, p.lens[b] + p.len_add) : p.seq;  // valid keys of this sequence
  const bf16_t* kb = qb + HD;
  const bf16_t* vb = qb + 2 * HD;

  const int qt_begin = yb * p.qt_per_block;
  const int qt_end = min(nqt, qt_begin + p.qt_per_block);
  const float sc = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e)

  // per-wave running state (multi-chunk mode: exactly one query tile per wave)
  f32x16 o[2];
  float m_run = -INFINITY, l_run = 0.0f;
  bf16x8 qf[4];

  auto load_q = [&](int qt) {
    int q = qt * 32 + l31;
    q = q < p.seq ? q : p.seq - 1;
    const bf16_t* src = qb + (int64_t)q * p.ld_qkv + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
#pragma unroll
    for (int nd = 0; nd < 2; ++nd)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[nd][e] = 0.0f;
    m_run = -INFINITY;
    l_run = 0.0f;
  };

  // staging of one key chunk.  The LDS-DMA is an opaque instruction (glds16_raw): hipcc orders nothing after it, the
  // explicit vmcnt(0) + barrier of stage_wait does.
  auto stage_issue = [&](int kc0) {
    // K: KEYS rows of 128 B, 8 rows per wave-instruction
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int piece = wave * NT + q;  // 0 .. 4*NT-1, rows piece*8 .. +7
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int key = kc0 + r;
      key = key < p.seq ? key : p.seq - 1;
      glds16_raw(kb + (int64_t)key * p.ld_qkv + c * 8, sK + piece * 1024);
    }
    // V rows the same way (row-major, chunk ^ (((key >> 1) & 1) << 2)), consumed by ds_read_b64_tr_b16 in process():
    // no register-staged transpose; rows past the sequence repeat the last key (their P is exp(-inf) = 0)
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

  auto process = [&](int kc0, bool rescale) {
    f32x16 s[NT];
    __builtin_amdgcn_s_setprio(0);  // (low for the K.Q^T MFMAs, high for the vector-heavy rest: see attn_persist_kernel)
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
    __builtin_amdgcn_s_setprio(2);
    // mask keys >= seq (only tiles that straddle or lie beyond the end)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (kc0 + t * 32 + 32 > len) mask_keys(s[t], kc0 + t * 32, hh, len);
    }
    float mx = row_max_tiles<NT>(s);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (rescale) {
      const float alpha = fast_exp2((m_run - m_new) * sc);
      l_run *= alpha;
#pragma unroll
      for (int nd = 0; nd < 2; ++nd)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[nd][e] *= alpha;
    }
    m_run = m_new;
    const float mb = m_new * sc;
    f32x2 lsum2 = {0.0f, 0.0f};  // fp32 row sum, (even, odd) elements (same order in the persistent kernel: bitwise equal)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (kc0 + t * 32 + s2 * 16 >= len) continue;  // fully masked slice (wave-uniform): P = 0, nothing to add
        bf16x8 pf;
        softmax_slice8(s[t], s2, sc, -mb, lsum2, pf);
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
        __builtin_amdgcn_sched_barrier(0);  // keep the exp/cvt of later key slices from being hoisted (VGPR cap)
      }
    }
    l_run += lsum2[0] + lsum2[1];
  };

  auto store = [&](int qt) {
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    const int q = qt * 32 + l31;
    attn_store_row(p, o, inv, q < p.seq, (int64_t)b * p.seq + (q < p.seq ? q : 0), hd, hh);
  };

  if (p.n_chunks == 1) {
    stage_issue(0);
    stage_wait();
    for (int qt = qt_begin + wave; qt < qt_end; qt += 4) {
      load_q(qt);
      process(0, false);
      store(qt);
    }
  } else {
    // One image pair, restaged per chunk: at 164 VGPRs and 32 KB of LDS three workgroups share a CU and compute while
    // this one waits.  (Ping-pong images with the next chunk's DMA in flight under process() take 170 VGPRs and 64 KB --
    // two workgroups per CU -- and measured slower: ViT-L/14@336 attention 3.46 -> 3.74 ms per step.)
    const int qt = qt_begin + wave;
    const bool valid = qt < qt_end;  // wave-uniform
    load_q(valid ? qt : qt_begin);
    for (int c = 0; c < p.n_chunks; ++c) {
      if (c) __syncthreads();  // every wave is done reading the previous chunk
      stage_issue(c * KEYS);
      stage_wait();
      if (valid) process(c * KEYS, c > 0);
    }
    if (valid) store(qt);
  }
}

// -------------------------------------------------------------------------------------------------
// Persistent variant for single-chunk sequences (seq <= NT*32, at most 7 query tiles):
//   one 448-thread workgroup per CU walks the (image, head) items; wave w owns query tile w of the
//   CURRENT item (7 tiles at seq 197: one per wave, no tail).  K / V^T live in two LDS buffers: at the
//   top of an item every wave issues its share of the NEXT item's staging loads (K by
//   global_load_lds straight into the other buffer, V rows into registers), computes its tile while
//   they are in flight, then packs and writes its V^T share.  One s_barrier per item; the staging
//   latency that the one-shot kernel exposes at the head of every workgroup (52 % of its wave-cycles
//   were waits) sits under the MFMAs.
// -------------------------------------------------------------------------------------------------
// LOADER: an eighth wave (512 threads) issues ALL of the next item's DMA pieces and nothing else; the seven computing
// waves issue none.  An LDS-DMA instruction costs its wave 100-400 cycles of issue in a busy phase (stamped: 8 pieces
// per wave and item took 3 k of a 20 k-cycle item in the two-pass kernel below), and the loader sits on the SIMD that
// holds only one computing wave (7 tiles on 4 SIMDs: 2, 2, 2, 1).
template <int NT, bool LOADER>
__global__ __launch_bounds__(LOADER ? 512 : 448, 2) void attn_persist_kernel(AttnK p, int n_items) {
  constexpr int KEYS = NT * 32;
  constexpr int BUF = 2 * KEYS * 128;                 // K image + V image, both [key][64 d] rows of 128 B
  constexpr int NCW = 7;                              // waves == query tiles served per item
  constexpr int PPW = (NT * 8 + NCW - 1) / NCW;       // 8-row DMA pieces (K: NT*4, then V: NT*4) per wave (no loader wave)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int l31 = lane & 31;
  const int swz = (lane >> 1) & 7;
  const int HD = p.heads * 64;
  const int nqt = (p.seq + 31) >> 5;
  const float sc = 0.125f * 1.44269504088896341f;
#ifdef VDR_ATTN_STAMPS
  unsigned long long st[16] = {};  // tools/micro/attn_stamps.hip: s_memtime at the phase boundaries of the current item
#endif

  // ---- staging of one item: K and V rows straight into LDS by global_load_lds, no registers -------
  //   K: 16-B chunk ^ ((key >> 1) & 7)            (ds_read_b128 row reads of the 32x32x16 operand)
  //   V: 16-B chunk ^ (((key >> 1) & 1) << 2)     (ds_read_b64_tr_b16 blocks of 4 keys x 16 d: the two even / odd keys
  //                                                of a block land in different halves of their 32 banks)
  // V stays row-major: the transposed read hands every lane V[key0 .. key0+3][d] -- the P.V operand -- so the
  // register-staged transpose (32 VGPRs and 16 ds_write_b32 per thread and item) of the first version is gone.
  auto stage_issue = [&](int item, char* buf) {
    const int b = item / p.heads;
    const int hd = item - b * p.heads;
    const bf16_t* kb = p.qkv + (int64_t)b * p.seq * p.ld_qkv + hd * 64 + HD;
    if (LOADER) {
      // the loader wave: all NT*8 pieces, a rolled loop with the lane part of the address rebuilt per piece (kept
      // unrolled, hipcc hoists 56 64-bit addresses out of the item loop and spills)
      if (wave != NCW) return;
      __builtin_amdgcn_s_setprio(3);  // (the loader's 56 DMA instructions go ahead of the computing waves' work)
      const uint32_t ldb = (uint32_t)p.ld_qkv * 2;
      const int r0 = lane >> 3;
      const uint32_t ck = (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);               // K chunk, ^ 64 on odd pieces
      const uint32_t cv = (uint32_t)(((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 16);  // V chunk
#pragma unroll 2
      for (int i = 0; i < NT * 4; ++i) {
        const int r = i * 8 + r0;
        const uint32_t row = (uint32_t)(r < p.seq ? r : p.seq - 1) * ldb;
        glds16_raw(kb, row + (ck ^ (uint32_t)((i & 1) << 6)), buf + i * 1024);
        glds16_raw(kb + HD, row + cv, buf + (NT * 4 + i) * 1024);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + i * NCW;
      if (piece < NT * 8) {
        const int isv = piece >= NT * 4;
        const int r = (piece - isv * NT * 4) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (isv ? (((r >> 1) & 1) << 2) : ((r >> 1) & 7));
        const int key = r < p.seq ? r : p.seq - 1;
        glds16_raw(kb + isv * HD + (int64_t)key * p.ld_qkv + c * 8, buf + piece * 1024);
      }
    }
  };
  // end of an item: this wave's K / V pieces of the NEXT item have landed and its LDS reads of the current one are
  // done.  A computing wave already waited for its pieces (and the next Q) right before its output stores -- they had
  // the whole tile's compute time to land -- so that the stores themselves stay in flight across the barrier instead
  // of every wave sitting out their acknowledgement once per item.
  auto stage_write = [&](bool waited_before_stores) {
    if (waited_before_stores) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };

  // Q fragments of this wave's query tile (B operand of S^T = K.Q^T): 4 x 16 B straight from global
  auto load_q = [&](int item, int qt, bf16x8 (&q)[4]) {
    const int b = item / p.heads;
    const int hd = item - b * p.heads;
    int qr = qt * 32 + l31;
    qr = qr < p.seq ? qr : p.seq - 1;
    const bf16_t* src = p.qkv + ((int64_t)b * p.seq + qr) * p.ld_qkv + hd * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
  };

  // compute: one 32-query tile of an item against the staged K / V^T.  Fragment reads run one step
  // ahead of the MFMAs that consume them, and the exp/convert work of key slice i+1 is issued right
  // after the MFMAs of slice i so the VALU and the matrix pipe overlap.
  // Keys past the sequence (197 tokens: 27 of the last tile's 32) are masked by the MFMA itself: the LAST key tile's
  // score accumulator starts from this vector (0 for a valid key, -inf past the end) instead of from zero, so the
  // scores of the padding keys come out as -inf with no instruction spent on them.  As a per-item select on the 16
  // scores (mask_keys) it was ~85 instructions of every item, 7 % of the loop body.  The sequence length is the same
  // for every item of a launch, so the vector lives in 16 registers for the whole kernel.
  // (A launch whose sequence ends before the last tile -- NT is fixed per instantiation, 130 tokens run with NT = 7 --
  // keeps the select: the vector is all zeros then.)
  const bool last_tile_only = p.seq > (NT - 1) * 32;  // wave-uniform
  f32x16 kmask;
  {
    const int thr = last_tile_only ? p.seq - (NT - 1) * 32 - 4 * hh : 64;
#pragma unroll
    for (int e = 0; e < 16; ++e) kmask[e] = ((e & 3) + 8 * (e >> 2) >= thr) ? -INFINITY : 0.0f;
    asm volatile("" : "+v"(kmask));  // (opaque: kept, not recomputed per item)
  }
  auto compute_tile = [&](int item, int next_item, int qt, const char* buf, bf16x8 (&qf)[4]) {
    const int b = item / p.heads;
    const int hd = item - b * p.heads;
    const char* sK = buf + l31 * 128;
    // transposed V read: lane 4q+p of its 16-lane group addresses key row q, d columns 4p..4p+3 of a 4 x 16 block and
    // receives d column (lane & 15) of the 4 keys; groups: d half (lane >> 4) & 1, key offset 4 hh
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int tq = (lane & 15) >> 2, tp = lane & 3, dg = (lane >> 4) & 1;
    const int vkey = 4 * hh + tq;  // + 16 it (+ 8 for the high half): multiples of 8 keep (key >> 1) & 1
    const __attribute__((address_space(3))) char* sV =
        (const __attribute__((address_space(3))) char*)(buf + KEYS * 128) + vkey * 128 + 8 * (tp & 1);
    int vch[2];
#pragma unroll
    for (int nd = 0; nd < 2; ++nd) vch[nd] = ((4 * nd + 2 * dg + (tp >> 1)) ^ (((vkey >> 1) & 1) << 2)) * 16;
    int kch[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kch[ks] = ((2 * ks + hh) ^ swz) * 16;

    f32x16 s[NT];
    // K fragments "one MFMA ahead" in the source; hipcc folds the two registers into one and emits ds_read / wait /
    // v_mfma back to back.  A hand-pipelined form (inline-asm ds_read_b128 two MFMAs ahead, counted lgkmcnt: exactly the
    // intended ISA, 215 VGPRs) measured the same within the A/B harness's noise (tools/ab_libs.py: 81.2 vs 79.3 us):
    // with two waves per SIMD the partner's instructions already fill those waits, so the plain form stays.
    bf16x8 kf[2];
    // Wave priority follows the phase: low while this wave streams the 28 MFMAs of K.Q^T, high for the softmax / P.V
    // phase that is mostly vector work.  A wave that has an MFMA ready whenever the pipe frees up otherwise holds the
    // SIMD's issue slot and the other wave's vector instructions wait behind it (tools/micro/coissue.hip, modes 2 / 7:
    // a vector wave next to an MFMA wave runs at 440 cycles per round without and 224 with the higher priority, the MFMA
    // wave at its full 256 either way).
    __builtin_amdgcn_s_setprio(0);
    kf[0] = *reinterpret_cast<const bf16x8*>(sK + kch[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t == NT - 1) {
        s[t] = kmask;  // (NT = ceil(seq / 32): only the last tile can hold keys past the end)
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[t][e] = 0.0f;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int i = t * 4 + ks;
        if (i + 1 < NT * 4)
          kf[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sK + ((i + 1) >> 2) * 32 * 128 + kch[(i + 1) & 3]);
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i & 1], qf[ks], s[t], 0, 0, 0);
      }
    }
#ifdef VDR_ATTN_STAMPS
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(s[t]));
#endif
    VDR_STAMP(2);
    __builtin_amdgcn_s_setprio(2);
    // qf is dead from here: fetch the next item's Q into it; the loads land under the softmax / P.V
    if (next_item >= 0) load_q(next_item, qt, qf);
    if (!last_tile_only) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t * 32 + 32 > p.seq) mask_keys(s[t], t * 32, hh, p.seq);
    }
    float mx = row_max_tiles<NT>(s);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mb = mx * sc;
#ifdef VDR_ATTN_STAMPS
    asm volatile("" : "+v"(mx));
#endif
    VDR_STAMP(3);
    f32x2 lsum2 = {0.0f, 0.0f};
    f32x16 o[2];
#pragma unroll
    for (int nd = 0; nd < 2; ++nd)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[nd][e] = 0.0f;

    constexpr int NI = NT * 2;  // 16-key slices
    // slices that lie entirely past the sequence (197 tokens: the 14th of 14) carry P = exp(-inf) = 0: neither their
    // exponentials nor their two MFMAs are issued (adds of +0 left out: results unchanged)
    const int ni_valid = __builtin_amdgcn_readfirstlane((p.seq + 15) >> 4);
    bf16x4 vlo[2], vhi[2];      // V^T fragments of the slice in flight (both d halves)
    bf16x8 pf[2];
    auto read_v = [&](int it) {
#pragma unroll
      for (int nd = 0; nd < 2; ++nd) {
        vlo[nd] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + it * 16 * 128 + vch[nd]));
        vhi[nd] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + it * 16 * 128 + 8 * 128 + vch[nd]));
      }
    };
    auto make_p = [&](int it, int set) {
      const int t = it >> 1, s2 = it & 1;
      softmax_slice8(s[t], s2, sc, -mb, lsum2, pf[set]);
    };
    read_v(0);
    make_p(0, 0);
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int cur = it & 1;
      if (it >= NI - 2 && it >= ni_valid) break;  // only the last tile can hold a fully masked slice (nqt tiles cover seq)
#pragma unroll
      for (int nd = 0; nd < 2; ++nd) {
        bf16x8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = vlo[nd][j];
          vf[4 + j] = vhi[nd][j];
        }
        o[nd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[cur], o[nd], 0, 0, 0);
      }
      if (it + 1 < NI && (it + 1 < NI - 2 || it + 1 < ni_valid)) {
        read_v(it + 1);            // LDS latency hides under the exp/convert block below
        make_p(it + 1, cur ^ 1);   // VALU work of the next slice runs while the two MFMAs execute
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float lsum = lsum2[0] + lsum2[1];
    const float l = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / l;
    const int q = qt * 32 + l31;
#ifdef VDR_ATTN_STAMPS
    asm volatile("" : "+v"(o[0]), "+v"(o[1]));
#endif
    VDR_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next item's K / V pieces and Q (issued long ago); see stage_write
    VDR_STAMP(5);
    // Retire the next item's Q fragments HERE, before the stores: hipcc meets their first "use" with a vmcnt wait of its
    // own, and placed after the stores that wait (vmcnt(0): it does not count the predicated stores) would sit out the
    // stores' acknowledgement every item.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));
    attn_store_row(p, o, inv, q < p.seq, (int64_t)b * p.seq + (q < p.seq ? q : 0), hd, hh);
    VDR_STAMP(6);
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  const bool computes = wave < nqt;  // nqt <= NCW (checked by the launcher)
  bf16x8 qf[4];
  stage_issue(item, smem);
  if (computes) load_q(item, wave, qf);
  stage_write(false);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));  // retired by the vmcnt(0) just above
  int cur = 0;
  const int stride = __builtin_amdgcn_readfirstlane(gridDim.x);  // (kept in an SGPR: the memory clobbers below would re-read it)
#ifdef VDR_ATTN_STAMPS
  int it_no = 0;
#endif
  for (; item < n_items; item += stride) {
    VDR_STAMP(0);
#ifdef VDR_ATTN_STAMPS
    st[9] = wall_clock64();  // constant 100 MHz: calibrates the shader clock the other stamps count
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // buffer `cur` holds this item; the other buffer is free
    asm volatile("" ::: "memory");
    VDR_STAMP(1);
    const int next = item + stride;
#ifdef VDR_TUNING
    const bool more = next < n_items && !(p.abl & 2);
    const bool do_compute = !(p.abl & 1);
#else
    const bool more = next < n_items;
    const bool do_compute = true;
#endif
    if (more) stage_issue(next, smem + (cur ^ 1) * BUF);
    if (computes && do_compute) compute_tile(item, next < n_items ? next : -1, wave, smem + cur * BUF, qf);
    if (!computes) VDR_STAMP(2);  // (loader / idle wave: its pieces are issued)
    if (more) stage_write(computes && do_compute);
    VDR_STAMP(7);
#ifdef VDR_ATTN_STAMPS
    st[10] = wall_clock64();
    if (lane == 0 && it_no < 16) {
      unsigned long long* d = p.stamps + (((size_t)blockIdx.x * 8 + wave) * 16 + it_no) * 16;
      for (int i = 0; i < 11; ++i) d[i] = st[i];
    }
    ++it_no;
#endif
    cur ^= 1;
  }
}

template <int NT, bool LOADER>
static hipError_t launch_persist(const AttnK& k, int batch, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(2 * NT * 32 * 128);
  auto fn = attn_persist_kernel<NT, LOADER>;
  static PerDeviceFlag attr;  // per instantiation and device: raised once, not per launch
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  if (!attr.done[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr.done[dev] = true;
  }
  int n_cu = device_cu_count(dev);
  if (n_cu <= 0) n_cu = 256;
  const int n_items = batch * k.heads;
  const int grid = n_items < n_cu ? n_items : n_cu;
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(LOADER ? 512 : 448), lds, s, k, n_items);
  return hipGetLastError();
}

template <int NT>
static hipError_t launch_nt(const AttnK& k, int batch, hipStream_t s) {
  constexpr size_t image = 2 * (size_t)NT * 32 * 128;       // K image + V image
  const size_t lds = image;
  auto fn = attn_kernel<NT>;
  static PerDeviceFlag attr;  // per instantiation and device: raised once, not per launch
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  if (image > 65536 && !attr.done[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)image);
    if (e != hipSuccess) return e;
    attr.done[dev] = true;
  }
  const int nqt = (k.seq + 31) / 32;
  const dim3 grid((unsigned)(batch * k.heads * ((nqt + k.qt_per_block - 1) / k.qt_per_block)));
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, k);
  return hipGetLastError();
}

hipError_t launch_attention(const void* qkv, void* out, int batch, int seq, int heads, int variant,
                            hipStream_t s, void* out_scale, const int* lens, int len_add) {
  if (batch <= 0 || seq <= 0 || heads <= 0) return hipErrorInvalidValue;
  AttnK k;
#ifdef VDR_TUNING
  k.abl = variant / 10;
  variant %= 10;
#else
  k.abl = 0;
  // 0 library choice, 1 chunked (online softmax), 2 persistent without the loader wave, 3 one-shot, 4 persistent with it
  if (variant < 0 || variant > 4) return hipErrorInvalidValue;  // (ablation encodings exist in tuning builds only)
#endif
  k.qkv = (const bf16_t*)qkv;
  k.out = (bf16_t*)out;
  k.o_scale = (uint8_t*)out_scale;
  k.lens = lens;
  k.len_add = len_add;
  k.os_rows = mx_rows_pad((int64_t)batch * seq);
  k.seq = seq;
  k.heads = heads;
  k.ld_qkv = (int64_t)3 * heads * 64;
  k.ld_out = (int64_t)heads * 64;
  const int nqt = (seq + 31) / 32;
  if (variant == 1 || seq > 288) {
    // online softmax over 128-key chunks, one query tile per wave
    k.qt_per_block = 4;
    k.n_chunks = (seq + 127) / 128;
    return launch_nt<4>(k, batch, s);
  }
  k.qt_per_block = nqt;
  k.n_chunks = 1;
  if ((variant == 2 || variant == 4 || variant == 0) && !lens) {  // (per-sequence lengths: one-shot kernel only)
    // persistent warp-specialised kernel (needs a few items per workgroup to pay off)
    if (variant == 4 && seq > 128 && seq <= 224) return launch_persist<7, true>(k, batch, s);
    if (variant == 0 && seq > 128 && seq <= 224 && batch * heads >= 512) return launch_persist<7, true>(k, batch, s);  // nqt <= 7 compute waves
    if (seq > 128 && seq <= 224 && batch * heads >= 512) return launch_persist<7, false>(k, batch, s);
    if (variant == 2) {
      if (seq <= 128) return launch_persist<4, false>(k, batch, s);
      if (seq <= 224) return launch_persist<7, false>(k, batch, s);
    }
  }
  if (seq <= 64) return launch_nt<2>(k, batch, s);
  if (seq <= 128) return launch_nt<4>(k, batch, s);
  if (seq <= 224) return launch_nt<7>(k, batch, s);
  return launch_nt<9>(k, batch, s);
}

}  // namespace vdr
