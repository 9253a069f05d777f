// PARKED EXPERIMENT (round 3; was csrc/gemm_stream.hip, tile variant 30 / vdr_config.stream_gemm of ABI 6): built, bitwise
// equal to ring4, not faster in the forward -- no longer compiled into libvdr.so.  tools/micro/stream_stamps.hip
// still builds it (-I vit-deep-radiomics_amd/csrc) for its stamps.
//
// Persistent "stream" GEMM for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T), bf16 in / fp32 accumulate / bf16 out.
//
// Replaces the same nn.Linear calls as gemm_kernels.h (reference src/models_archs.py:130-135; attn.qkv / attn.proj /
// mlp.fc1 / mlp.fc2 of the frozen ViTs called at src/tfds_dense_descriptor.py:123) for launches with many tiles.
//
// Why it exists (round-2 stamps of ring4, profiles/r02_gemm_stamps.txt): a 128 x 256 tile at K = 768 spends 69 % of its
// life in the K loop, 19 % in the epilogue, 8 % in the prologue, and a workgroup slot is occupied only 81-85 % of a
// launch.  Here none of the three exists as a phase:
//   * ONE 8-wave workgroup per CU (256 VGPRs per wave, 152 KB of LDS) walks a list of tiles;
//   * the K loop is a single stream of 64-deep steps that runs ACROSS tile boundaries: the LDS ring (3 stages of
//     [128 rows A | 256 rows W] x 128 B, whole cache lines of both operands, W in the plain PyTorch layout) never
//     drains, the loads of tile t+1's first steps are issued during tile t's last steps;
//   * a wave holds TWO accumulator sets: the finished one of tile t is turned into outputs (LayerNorm fold, bias,
//     erf-GELU / residual, bf16, stores) in slices placed between the MFMAs of tile t+1's first 8 steps -- inside one
//     wave vector and matrix instructions overlap almost for free (tools/micro/coissue.hip: 276 cycles for 8 MFMAs + 32
//     fmas against 256 + 176 in two waves), which is what the two-workgroups-per-CU form could not do;
//   * no LDS staging of the outputs: the W rows of a wave tile are assigned to MFMA row slots by a permutation
//     (slot i of column tile jt holds column 32 (jt >> 1) + 8 (i >> 2) + 4 (jt & 1) + (i & 3)) under which a lane's
//     accumulators of tiles (jt, jt+1) are 8 CONSECUTIVE output columns of one row: 16-byte stores straight from the
//     accumulator layout (16 rows x 64 B per instruction).
// Same products in the same order as ring3 / ring4 (32-deep MFMA units in ascending k, identical epilogue formulas):
// outputs are bitwise those of the other kernels (tests/test_ops_gpu.py).
//
// Synchronisation (one raw s_barrier per 64-deep step, in its middle): fragment reads run half a step ahead of the
// MFMAs, so at the middle of step g every wave has read stage g completely; behind the barrier the stage-(g+3) loads go
// into that slot.  A stage is waited for (counted vmcnt; every vector-memory operation of the loop is issued by inline
// assembly or a store builtin, so the counts are exact) two steps after it was issued.
#include "gemm_kernels.h"

namespace vdr {

struct StreamK {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  const float* colsum;
  const float* ln_stats;
  const bf16_t* resid;
  const float* gamma;
  bf16_t* C;
  float* ln_part;
  int64_t part_stride;
  int M, N, K;
  int lda, ldw, ldc, ldr;  // elements
  int tiles_m, tiles_n, gn, ntiles;
  int nt_store;
#ifdef VDR_STREAM_STAMPS
  unsigned long long* stamps;  // tools/micro/stream_stamps.hip: [workgroup][wave][8] summed phase durations (shader cycles)
#endif
};

constexpr int ST_BM = 128, ST_BN = 256;
constexpr int ST_STAGE = (ST_BM + ST_BN) * 128;  // 48 KB
constexpr int ST_WOFF = ST_BM * 128;             // W image behind the A image of a stage
constexpr int ST_CONST = 3 * ST_STAGE;           // per-tile constants: 2 x 4 KB (by tile parity)
constexpr int ST_LDS = ST_CONST + 2 * 4096;      // 155648 B
constexpr int ST_EU = 11;                        // unrolled head steps of a tile (they carry the previous tile's epilogue)

__device__ const float g_stream_zero[256] = {};
__device__ const float g_stream_one[256] = {
#define O8 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f
#define O64 O8, O8, O8, O8, O8, O8, O8, O8
    O64, O64, O64, O64
#undef O64
#undef O8
};

// one opaque LDS-DMA: 64-bit wave-uniform base + 32-bit per-lane byte offset -> LDS lds_addr + lane * SIZE
VDR_DEV void dma16(const void* base_uniform, uint32_t lane_off, uint32_t lds_addr) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base_uniform), "{m0}"(lds_addr) : "memory");
}
// ... through a buffer resource: lanes whose offset is not below num_records load nothing (rows past the end of the
// operand in edge tiles): the same instruction for interior and edge tiles, no branch in the instruction stream
VDR_DEV void dma16b(u32x4 srd, uint32_t lane_off, uint32_t lds_addr) {
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(lane_off), "s"(srd), "{m0}"(lds_addr) : "memory");
}
// ... issued by the loader waves only (`on` is wave-uniform): the others branch over the instruction INSIDE the asm
// statement, so the instruction stream the compiler sees stays one basic block per half step
VDR_DEV void dma16_if(uint32_t on, const void* base_uniform, uint32_t lane_off, uint32_t lds_addr) {
  on = __builtin_amdgcn_readfirstlane(on);  // (under SGPR pressure hipcc parks the flag in a VGPR and hands THAT to the "s" operand)
  asm volatile(
      "s_cmp_eq_u32 %[on], 0\n\t"
      "s_cbranch_scc1 .Lst_skip%=\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[vo], %[p]\n\t"
      ".Lst_skip%=:"
      ::[on] "s"(on), [vo] "v"(lane_off), [p] "s"(base_uniform), "{m0}"(lds_addr)
      : "memory", "scc");
}
VDR_DEV void dma4v_if(uint32_t on, const void* addr, uint32_t lds_addr) {
  on = __builtin_amdgcn_readfirstlane(on);
  asm volatile(
      "s_cmp_eq_u32 %[on], 0\n\t"
      "s_cbranch_scc1 .Lst_skip%=\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %[a], off\n\t"
      ".Lst_skip%=:"
      ::[on] "s"(on), [a] "v"(addr), "{m0}"(lds_addr)
      : "memory", "scc");
}
VDR_DEV void dma16v(const void* addr, uint32_t lds_addr) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(addr), "{m0}"(lds_addr) : "memory");
}
VDR_DEV void dma4v(const void* addr, uint32_t lds_addr) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(addr), "{m0}"(lds_addr) : "memory");
}
VDR_DEV bf16x8 lds_rd(uint32_t addr) { return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)addr); }

typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int N, int I = 0, typename F>
VDR_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
#define ST_I(x) std::integral_constant<int, (x)> {}

// Compile-time schedule of a tile's epilogue over the MFMA slots (32 per step) of the NEXT tile's first steps.
// Quantum q (accumulator tile 4 it + jt, 4 values per lane) owns slots [q ST_QS, (q + 1) ST_QS), its NSTG stages evenly
// spread; the 16-byte store of a quantum pair goes into the first FIRST-half slot behind the pair's last stage (stores
// in first halves only: the mid-step wait of a loader wave then counts  12 + stores of this and the previous step (+ the
// 3 constant loads of step 0); the other waves have no loads to wait for).
constexpr int ST_QS = 18;  // slots per quantum: 16 quanta end in step 8, the last store sits in step 9
constexpr int st_nstg(int epi) { return epi == EPI_BIAS_GELU ? 14 : 4; }
constexpr int st_stage_at(int S, int epi) {  // 32 q + j of the stage placed in slot S, or -1
  const int q = S / ST_QS, r = S % ST_QS, n = st_nstg(epi);
  if (q >= 16) return -1;
  for (int j = 0; j < n; ++j)
    if ((j * ST_QS) / n == r) return q * 32 + j;
  return -1;
}
constexpr int st_store_slot(int k, int epi) {
  const int n = st_nstg(epi);
  int s = (2 * k + 1) * ST_QS + ((n - 1) * ST_QS) / n + 1;
  while ((s / 16) % 2 != 0) s = (s / 16 + 1) * 16;
  return s;
}
constexpr int st_store_at(int S, int epi) {
  for (int k = 0; k < 8; ++k)
    if (st_store_slot(k, epi) == S) return k;
  return -1;
}
constexpr int st_stores_in_step(int s, int epi) {
  int n = 0;
  for (int k = 0; k < 8; ++k)
    if (st_store_slot(k, epi) / 32 == s) ++n;
  return n;
}
constexpr int st_nv(int s, int epi) {  // vmcnt of the mid-step wait of tile step s
  return 12 + (s >= 1 ? st_stores_in_step(s - 1, epi) : 0) + st_stores_in_step(s, epi) + (s == 1 || s == 2 ? 3 : 0);
}

struct StTile {
  int m0, n0;
  bool valid;
};

template <int EPI, bool FOLD, bool NT>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(StreamK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, wave tile 64 x 64
  const int r15 = lane & 15, q4 = lane >> 4;

  // ---- fragment read addresses (relative to a stage) ------------------------------------------------------------
  // A image [128][128 B], 16-B slot = chunk ^ ((row >> 1) & 7); lane (r15, q4) reads chunk 4 h + q4 of row 16 it + r15
  const uint32_t a_rd0 = (uint32_t)((wm * 64 + r15) * 128 + ((q4 ^ ((r15 >> 1) & 7)) << 4));
  // W image [256][128 B], slot = chunk ^ (bit 1 | bits 3,4 << 1 of the row): invariant under the +4 / +32 / +64 row
  // offsets of the column-tile permutation, so the four tiles of a wave are immediate offsets of one address
  const int rowl = ((r15 >> 2) << 3) + (r15 & 3);
  const int sww = ((rowl >> 1) & 1) | (((rowl >> 3) & 3) << 1);
  const uint32_t b_rd0 = (uint32_t)(ST_WOFF + (wn * 64 + rowl) * 128 + ((q4 ^ sww) << 4));
  constexpr int JOFF[4] = {0, 4 * 128, 32 * 128, 36 * 128};

  // ---- loader: waves 0-3 bring the whole stage, wave w the pieces q = w + 4 j (8 rows x 128 B each) of A (j < 4) and W
  // (j < 8); waves 4-7 only compute.  Measured with every wave loading its share (tools/micro/stream_stamps.hip): of a
  // SIMD's two waves the older one wins the arbitration, finishes its step early and waits ~560 cycles at the barrier
  // while the younger, running alone, is what the step takes -- the loads go where the slack is.
  const uint32_t loader = wave < 4 ? 1u : 0u;
  const int r3 = lane >> 3, c8 = lane & 7;
  const uint32_t a_voff = (uint32_t)(r3 * p.lda * 2 + ((c8 ^ ((4 * wave + (r3 >> 1)) & 7)) << 4));
  const uint32_t w_voff = (uint32_t)(r3 * p.ldw * 2 + ((c8 ^ (((r3 >> 1) & 1) | ((wave & 3) << 1))) << 4));
  static_assert(ST_STAGE % 128 == 0 && ST_WOFF % 128 == 0, "the half-select XOR assumes 128-B aligned images");
  const int nk = p.K >> 6;

  // ---- tile list (static stride over the resident workgroups; ids of one XCD are neighbours) ----------------------
  int next_id = blockIdx.x;
  auto fetch_tile = [&]() -> StTile {
    StTile t;
    t.valid = next_id < p.ntiles;
    int tm = 0, tn = 0;
    if (t.valid) {
      const int wg = xcd_remap(next_id, p.ntiles);
      if (p.gn <= 0 || p.gn >= p.tiles_n) {
        tm = wg / p.tiles_n;
        tn = wg - tm * p.tiles_n;
      } else {
        const int per_group = p.tiles_m * p.gn;
        const int g = wg / per_group;
        const int r = wg - g * per_group;
        const int width = min(p.gn, p.tiles_n - g * p.gn);
        tm = r / width;
        tn = g * p.gn + (r - tm * width);
      }
    }
    next_id += gridDim.x;
    t.m0 = tm * ST_BM;
    t.n0 = tn * ST_BN;
    return t;
  };

  // ---- DMA cursor -------------------------------------------------------------------------------------------------
  // dt: tile the loader is in, dk: its next 64-deep step, dn: the tile after dt (fetched once per tile at the compute
  // tile boundary, away from the steps).  Past the last tile the loader re-reads the last tile: same operation count
  // for the counted waits, nobody consumes it.
  StTile dt = fetch_tile();
  int dk = 0;
  StTile dn = fetch_tile();
  bool crossed_into_new = false;  // the loader's last tile change entered a real tile
  // running source of the loader: row 8 (wave & 3) of the tile's A / W panel at step dk; piece j adds 32 j rows.  The
  // rows of an edge tile past M are read like the others (the caller guarantees them readable: GemmArgs::a_rows; their
  // products are never stored), N is a multiple of the tile width.
  const char* a_ptr;
  const char* w_ptr;
  const uint32_t a_j32 = (uint32_t)p.lda * 64u, w_j32 = (uint32_t)p.ldw * 64u;  // bytes of 32 rows
  auto dma_rebase = [&]() {
    a_ptr = reinterpret_cast<const char*>(p.A + (int64_t)(dt.m0 + 8 * (wave & 3)) * p.lda);
    w_ptr = reinterpret_cast<const char*>(p.W + (int64_t)(dt.n0 + 8 * (wave & 3)) * p.ldw);
  };
  dma_rebase();
  // piece J of the stage at (dt, dk): J < 4 an A piece, else a W piece
  auto dma_piece = [&](auto j_tag, int slot) {
    constexpr int J = decltype(j_tag)::value;
    constexpr bool ISA = J < 4;
    constexpr int j = ISA ? J : J - 4;
    const uint32_t d = lds0 + (uint32_t)slot * ST_STAGE + (uint32_t)(wave & 3) * 1024 + (ISA ? 0 : ST_WOFF) + j * 4096;
#ifndef ST_ABL_NODMA
    dma16((ISA ? a_ptr : w_ptr) + (uint64_t)(j * (ISA ? a_j32 : w_j32)), ISA ? a_voff : w_voff, d);
#endif
  };
  auto dma_advance = [&]() {
    a_ptr += 128;
    w_ptr += 128;
    if (++dk == nk) {
      dk = 0;
      crossed_into_new = dn.valid;
      if (dn.valid) dt = dn;
      dma_rebase();
    }
  };
  // per-tile constants -> LDS (3 operations per loader wave w: 64 columns of the bias, of the column sums (or LayerScale),
  // and the statistics of 32 rows)
  auto dma_consts = [&](auto which_tag, const StTile& t, int par) {
    constexpr int WHICH = decltype(which_tag)::value;
    const uint32_t d = lds0 + ST_CONST + (uint32_t)par * 4096 + (uint32_t)(wave & 3) * 256;
    if constexpr (WHICH < 2) {
      int col = t.n0 + (wave & 3) * 64 + lane;
      const bool cok = col < p.N;
      col = cok ? col : 0;
      const float* sel = WHICH == 0 ? p.bias : (EPI == EPI_BIAS_RESID ? p.gamma : (FOLD ? p.colsum : nullptr));
      const float* dflt = WHICH == 1 && EPI == EPI_BIAS_RESID ? g_stream_one : g_stream_zero;
      const float* src = sel && cok ? sel + col : dflt + lane;
      dma4v(src, d + WHICH * 1024);
    } else {
      // (mean, rstd) of rows m0 + 32 (wave & 3) + lane / 2
      int row = t.m0 + (wave & 3) * 32 + (lane >> 1);
      row = row < p.M ? row : p.M - 1;
      const float* ssrc = FOLD ? p.ln_stats + 2 * (int64_t)row + (lane & 1) : g_stream_zero + lane;
      dma4v(ssrc, d + 2048);
    }
  };

  // ---- epilogue state of the previous tile -------------------------------------------------------------------------
  f32x4 acc[4][4], prev[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) prev[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  StTile et;
  et.valid = false;
  et.m0 = et.n0 = 0;
  int epar = 0;  // parity of the constants of the tile in `prev`
  __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0, 0x00020000);
  uint32_t c_voff = 0;
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u;
  auto epi_setup = [&]() {  // wave-tile view of C for the tile in `prev`
    const int64_t mb = (int64_t)et.m0 + wm * 64;
    const int nb = et.n0 + wn * 64;
    const int64_t left = (int64_t)p.M - mb;
    const int valid = !et.valid ? 0 : left >= 64 ? 64 : (left > 0 ? (int)left : 0);
    c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + mb * p.ldc + nb), 0, (int)((uint32_t)valid * ldc2), 0x00020000);
    const uint32_t v = (uint32_t)r15 * ldc2 + (uint32_t)q4 * 16u;
    c_voff = nb < p.N ? v : 0x7fffffffu;  // (N % 64 == 0: a wave tile is in range as a whole or not at all)
  };
  epi_setup();

  // The epilogue of one accumulator tile (quantum Q = 4 it + jt: 4 values per lane) as NSTG stages of ~4 independent vector
  // instructions, each placed behind one MFMA of the next tile; state of the quantum in flight:
  f32x4 e_b, e_c;      // bias and column sums of the lane's 4 columns
  float e_rs = 1.0f, e_nrm = 0.0f;  // rstd and -rstd * mean of its row
  float e_v[4], e_a[4], e_r[4], e_q[4];
  uint32_t e_pk[2][2];  // packed bf16 pairs of the even / odd quantum of a pair, until their 16-byte store
  auto epi_stage = [&](auto q_tag, auto j_tag) {
    constexpr int Q = decltype(q_tag)::value, J = decltype(j_tag)::value;
    constexpr int it = Q >> 2, jt = Q & 3;
    constexpr bool GELU = EPI == EPI_BIAS_GELU;
    constexpr int LAST = GELU ? 13 : 3;
    if constexpr (J == 0) {
      const uint32_t cst = lds0 + ST_CONST + (uint32_t)epar * 4096;
      const int colq = wn * 64 + 32 * (jt >> 1) + 8 * q4 + 4 * (jt & 1);
      e_b = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(cst + colq * 4));
      if constexpr (FOLD) {
        e_c = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(cst + 1024 + colq * 4));
        const f32x2 st = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((uintptr_t)(cst + 2048 + (wm * 64 + it * 16 + r15) * 8));
        e_rs = st[1];
        e_nrm = -st[1] * st[0];
      }
    } else if constexpr (J == 1) {
      // rs (acc - mu c) + b  =  rs acc + (b - rs mu c): the same two fused multiply-adds as epilogue_bf16 (gemm_epi.h)
#pragma unroll
      for (int e = 0; e < 4; ++e) e_v[e] = FOLD ? fmaf(e_nrm, e_c[e], e_b[e]) : e_b[e];
    } else if constexpr (J == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) e_v[e] = FOLD ? fmaf(e_rs, prev[jt][it][e], e_v[e]) : prev[jt][it][e] + e_v[e];
    } else if constexpr (GELU && J == 3) {  // gelu_erf (vdr_dev.h), one operation of its chain per stage
#pragma unroll
      for (int e = 0; e < 4; ++e) asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(e_a[e]) : "v"(e_v[e]), "v"(5.7f));
    } else if constexpr (GELU && J == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) asm("v_max_f32_e32 %0, 0, %1" : "=v"(e_r[e]) : "v"(e_v[e]));
    } else if constexpr (GELU && J == 5) {
#pragma unroll
      for (int e = 0; e < 4; ++e) e_q[e] = fmaf(2.480073296e-05f, e_a[e], -6.399250922e-04f);
    } else if constexpr (GELU && J >= 6 && J <= 10) {
      constexpr float C[5] = {7.365777341e-03f, -5.164207073e-02f, -4.607286841e-01f, -1.150403490e+00f, -1.000050145e+00f};
#pragma unroll
      for (int e = 0; e < 4; ++e) e_q[e] = fmaf(e_q[e], e_a[e], C[J - 6]);
    } else if constexpr (GELU && J == 11) {
#pragma unroll
      for (int e = 0; e < 4; ++e) e_q[e] = fast_exp2(e_q[e]);
    } else if constexpr (GELU && J == 12) {
#pragma unroll
      for (int e = 0; e < 4; ++e) e_v[e] = e_r[e] - e_a[e] * e_q[e];
    } else if constexpr (J == LAST) {
      bf16x4 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) ob[e] = (bf16_t)e_v[e];
      const u32x2 w2 = __builtin_bit_cast(u32x2, ob);
      e_pk[jt & 1][0] = w2[0];
      e_pk[jt & 1][1] = w2[1];
    }
  };
  // 16-byte store of quantum pair K2 (tiles (it, 2 jp) and (it, 2 jp + 1): 8 consecutive columns of one row)
  auto epi_store = [&](auto k_tag) {
    constexpr int K2 = decltype(k_tag)::value;
    constexpr int it = K2 >> 1, jp = K2 & 1;
    u32x4 v;
    v[0] = e_pk[0][0];
    v[1] = e_pk[0][1];
    v[2] = e_pk[1][0];
    v[3] = e_pk[1][1];
    // (row step in the VGPR offset: the hardware's range check covers voffset + immediate, rows past M are dropped; a
    // masked lane's 0x7fffffff stays out of range)
    const uint32_t voff = c_voff + (uint32_t)(it * 16) * ldc2 + (uint32_t)(jp * 64);
    __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, voff, 0, NT ? 2 : 0);
  };

  // ---- prologue: stages 0..2 of the first tile ---------------------------------------------------------------------
  for (int st = 0; st < 3; ++st) {
    if (loader) static_for<12>([&](auto j) { dma_piece(j, st); });
    dma_advance();
  }
  StTile ct = dt;  // compute tile == first tile (nk > 3: the loader has not left it)
  int cpar = 0;
  wait_vmcnt<24>();  // stage 0 landed (this wave's pieces)
  asm volatile("s_barrier" ::: "memory");

  bf16x8 fa[2][4], fb[2][4];
  int slot = 0;
#ifdef VDR_STREAM_STAMPS
  // diagnostic build: s_memtime around the mid-step synchronisation of every step, summed per wave (each stamp waits for
  // its own return: ~4 x 60 cycles per step of perturbation)
  uint32_t ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev, t_a, t_b, t_c, t_d;
  const unsigned long long wall0 = wall_clock64();
#define ST_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
  ST_STAMP(t_prev);
#else
#define ST_STAMP(v)
#endif
  // fragment X of a 32-deep half of a stage: X < 4 W tile X (B operand slot), else A row tile X - 4; (aa, bb) = the
  // lane's addresses for that half (half_addr), the tiles are immediate offsets
  uint32_t rd_aa = 0, rd_bb = 0;
  auto half_addr = [&](int sl, int h) {
    const uint32_t base = lds0 + (uint32_t)sl * ST_STAGE;
    // (second 32-deep half of the 128-B rows: chunk + 4 = byte offset ^ 64, the images are 128-B aligned)
    rd_aa = base + (a_rd0 ^ (uint32_t)(h << 6));
    rd_bb = base + (b_rd0 ^ (uint32_t)(h << 6));
  };
  auto rd_frag = [&](auto x_tag, bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    constexpr int X = decltype(x_tag)::value;
    if constexpr (X < 4) b[X] = lds_rd(rd_bb + JOFF[X]);
    else a[X - 4] = lds_rd(rd_aa + (X - 4) * 2048);
  };
  half_addr(0, 0);
  static_for<8>([&](auto x) { rd_frag(x, fa[0], fb[0]); });
  wait_vmcnt<12>();  // stage 1 landed: the first step reads its first half before the first mid-step barrier
  asm volatile("s_barrier" ::: "memory");

  // One 64-deep step = 2 x 16 MFMA slots.  Behind the MFMA of a slot: one fragment read of the NEXT half (W tiles under
  // slots 0-3, A row tiles under 8, 10, 12, 14: an A fragment is needed 4 slots later than the one before it, so the
  // late reads keep 8 registers free), in the second half one LDS-DMA piece of stage g + 3 (slots 1, 3, .., 11: issued
  // together behind the barrier they idle the matrix pipe for the time all eight waves spend on them) and at most one
  // epilogue stage / store of the previous tile.  Every slot is its own scheduling region: the order below is the order
  // of the instruction stream.
  // SI: step index in the tile for the epilogue schedule (-1: none).  MODE 0: body, 1: first step of a tile
  // (accumulators start from zero), 2: last step (results go to `prev`).  NV: vmcnt of the mid-step wait.
  auto step = [&](auto si_tag, auto mode_tag, auto nv_tag) {
    constexpr int SI = decltype(si_tag)::value, MODE = decltype(mode_tag)::value, NV = decltype(nv_tag)::value;
    const int nslot = slot == 2 ? 0 : slot + 1;
    half_addr(slot, 1);
#if defined(ST_ABL_XIDLE)
    const bool do_mma = wave >= 4;  // diagnostic: waves 0-3 only synchronise -> what waves 4-7 do alone on their SIMDs
#elif defined(ST_ABL_YIDLE)
    const bool do_mma = wave < 4;
#else
    constexpr bool do_mma = true;
#endif
    if (do_mma)
    static_for<16>([&](auto ls_tag) {
      constexpr int LS = decltype(ls_tag)::value, i = LS >> 2, j = LS & 3;
      if constexpr (MODE == 1) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], acc[j][i], 0, 0, 0);
      if constexpr (LS < 4) rd_frag(ST_I(LS), fa[1], fb[1]);
      else if constexpr (LS >= 8 && !(LS & 1)) rd_frag(ST_I(4 + (LS - 8) / 2), fa[1], fb[1]);
      if constexpr (SI >= 0) {
        constexpr int S = SI * 32 + LS, sg = st_stage_at(S, EPI), sk = st_store_at(S, EPI);
        if constexpr (sg >= 0) epi_stage(ST_I(sg >> 5), ST_I(sg & 31));
        if constexpr (sk >= 0) epi_store(ST_I(sk));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- middle: this wave has read stage `slot` completely; its pieces of the next stage have landed
    ST_STAMP(t_a);  // (sampled at issue, returns behind the LDS reads: t_b - t_a = what the fragment reads still took)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ST_STAMP(t_b);
    wait_vmcnt<NV>();
    ST_STAMP(t_c);
    asm volatile("s_barrier" ::: "memory");
    ST_STAMP(t_d);
    // The loader waves issue the whole stage g + 3 (12 pieces each) HERE, in one burst, before they resume their MFMAs;
    // their SIMD partners go straight on, alone on the matrix pipe for the length of the burst.  The CU's address path
    // takes ~21 cycles per 1-KB piece whoever issues it (measured: ~1000 cycles for the 48 pieces of a stage, about the
    // length of the step's 1024 MFMA cycles): (i) one piece behind each MFMA slot of every wave made each slot longer
    // than its MFMA (~6 scalar instructions + the piece); (ii) the same for the loaders only, the others branching over
    // each piece inside the asm statement, cost the others a taken branch per slot; (iii) two copies of the second half
    // behind one wave-uniform branch made hipcc spill 120-208 bytes per lane.  (tools/micro/stream_stamps.hip)
    if (loader) {
      static_for<12>([&](auto j) { dma_piece(j, slot); });
      if constexpr (SI == 0) static_for<3>([&](auto c) { dma_consts(c, ct, cpar); });
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef VDR_STREAM_STAMPS
    ph[0] += (uint32_t)(t_a - t_prev);  // second half of the step before + first half of this one
    ph[1] += (uint32_t)(t_b - t_a);
    ph[2] += (uint32_t)(t_c - t_b);
    ph[3] += (uint32_t)(t_d - t_c);
    ph[4] += 1;
    t_prev = t_d;
#endif
    half_addr(nslot, 0);
    // Second half: the same 16 slots for every wave.
    if (do_mma)
    static_for<16>([&](auto ls_tag) {
      constexpr int LS = decltype(ls_tag)::value, i = LS >> 2, j = LS & 3;
      if constexpr (MODE == 2) prev[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[j][i], 0, 0, 0);
      else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[j][i], 0, 0, 0);
      if constexpr (LS < 4) rd_frag(ST_I(LS), fa[0], fb[0]);
      else if constexpr (LS >= 8 && !(LS & 1)) rd_frag(ST_I(4 + (LS - 8) / 2), fa[0], fb[0]);
      if constexpr (SI >= 0) {
        constexpr int S = SI * 32 + 16 + LS, sg = st_stage_at(S, EPI);
        static_assert(st_store_at(S, EPI) < 0, "stores sit in first halves only (vmcnt bookkeeping)");
        if constexpr (sg >= 0) epi_stage(ST_I(sg >> 5), ST_I(sg & 31));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    dma_advance();
    slot = nslot;
  };

  for (;;) {
    static_for<ST_EU>([&](auto si) {
      constexpr int SI = decltype(si)::value;
      step(si, ST_I(SI == 0 ? 1 : 0), ST_I(st_nv(SI, EPI)));
    });
    for (int s = ST_EU; s < nk - 1; ++s) step(ST_I(-1), ST_I(0), ST_I(12));
    step(ST_I(-1), ST_I(2), ST_I(12));
    // tile boundary: `prev` holds tile ct
    et = ct;
    epar = cpar;
    epi_setup();
    if (!crossed_into_new) break;  // the loader never left this tile: it was the last
    ct = dt;
    cpar ^= 1;
    dn = fetch_tile();
  }
#ifdef VDR_STREAM_STAMPS
  unsigned long long t_loop_end;
  ST_STAMP(t_loop_end);
#endif
  // ---- flush: epilogue of the last tile, nothing to hide it under ---------------------------------------------------
  wait_vmcnt<0>();
  asm volatile("s_barrier" ::: "memory");
  static_for<16>([&](auto q) {
    constexpr int Q = decltype(q)::value;
    static_for<(EPI == EPI_BIAS_GELU ? 14 : 4)>([&](auto j) { epi_stage(q, j); });
    if constexpr (Q & 1) epi_store(ST_I(Q >> 1));
  });
#ifdef VDR_STREAM_STAMPS
  {
    unsigned long long t_end;
    ST_STAMP(t_end);
    if (lane == 0 && p.stamps) {
      unsigned long long* d = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
      d[0] = ph[0];
      d[1] = ph[1];
      d[2] = ph[2];
      d[3] = ph[3];
      d[4] = ph[4];
      d[5] = t_end - t_loop_end;
      d[6] = wall0;
      d[7] = wall_clock64();
    }
  }
#endif
}

#ifdef VDR_STREAM_STAMPS
inline unsigned long long* g_stream_stamps = nullptr;  // tools/micro/stream_stamps.hip
#endif

static bool stream_shape_ok(const GemmArgs& a, int epi) {
  if (a.K % 64 || a.K / 64 < ST_EU + 1 || a.N % ST_BN || a.M <= 0) return false;
  // the loader reads whole tiles: the rows of the last tile row past M must be readable memory (the engine's workspace
  // buffers are; GemmArgs::a_rows says how many rows the caller guarantees)
  if ((a.M + ST_BM - 1) / ST_BM * ST_BM > (a.a_rows > a.M ? a.a_rows : a.M)) return false;
  if (a.w_interleaved || a.out_f32 || a.win_ws || a.a_rpg || a.patch_p || a.ln_cpart || a.ln_part) return false;
  if (epi != EPI_BIAS && epi != EPI_BIAS_GELU) return false;
  if (a.M >= (1 << 30) || a.ldc >= (1 << 24) || a.lda >= (1 << 24) || a.ldw >= (1 << 24)) return false;
  if (a.ln_stats && !a.colsum) return false;
  return true;
}

bool gemm_stream_eligible(const GemmArgs& a, int epi) {
  return stream_shape_ok(a, epi) && ((a.M + ST_BM - 1) / ST_BM) * (int64_t)(a.N / ST_BN) >= 1024;
}

hipError_t launch_gemm_stream(const GemmArgs& a, int epi, hipStream_t s) {
  if (!stream_shape_ok(a, epi)) return hipErrorInvalidValue;
  StreamK k{};
  k.A = (const bf16_t*)a.A;
  k.W = (const bf16_t*)a.W;
  k.bias = a.bias;
  k.colsum = a.colsum;
  k.ln_stats = a.ln_stats;
  k.resid = (const bf16_t*)a.resid;
  k.gamma = a.gamma;
  k.C = (bf16_t*)a.C;
  k.ln_part = a.ln_part;
  k.part_stride = a.part_stride;
  k.M = (int)a.M;
  k.N = a.N;
  k.K = a.K;
  k.lda = (int)a.lda;
  k.ldw = (int)a.ldw;
  k.ldc = (int)a.ldc;
  k.ldr = (int)a.ldr;
  k.tiles_m = (int)((a.M + ST_BM - 1) / ST_BM);
  k.tiles_n = (a.N + ST_BN - 1) / ST_BN;
  {
    const size_t panel = (size_t)ST_BN * a.K * 2;
    int gn = (int)((1700u << 10) / panel);
    if (gn < 2 || gn >= k.tiles_n) gn = 0;
    k.gn = gn;
  }
  const int64_t nt = (int64_t)k.tiles_m * k.tiles_n;
  if (nt > 0x7fffffff) return hipErrorInvalidValue;
  k.ntiles = (int)nt;
  k.nt_store = (double)a.M * (double)a.ldc * 2.0 >= 128e6 && !a.resid;
  const bool fold = a.ln_stats != nullptr;
  if (fold && !a.colsum) return hipErrorInvalidValue;
#ifdef VDR_STREAM_STAMPS
  k.stamps = g_stream_stamps;
#endif

  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  const int n_cu = device_cu_count(dev);
  if (n_cu <= 0) return hipErrorUnknown;
  const int grid = k.ntiles < n_cu ? k.ntiles : n_cu;
#define ST_LAUNCH2(E, F, NTV)                                                                                            \
  {                                                                                                                 \
    static PerDeviceFlag attr;                                                                                      \
    if (!attr.done[dev]) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_stream_kernel<E, F, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS); \
      if (e != hipSuccess) return e;                                                                                \
      attr.done[dev] = true;                                                                                           \
    }                                                                                                               \
    hipLaunchKernelGGL((gemm_stream_kernel<E, F, NTV>), dim3(grid), dim3(512), ST_LDS, s, k);                       \
  }
#define ST_LAUNCH(E, F)              \
  if (k.nt_store) ST_LAUNCH2(E, F, true) \
  else ST_LAUNCH2(E, F, false)
  if (epi == EPI_BIAS) {
    if (fold) ST_LAUNCH(EPI_BIAS, true) else ST_LAUNCH(EPI_BIAS, false)
  } else {
    if (fold) ST_LAUNCH(EPI_BIAS_GELU, true) else ST_LAUNCH(EPI_BIAS_GELU, false)
  }
#undef ST_LAUNCH
#undef ST_LAUNCH2
  return hipGetLastError();
}

}  // namespace vdr
