// EXPERIMENT, not shipped (tools/micro/stream_stamps.hip -DST_MFMA32 includes it instead of csrc/gemm_stream.hip):
// the stream GEMM rewritten on v_mfma_f32_32x32x16_bf16, every wave issuing 6 of the step's 48 LDS-DMA pieces.
// Result (profiles/r03_stream_gemm_stamps.txt, "32x32x16" section): bitwise the same outputs as the 16x16x32 kernel,
// but slower -- qkv 226.5 / fc1 364.3 / K=3072 234.6 us against 161.1 / 245.5 / 230.4 us.  A wave alone needs 48 cycles
// per MFMA (42.7 with no fragment reads and no DMA at all), and a store instruction of this accumulator layout touches
// 32 rows x 32 B instead of 16 rows x 64 B, which doubles the write transactions of the write-heavy launches.
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
//   * the MFMA is v_mfma_f32_32x32x16_bf16: it holds the issue port for 8 of its 32 cycles, so ONE wave can keep the pipe
//     busy with a fragment read, an LDS-DMA piece and epilogue stages in every gap (with the 16x16x32 shape, 8 of 16 cycles,
//     one wave alone needed 25.5 cycles per MFMA in this loop: tools/micro/stream_stamps.hip, XIDLE / YIDLE ablations);
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

struct StTile {
  int m0, n0;
  bool valid;
};

template <int N, int I = 0, typename F>
VDR_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
#define ST_I(x) std::integral_constant<int, (x)> {}

// Compile-time schedule of a tile's epilogue over the MFMA slots (16 per 64-deep step) of the NEXT tile's first ST_ES
// steps.  The epilogue of a wave's 64 x 64 tile is 16 quanta (4 values per lane each) of NSTG stages of four independent
// vector instructions; the stages, in order, are spread evenly over the slots (0, 1 or 2 per slot); the 16-byte store of a
// quantum pair goes into the first FIRST-half slot behind the pair's last stage (stores in first halves only: the
// mid-step wait then counts  6 + the stores of this and the previous step (+ the 2 constant loads of step 0)).
constexpr int ST_ES = 10;                 // steps that carry epilogue stages
constexpr int ST_SLOTS = ST_ES * 16;
constexpr int st_nstg(int epi) { return epi == EPI_BIAS_GELU ? 14 : 4; }
constexpr int st_first_stage(int S, int epi) { return S >= ST_SLOTS ? 16 * st_nstg(epi) : (S * 16 * st_nstg(epi)) / ST_SLOTS; }
constexpr int st_store_slot(int k, int epi) {  // store of quanta 2k, 2k + 1
  const int need = (2 * k + 2) * st_nstg(epi);  // stages that must have run
  int s = 0;
  while (st_first_stage(s, epi) < need) ++s;    // first slot at whose START they have
  while ((s / 8) % 2 != 0) s = (s / 8 + 1) * 8;  // first half of a step: slots 0..7 of 16
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
    if (st_store_slot(k, epi) / 16 == s) ++n;
  return n;
}
constexpr int st_nv(int s, int epi) {  // vmcnt of the mid-step wait of tile step s
  return 6 + (s >= 1 ? st_stores_in_step(s - 1, epi) : 0) + st_stores_in_step(s, epi) + (s == 1 || s == 2 ? 2 : 0);
}
constexpr bool st_schedule_ok(int epi) {
  for (int k = 0; k < 8; ++k) {
    if (st_store_slot(k, epi) / 16 >= ST_EU) return false;               // every store inside the unrolled steps
    if (k && st_store_slot(k, epi) == st_store_slot(k - 1, epi)) return false;
    // the packed halves of pair k must not be overwritten before their store: the next even quantum's last stage runs later
    if (k < 7 && st_first_stage(st_store_slot(k, epi) + 1, epi) > (2 * k + 3) * st_nstg(epi) - 1) return false;
  }
  return true;
}
static_assert(st_schedule_ok(EPI_BIAS) && st_schedule_ok(EPI_BIAS_GELU), "epilogue schedule");

template <int EPI, bool FOLD, bool NT>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(StreamK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, wave tile 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
  const int r31 = lane & 31, hh = lane >> 5;

  // ---- fragment read addresses (relative to a stage) ------------------------------------------------------------
  // Both images are [rows][128 B] with the 16-B slot = chunk ^ ((row >> 1) & 7).  The 32x32x16 operand of 16-deep
  // substep ks is chunk 2 ks + hh of row r31 (+ 32 per MFMA tile): conflict-free for ds_read_b128's lane groups.
  // W row slot s of a 32-column tile holds column 16 ((s >> 2) & 1) + 4 (s >> 3) + (s & 3): the accumulator registers
  // of a lane (slots 8 g + 4 hh + e) are then the 16 CONSECUTIVE columns 16 hh + 4 g + e of its row.
  const int colp = 16 * ((r31 >> 2) & 1) + 4 * (r31 >> 3) + (r31 & 3);
  const uint32_t a_row = (uint32_t)((wm * 64 + r31) * 128), a_sw = (uint32_t)((r31 >> 1) & 7);
  const uint32_t b_row = (uint32_t)(ST_WOFF + (wn * 64 + colp) * 128), b_sw = (uint32_t)((colp >> 1) & 7);
  uint32_t a_ad[4], b_ad[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_ad[ks] = a_row + ((((uint32_t)(2 * ks + hh)) ^ a_sw) << 4);
    b_ad[ks] = b_row + ((((uint32_t)(2 * ks + hh)) ^ b_sw) << 4);
  }

  // ---- loader: wave w brings the pieces q = w + 8 j (8 rows x 128 B each) of A (j < 2) and of W (j < 4) of every stage.
  // One wave alone sustains the matrix pipe with the 32x32x16 MFMA (24 of its 32 cycles leave the issue port free), so
  // the ~21 cycles the CU's address path takes per piece are covered by the SIMD partner's MFMAs whoever issues them.
  const int r3 = lane >> 3, c8 = lane & 7;
  const uint32_t d_sw = (uint32_t)(c8 ^ ((4 * wave + (r3 >> 1)) & 7)) << 4;  // rows 8 (w + 8 j) + r3: (row >> 1) & 7
  const uint32_t a_voff = (uint32_t)(r3 * p.lda * 2) + d_sw;
  const uint32_t w_voff = (uint32_t)(r3 * p.ldw * 2) + d_sw;
  static_assert(ST_STAGE % 128 == 0 && ST_WOFF % 128 == 0, "128-B aligned images");
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
  // running source of the loader: row 8 w of the tile's A / W panel at step dk; piece j adds 64 j rows.  The rows of an edge
  // tile past M are read like the others (the caller guarantees them readable: GemmArgs::a_rows; their products are never
  // stored), N is a multiple of the tile width.
  const char* a_ptr;
  const char* w_ptr;
  const uint32_t a_j64 = (uint32_t)p.lda * 128u, w_j64 = (uint32_t)p.ldw * 128u;  // bytes of 64 rows
  auto dma_rebase = [&]() {
    a_ptr = reinterpret_cast<const char*>(p.A + (int64_t)(dt.m0 + 8 * wave) * p.lda);
    w_ptr = reinterpret_cast<const char*>(p.W + (int64_t)(dt.n0 + 8 * wave) * p.ldw);
  };
  dma_rebase();
  auto dma_piece = [&](auto j_tag, int slot) {  // J < 2: an A piece, else a W piece
    constexpr int J = decltype(j_tag)::value;
    constexpr bool ISA = J < 2;
    constexpr int j = ISA ? J : J - 2;
    const uint32_t d = lds0 + (uint32_t)slot * ST_STAGE + (uint32_t)wave * 1024 + (ISA ? 0 : ST_WOFF) + j * 8192;
#ifndef ST_ABL_NODMA
    dma16((ISA ? a_ptr : w_ptr) + (uint64_t)(j * (ISA ? a_j64 : w_j64)), ISA ? a_voff : w_voff, d);
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
  // per-tile constants -> LDS (2 operations per wave): WHICH 0: waves 0-3 bias, 4-7 column sums (or LayerScale); 1: the
  // (mean, rstd) of 32 rows (waves 4-7 write a duplicate behind them)
  auto dma_consts = [&](auto which_tag, const StTile& t, int par) {
    constexpr int WHICH = decltype(which_tag)::value;
    const uint32_t d = lds0 + ST_CONST + (uint32_t)par * 4096 + (uint32_t)wave * 256;
    if constexpr (WHICH == 0) {
      int col = t.n0 + (wave & 3) * 64 + lane;
      const bool cok = col < p.N;
      col = cok ? col : 0;
      const float* hi = EPI == EPI_BIAS_RESID ? p.gamma : (FOLD ? p.colsum : nullptr);
      const float* dflt = EPI == EPI_BIAS_RESID && wave >= 4 ? g_stream_one : g_stream_zero;
      const float* sel = wave < 4 ? p.bias : hi;  // (wave-uniform)
      const float* src = sel && cok ? sel + col : dflt + lane;
      dma4v(src, d);
    } else {
      int row = t.m0 + (wave & 3) * 32 + (lane >> 1);
      row = row < p.M ? row : p.M - 1;
      const float* ssrc = FOLD ? p.ln_stats + 2 * (int64_t)row + (lane & 1) : g_stream_zero + lane;
      dma4v(ssrc, d + 2048);
    }
  };

  // ---- epilogue state of the previous tile -------------------------------------------------------------------------
  // acc[jt][it][4 g + e] = D[column 32 jt + 16 hh + 4 g + e][row 32 it + r31] of the wave tile (see colp above)
  f32x16 acc[2][2], prev[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) prev[j][i][e] = 0.f;
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
    const uint32_t v = (uint32_t)r31 * ldc2 + (uint32_t)hh * 32u;
    c_voff = nb < p.N ? v : 0x7fffffffu;  // (N % 64 == 0: a wave tile is in range as a whole or not at all)
  };
  epi_setup();

  // The epilogue of one quantum (Q = 8 it + 4 jt + g: 4 consecutive columns of one row per lane) as NSTG stages of four
  // independent vector instructions; state of the quantum in flight:
  f32x4 e_b, e_c;                   // bias and column sums of the lane's 4 columns
  float e_rs = 1.0f, e_nrm = 0.0f;  // rstd and -rstd * mean of its row
  float e_v[4], e_a[4], e_r[4], e_q[4];
  uint32_t e_pk[2][2];  // packed bf16 pairs of the even / odd quantum of a pair, until their 16-byte store
  auto epi_stage = [&](auto q_tag, auto j_tag) {
    constexpr int Q = decltype(q_tag)::value, J = decltype(j_tag)::value;
    constexpr int it = Q >> 3, jt = (Q >> 2) & 1, g = Q & 3;
    constexpr bool GELU = EPI == EPI_BIAS_GELU;
    constexpr int LAST = GELU ? 13 : 3;
    if constexpr (J == 0) {
      const uint32_t cst = lds0 + ST_CONST + (uint32_t)epar * 4096;
      const int colq = wn * 64 + 32 * jt + 16 * hh + 4 * g;
      e_b = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(cst + colq * 4));
      if constexpr (FOLD) {
        e_c = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(cst + 1024 + colq * 4));
        const f32x2 st = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((uintptr_t)(cst + 2048 + (wm * 64 + it * 32 + r31) * 8));
        e_rs = st[1];
        e_nrm = -st[1] * st[0];
      }
    } else if constexpr (J == 1) {
      // rs (acc - mu c) + b  =  rs acc + (b - rs mu c): the same two fused multiply-adds as epilogue_bf16 (gemm_epi.h)
#pragma unroll
      for (int e = 0; e < 4; ++e) e_v[e] = FOLD ? fmaf(e_nrm, e_c[e], e_b[e]) : e_b[e];
    } else if constexpr (J == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) e_v[e] = FOLD ? fmaf(e_rs, prev[jt][it][4 * g + e], e_v[e]) : prev[jt][it][4 * g + e] + e_v[e];
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
      e_pk[g & 1][0] = w2[0];
      e_pk[g & 1][1] = w2[1];
    }
  };
  // 16-byte store of quantum pair K2 = (it, jt, g pair): 8 consecutive columns of one row
  auto epi_store = [&](auto k_tag) {
    constexpr int K2 = decltype(k_tag)::value;
    constexpr int it = K2 >> 2, jt = (K2 >> 1) & 1, gp = K2 & 1;
    u32x4 v;
    v[0] = e_pk[0][0];
    v[1] = e_pk[0][1];
    v[2] = e_pk[1][0];
    v[3] = e_pk[1][1];
    // (row step in the VGPR offset: the hardware's range check covers voffset + immediate, rows past M are dropped; a
    // masked lane's 0x7fffffff stays out of range)
    const uint32_t voff = c_voff + (uint32_t)(it * 32) * ldc2 + (uint32_t)(jt * 64 + gp * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, voff, 0, NT ? 2 : 0);
  };

  // ---- prologue: stages 0..2 of the first tile ---------------------------------------------------------------------
  for (int st = 0; st < 3; ++st) {
    static_for<6>([&](auto j) { dma_piece(j, st); });
    dma_advance();
  }
  StTile ct = dt;  // compute tile == first tile (nk > 3: the loader has not left it)
  int cpar = 0;
  wait_vmcnt<12>();  // stage 0 landed (this wave's pieces)
  asm volatile("s_barrier" ::: "memory");

  // fragments of the four 16-deep substeps of a step: read two substeps (8 MFMAs) ahead of their use
  bf16x8 fa[4][2], fb[4][2];
  int slot = 0;
  // fragment X of substep KS of stage sl: X < 2 W tile X (the A operand of the transposed MFMA), else A row tile X - 2
  auto rd_frag = [&](auto ks_tag, auto x_tag, uint32_t stage_base) {
    constexpr int KS = decltype(ks_tag)::value, X = decltype(x_tag)::value;
    if constexpr (X < 2) fb[KS][X] = lds_rd(stage_base + b_ad[KS] + X * 4096);
    else fa[KS][X - 2] = lds_rd(stage_base + a_ad[KS] + (X - 2) * 4096);
  };
  static_for<2>([&](auto ks) { static_for<4>([&](auto x) { rd_frag(ks, x, lds0); }); });
  wait_vmcnt<6>();  // stage 1 landed: the first step reads from it before its mid-step barrier
  asm volatile("s_barrier" ::: "memory");
#ifdef VDR_STREAM_STAMPS
  // diagnostic build: s_memtime around the mid-step synchronisation of every step, summed per wave (each stamp waits for
  // its own return: ~4 x 50 cycles per step of perturbation)
  uint32_t ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev, t_a, t_b, t_c, t_d;
  const unsigned long long wall0 = wall_clock64();
#define ST_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
  ST_STAMP(t_prev);
#else
#define ST_STAMP(v)
#endif

  // One 64-deep step = 16 MFMA slots (4 substeps x 2 x 2 tiles of 32 x 32 x 16), 32 matrix-pipe cycles each.  Behind the
  // MFMA of a slot: one fragment read (two substeps ahead), in the second half one LDS-DMA piece of stage g + 3 (6 of its
  // 8 slots) and up to two epilogue stages / one store of the previous tile.  Every slot is its own scheduling region: the
  // order below is the order of the instruction stream.
  // SI: step index in the tile for the epilogue schedule (-1: none).  MODE 0: body, 1: first step of a tile
  // (accumulators start from zero), 2: last step (results go to `prev`).  NV: vmcnt of the mid-step wait.
  auto step = [&](auto si_tag, auto mode_tag, auto nv_tag) {
    constexpr int SI = decltype(si_tag)::value, MODE = decltype(mode_tag)::value, NV = decltype(nv_tag)::value;
    const int nslot = slot == 2 ? 0 : slot + 1;
    const uint32_t base_cur = lds0 + (uint32_t)slot * ST_STAGE, base_nxt = lds0 + (uint32_t)nslot * ST_STAGE;
#if defined(ST_ABL_XIDLE)
    const bool do_mma = wave >= 4;  // diagnostic: waves 0-3 only synchronise -> what waves 4-7 do alone on their SIMDs
#elif defined(ST_ABL_YIDLE)
    const bool do_mma = wave < 4;
#else
    constexpr bool do_mma = true;
#endif
    auto slot_body = [&](auto ls_tag) {
      constexpr int LS = decltype(ls_tag)::value, KS = LS >> 2, i = (LS >> 1) & 1, j = LS & 1;
      if constexpr (MODE == 1 && KS == 0) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[KS][j], fa[KS][i], z, 0, 0, 0);
      } else if constexpr (MODE == 2 && KS == 3) {
        prev[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[KS][j], fa[KS][i], acc[j][i], 0, 0, 0);
      } else {
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[KS][j], fa[KS][i], acc[j][i], 0, 0, 0);
      }
      // fragment (LS & 3) of the substep two ahead: substeps 2, 3 of this stage, then 0, 1 of the next one
#ifndef ST_ABL_NOREAD
      if constexpr (KS < 2) rd_frag(ST_I(KS + 2), ST_I(LS & 3), base_cur);
      else rd_frag(ST_I(KS - 2), ST_I(LS & 3), base_nxt);
#endif
      if constexpr (LS >= 8 && (LS & 3) != 3) dma_piece(ST_I((LS - 8) - ((LS - 8) >> 2)), slot);
      if constexpr (SI == 0 && LS == 15) {  // (both BEHIND the stage's last piece: the counted waits of steps 1 and 2 add 2)
        dma_consts(ST_I(0), ct, cpar);
        dma_consts(ST_I(1), ct, cpar);
      }
      if constexpr (SI >= 0) {
        constexpr int S = SI * 16 + LS, s0 = st_first_stage(S, EPI), s1 = st_first_stage(S + 1, EPI), sk = st_store_at(S, EPI);
        static_assert(LS < 8 || sk < 0, "stores sit in first halves only (vmcnt bookkeeping)");
        if constexpr (sk >= 0) epi_store(ST_I(sk));
        static_for<(s1 - s0)>([&](auto d) {
          constexpr int idx = s0 + decltype(d)::value;
          epi_stage(ST_I(idx / st_nstg(EPI)), ST_I(idx % st_nstg(EPI)));
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (do_mma) static_for<8>(slot_body);
    // ---- middle: this wave has read stage `slot` completely; its pieces of the next stage have landed
    ST_STAMP(t_a);  // (sampled at issue, returns behind the LDS reads: t_b - t_a = what the fragment reads still took)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ST_STAMP(t_b);
#ifdef ST_ABL_NEARWAIT
    wait_vmcnt<(NV >= 6 ? NV - 6 : 0)>();  // diagnostic: also wait for the stage issued ONE step ago (what a 2-stage ring would need)
#else
    wait_vmcnt<NV>();
#endif
    ST_STAMP(t_c);
    asm volatile("s_barrier" ::: "memory");
    ST_STAMP(t_d);
#ifdef VDR_STREAM_STAMPS
    ph[0] += (uint32_t)(t_a - t_prev);  // second half of the step before + first half of this one
    ph[1] += (uint32_t)(t_b - t_a);
    ph[2] += (uint32_t)(t_c - t_b);
    ph[3] += (uint32_t)(t_d - t_c);
    ph[4] += 1;
    t_prev = t_d;
#endif
    if (do_mma) static_for<8>([&](auto l) { slot_body(ST_I(8 + decltype(l)::value)); });
    else static_for<6>([&](auto j) { dma_piece(j, slot); });
    dma_advance();
    slot = nslot;
  };

  for (;;) {
    static_for<ST_EU>([&](auto si) {
      constexpr int SI = decltype(si)::value;
      step(si, ST_I(SI == 0 ? 1 : 0), ST_I(st_nv(SI, EPI)));
    });
    for (int s = ST_EU; s < nk - 1; ++s) step(ST_I(-1), ST_I(0), ST_I(6));
    step(ST_I(-1), ST_I(2), ST_I(6));
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
