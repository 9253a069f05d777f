// Tile variant 31: the 256 x 256 x 64 "8-phase" GEMM for the write-once linears with many tiles (attn.qkv, mlp.fc1 of the
// frozen ViTs called at src/tfds_dense_descriptor.py:123; the nn.Linear calls of nn.MultiheadAttention /
// nn.TransformerEncoderLayer, reference src/models_archs.py:130-135):  C[M, N] = epi(A[M, K] . W[N, K]^T),
// bf16 in / fp32 accumulate / bf16 out, EPI_BIAS or EPI_BIAS_GELU, optionally with the consumer-side LayerNorm fold.
//
// The structure is cdna_hip_programming.md 5, "The 256^2 8-phase template", built from its description and measured as a
// micro-benchmark first (tools/micro/gemm8p.hip, profiles/r04_gemm8p_micro.txt: bitwise equal to ring4; with its output
// leaving as non-temporal BUFFER stores behind a wave-uniform descriptor qkv 0.80 x and fc1 0.85 x of ring4's time at a
// bias-only epilogue, 1.25-1.37 PF at 4096^3):
//   * one 512-thread workgroup per CU, persistent over a static tile list; 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 of
//     the tile = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16 (128 registers);
//   * LDS: 2 K-tile buffers x {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255} x [128 rows][128 B] = 128 KB,
//     every half-tile filled by 2 LDS-DMA instructions per thread (8 rows x 128 B: whole lines per wave-instruction),
//     16-byte chunks XOR-swizzled on the source address and on the ds_read_b128 address (conflict-free fragment reads);
//     W in the plain PyTorch layout [N][K] (no packed copy);
//   * a K-tile is 4 phases of 16 MFMAs (a 64 x 32 quadrant of the wave tile x K = 64); a phase is
//         ds_read fragments | one or two half-tiles of LDS-DMA | [counted vmcnt] | lgkmcnt(0) | s_barrier |
//         s_setprio 1 | 16 MFMAs | s_setprio 0 | s_barrier
//     and waves 4-7 (the second M half: the second wave of every SIMD) run ONE barrier behind waves 0-3, so a wave's MFMA
//     segment always faces its SIMD partner's load segment (ping-pong);
//   * the DMA stream never drains and runs on ACROSS output tiles: A1 + W0 of K-tile t+1 are issued in t's 1st phase, W1
//     in its 2nd, A0 of t+2 in its 4th (a half-tile is re-staged one phase after the lgkmcnt + barrier that retired its
//     last read); the counted vmcnt(2) of the 4th phase leaves that last half-tile in flight;
//   * no epilogue phase: a 64-row half of the wave tile is final after the 2nd / 4th phase of the tile's last K-tile and
//     is finished (LayerNorm fold, bias, erf-GELU, bf16) and stored in the LOAD segments of the following phases, facing
//     the partner wave's MFMAs; the next tile's first MFMA into a quadrant starts from C = 0.  W rows are permuted inside a
//     32-column block so that a lane's registers of two neighbouring accumulator tiles are 8 consecutive columns, and lanes
//     r, r ^ 8 of a 16-lane row exchange one 16-byte block (DPP row_ror:8): a store instruction covers 8 rows x 128 B --
//     whole lines straight from the accumulator layout, no LDS staging;
//   * per-tile constants (bias, column sums of the folded weight, (mean, rstd) of the tile's 256 rows: 4 KB) arrive by
//     LDS-DMA too, double buffered by tile parity: no vector-memory load in the kernel is a plain load, so every counted
//     vmcnt written here is exact (LDS-DMA pieces and stores count together, in issue order).
// Same products in the same order and the same epilogue formula as ring3 / ring4: outputs are bitwise equal
// (tests/test_ops_gpu.py::test_linear_8phase_*).  Shapes: N % 256 == 0, K % 128 == 0, K >= 256; M % 256 == 0, or A / the row
// statistics readable up to M rounded up to 256 (GemmArgs::a_rows: the forward's workspace buffers are) -- the rows of the
// last tile past M are computed from whatever is there and never stored; enough tiles to fill the chip evenly
// (gemm_8p_eligible: ViT-g/14 at batch 32 has 594 tiles = 2.3 rounds of 256 workgroups and stays on ring4).
#include <utility>

#include "gemm_kernels.h"

namespace vdr {

namespace {

template <int... I, class F>
VDR_DEV void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
VDR_DEV void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>{}, f);
}

// 1 KB per wave-instruction, global -> LDS; uniform 64-bit base + 32-bit lane offset; LDS address in M0.  (The base is
// produced by scalar arithmetic: tools/hazard_scan.py checks that no VALU writes it within 5 wait states of the asm.)
VDR_DEV void dma16(const void* base_uniform, uint32_t lane_off, uint32_t lds_addr) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base_uniform), "{m0}"(lds_addr) : "memory");
}

struct G8 {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;    // [N] (never null: the launcher substitutes zeros)
  const float* colsum;  // FOLD: [N]
  const float* stats;   // FOLD: (mean, rstd) [M][2]
  bf16_t* C;
  int M, N, K;
  int lda, ldw, ldc;    // row strides in elements
  int tn, ntiles;
  int nt_store;
};

constexpr int BUFB = 65536;  // one K-tile buffer: A0 | A1 | W0 | W1, 16 KB each
constexpr int HALFB = 16384;
constexpr int LDS_CONST = 2 * BUFB;  // 2 x [bias 1 KB | colsum 1 KB | stats 2 KB]
constexpr int LDS_TOTAL = 2 * BUFB + 2 * 4096;

template <int EPI, bool FOLD>
__global__ __launch_bounds__(512) void gemm_8p_kernel(G8 p) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU, "write-once outputs only");
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const uint32_t lda2 = (uint32_t)p.lda * 2, ldw2 = (uint32_t)p.ldw * 2;  // bytes per operand row

  // ---- staging: lane part of the source address; rows w*8 + (lane >> 3) (+ 64 j) of a half-tile, chunk lane & 7 ----
  const int row8 = wave * 8 + (lane >> 3);
  const uint32_t offA = (uint32_t)row8 * lda2 + (uint32_t)(((lane & 7) ^ ((row8 >> 1) & 7)) << 4);
  const uint32_t offB = (uint32_t)row8 * ldw2 + (uint32_t)(((lane & 7) ^ (((row8 >> 1) & 1) | (((row8 >> 3) & 3) << 1))) << 4);

  // ---- fragment read addresses (byte offsets inside a K-tile buffer) ----
  //  A: row 16 i + r of the wave's half, chunk (4 kk + q) ^ ((row >> 1) & 7)
  //  W: accumulator tile j of the wave's 64 columns reads row slots 32 (j >> 1) + 4 (j & 1) + 8 (r >> 2) + (r & 3):
  //     lane (r, q) then holds columns 32 (j >> 1) + 8 q + 4 (j & 1) + e of output row r; chunk ^ (row bits 1, 3, 4)
  uint32_t a_rd[2], b_rd[2];
  {
    const int sA = (r >> 1) & 7;
    const int sB = ((r >> 1) & 1) | ((r >> 2) << 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      a_rd[kk] = lds0 + wr * HALFB + r * 128 + (((4 * kk + q) ^ sA) << 4);
      b_rd[kk] = lds0 + 2 * HALFB + (wc >> 1) * HALFB + (wc & 1) * 8192 + (8 * (r >> 2) + (r & 3)) * 128 + (((4 * kk + q) ^ sB) << 4);
    }
  }
  auto lds_read = [&](uint32_t addr) -> bf16x8 {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)addr);
  };

  // ---- tile list: workgroups of one XCD (id % 8) take neighbouring tiles (they share A row panels in that L2) ----
  // Row-major on purpose.  PMC (profiles/r04_pmc_traffic.txt): fc1 reads 453 MB per launch against 82 MB of operands -- every
  // XCD streams all of W (4.7 MB) once per round of 32 tiles.  Walking column groups of 3 / 4 / 6 tile columns instead
  // (an XCD's 32 tiles = 8 tile rows x 4 columns: W traffic down 9 x) measured SLOWER in the forward: qkv 1.79 -> 1.88 / 1.83
  // / 1.82 ms per step, fc1 2.50 -> 2.51 / 2.55 / 2.50 -- the L2 misses are not what the loop waits for.
  const int nwg = gridDim.x;
  const int vid = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  auto tile_bases = [&](int L, const char*& a, const char*& w, int& m0, int& n0) {
    const int tmi = L / p.tn, tni = L - tmi * p.tn;
    m0 = tmi * 256;
    n0 = tni * 256;
    a = (const char*)p.A + (size_t)m0 * lda2;
    w = (const char*)p.W + (size_t)n0 * ldw2;
  };

  f32x4 acc[8][4];
  bf16x8 fa[2][4];  // [kk][m tile of the current M half]
  bf16x8 fb[2][2];  // [kk][n tile of the current N half] (the n0 fragments are read again in a K-tile's 4th phase)

  // stage one half-tile: kind 0 A0, 1 A1, 2 W0, 3 W1
  auto stage = [&](auto kind_, const char* a, const char* w, int kt, int buf) {
    constexpr int kind = decltype(kind_)::value;
    const uint32_t ld2 = kind < 2 ? lda2 : ldw2;
    const char* src = (kind < 2 ? a : w) + (size_t)((kind & 1) * 128) * ld2 + (size_t)kt * 128;
    const uint32_t dst = lds0 + buf * BUFB + kind * HALFB + wave * 1024;
    dma16(src, kind < 2 ? offA : offB, dst);
    dma16(src + (size_t)64 * ld2, kind < 2 ? offA : offB, dst + 8192);
  };
  auto read_a = [&](auto mh_, int buf) {
    constexpr int mh = decltype(mh_)::value;
    sfor<2>([&](auto kk) { sfor<4>([&](auto i) { fa[kk][i] = lds_read(a_rd[kk] + buf * BUFB + (4 * mh + i) * 2048); }); });
  };
  auto read_b = [&](auto nh_, int buf) {
    constexpr int nh = decltype(nh_)::value;
    sfor<2>([&](auto kk) { sfor<2>([&](auto j) { fb[kk][j] = lds_read(b_rd[kk] + buf * BUFB + nh * 4096 + j * 512); }); });
  };
  auto mfma_quadrant = [&](auto mh_, auto nh_, auto zero_) {
    constexpr int mh = decltype(mh_)::value, nh = decltype(nh_)::value;
    constexpr bool zero = decltype(zero_)::value;
    __builtin_amdgcn_s_setprio(1);
    sfor<2>([&](auto kk) {
      sfor<4>([&](auto i) {
        sfor<2>([&](auto j) {
          constexpr int ii = 4 * mh + i, jj = 2 * nh + j;
          if constexpr (zero && kk == 0)
            acc[ii][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else
            acc[ii][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[ii][jj], 0, 0, 0);
        });
      });
    });
    __builtin_amdgcn_s_setprio(0);
  };

  // Epilogue of two m-tiles (i0, i0 + 1) of one M half, both n halves.  The arithmetic is epilogue_bf16's (gemm_epi.h), in
  // the accumulator layout:  v = fma(rs, acc, fma(-rs mu, csum, bias)), erf-GELU, one bf16 rounding.  A lane holds 16 B of
  // output row r in each 32-column block (p = 0, 1); lanes r and r ^ 8 exchange one block so that a store instruction
  // covers 8 rows x 128 B instead of 16 rows x 64 B.
  // (the descriptor ends after the last valid row: the rows of a ragged last tile past M are dropped by the range check)
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, (short)0, p.M * p.ldc * 2, 0x00020000);
  auto epi_tiles = [&](auto mh_, auto i0_, int m0, int n0, int par) {
    constexpr int mh = decltype(mh_)::value, i0 = decltype(i0_)::value;
    // (the lane-derived addresses are rebuilt from an opaque copy of the lane id at every site: hoisted out of the K loop by
    // hipcc they are long-lived registers of a kernel at its 256-register cap -- spilt, and a scratch reload inside the loop
    // would count in vmcnt next to the LDS-DMA pieces)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int r = lane_ & 15, q = lane_ >> 4;
    const uint32_t cb = lds0 + LDS_CONST + par * 4096;
    const uint32_t baddr = cb + (wc * 64 + 8 * q) * 4;
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
    const bool lower = (r & 8) == 0;
    const int lane_off = ((m0 + wr * 128 + 64 * mh + (r & 7)) * p.ldc + n0 + wc * 64 + (lower ? 0 : 32) + 8 * q) * 2;
    sfor<2>([&](auto di) {
      constexpr int i = i0 + di, ii = 4 * mh + i;
      float rs = 1.0f, nrm = 0.0f;
      if constexpr (FOLD) {
        const f32x2 st = *reinterpret_cast<const lds_f32x2*>((uintptr_t)(cb + 2048 + (wr * 128 + 64 * mh + 16 * i + r) * 8));
        rs = st[1];
        nrm = -st[1] * st[0];
      }
      bf16x8 ob[2];
      // (one 32-column block at a time: its 4 constant vectors are 16 live registers, both blocks' would be 32 and spill)
      sfor<2>([&](auto pb) {
        const f32x4 b0 = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(baddr + pb * 128));
        const f32x4 b1 = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(baddr + pb * 128 + 16));
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FOLD) {
          c0 = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(baddr + 1024 + pb * 128));
          c1 = *reinterpret_cast<const lds_f32x4*>((uintptr_t)(baddr + 1024 + pb * 128 + 16));
        }
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          // epilogue_bf16's formula: two fused multiply-adds (without the fold rs = 1, nrm = 0, csum = 0: acc + bias exactly),
          // on pairs of neighbouring accumulator registers: v_pk_fma_f32, bit for bit the scalar operations, two per issue slot
          const f32x2 rs2 = {rs, rs}, nrm2 = {nrm, nrm};
          f32x2 v0 = __builtin_elementwise_fma(rs2, f32x2{acc[ii][2 * pb][e], acc[ii][2 * pb][e + 1]},
                                               __builtin_elementwise_fma(nrm2, f32x2{c0[e], c0[e + 1]}, f32x2{b0[e], b0[e + 1]}));
          f32x2 v1 = __builtin_elementwise_fma(rs2, f32x2{acc[ii][2 * pb + 1][e], acc[ii][2 * pb + 1][e + 1]},
                                               __builtin_elementwise_fma(nrm2, f32x2{c1[e], c1[e + 1]}, f32x2{b1[e], b1[e + 1]}));
          if constexpr (EPI == EPI_BIAS_GELU) {
            v0 = gelu_erf2(v0);
            v1 = gelu_erf2(v1);
          }
          ob[pb][e] = (bf16_t)v0[0];
          ob[pb][e + 1] = (bf16_t)v0[1];
          ob[pb][4 + e] = (bf16_t)v1[0];
          ob[pb][4 + e + 1] = (bf16_t)v1[1];
        }
        __builtin_amdgcn_sched_barrier(0);  // (keeps hipcc from fetching the next block's constants ahead: VGPR cap)
      });
      const bf16x8 o0 = ob[0], o1 = ob[1];
      const u32x4 w0 = __builtin_bit_cast(u32x4, o0), w1 = __builtin_bit_cast(u32x4, o1);
      u32x4 da, db;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t send = lower ? w1[d] : w0[d];
        const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
        da[d] = lower ? w0[d] : recv;  // rows 0-7 of the m-tile: own block 0 | row r-8's block 1
        db[d] = lower ? recv : w1[d];  // rows 8-15:             row r+8's block 0 | own block 1
      }
      const int o_a = lane_off + (16 * i) * p.ldc * 2, o_b = lane_off + (16 * i + 8) * p.ldc * 2;
      if (p.nt_store) {
        __builtin_amdgcn_raw_buffer_store_b128(da, crs, o_a, 0, 2 /* nt */);
        __builtin_amdgcn_raw_buffer_store_b128(db, crs, o_b, 0, 2);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(da, crs, o_a, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(db, crs, o_b, 0, 0);
      }
    });
  };
  auto seg_sync_a = [&]() {  // end of a load segment: own LDS reads retired, then the rendezvous
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto seg_sync_b = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  int L = vid;
  if (L >= p.ntiles) return;
  const char *ca, *cw, *na, *nw_;
  int m0, n0, nm0, nn0;
  tile_bases(L, ca, cw, m0, n0);
  const int nk = p.K >> 6;
  int par = 0;
  // constants of a tile: one 1-KB piece each by waves 0 .. 3 (bias, column sums, two halves of the row statistics)
  auto stage_consts = [&](int m0_, int n0_, int par_) {
    const uint32_t cb = lds0 + LDS_CONST + par_ * 4096;
    if (wave == 0) dma16((const char*)p.bias + (size_t)n0_ * 4, (uint32_t)lane * 16, cb);
    if constexpr (FOLD) {
      if (wave == 1) dma16((const char*)p.colsum + (size_t)n0_ * 4, (uint32_t)lane * 16, cb + 1024);
      if (wave == 2) dma16((const char*)p.stats + (size_t)m0_ * 8, (uint32_t)lane * 16, cb + 2048);
      if (wave == 3) dma16((const char*)p.stats + (size_t)(m0_ + 128) * 8, (uint32_t)lane * 16, cb + 3072);
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using T = std::true_type;
  using F = std::false_type;

  // ---- prologue: K-tile 0 complete, A0 of K-tile 1 in flight ----
  stage(I0{}, ca, cw, 0, 0);
  stage(I1{}, ca, cw, 0, 0);
  stage(I2{}, ca, cw, 0, 0);
  stage(I3{}, ca, cw, 0, 0);
  stage(I0{}, ca, cw, 1, 1);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the second M half runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  bool have_prev = false;
  int pm0 = 0, pn0 = 0;

  // one K-tile pair (K-tiles t0 in buffer 0, t0 + 1 in buffer 1); FIRST: the tile's first pair (C = 0, the previous
  // tile's second M half leaves in phases 2 and 3); LAST: the tile's last pair (the first M half leaves in phases 7 and 8,
  // the DMA stream moves on to the next tile: K-tiles t0 + 2, t0 + 3 are its K-tiles 0, 1)
  auto pair = [&](auto first_, auto last_, int t0) {
    constexpr bool FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
    const char* a2 = LAST ? na : ca;
    const char* w2 = LAST ? nw_ : cw;
    const int k2 = LAST ? 0 : t0 + 2, k3 = LAST ? 1 : t0 + 3;
    // Where a finished M half leaves.  Waves 0-3 ("X") do it at the head of a load segment, waves 4-7 ("Y", one barrier
    // behind) at the tail of the MFMA segment that faces it: the SAME slot of wall-clock time, so the two waves of a SIMD
    // run their epilogues side by side (a wave issues one vector instruction per ~5.4 cycles, a SIMD takes one per ~2.7:
    // profiles/r03_valu_rate_micro.txt) instead of one after the other with the partner's 16 MFMAs long finished --
    // measured with every wave in its load segments: the erf-GELU cost 35 us of an fc1 launch, tools/ab_epi.py.
    const bool X = wr == 0;
    // ---- K-tile t0, buffer 0 ----
    // phase 1: quadrant (m0, n0)
    read_b(I0{}, 0);
    read_a(I0{}, 0);
    if constexpr (FIRST) stage_consts(m0, n0, par);
    stage(I1{}, ca, cw, t0 + 1, 1);  // A1 and W0 of t0 + 1 (buffer 1's W halves were last read in the previous pair's phase 8)
    stage(I2{}, ca, cw, t0 + 1, 1);
    // (a tile's first pair stages W1 here too -- its half of buffer 1 was last read in the previous pair's phase 6 -- so that
    // the stores of the previous tile's second M half, phases 1 to 3, are all BEHIND this K-tile's last piece: the counted
    // wait of phase 4 never waits for the acknowledgement of a store)
    if constexpr (FIRST) stage(I3{}, ca, cw, t0 + 1, 1);
    seg_sync_a();
    mfma_quadrant(I0{}, I0{}, std::integral_constant<bool, FIRST>{});
    if constexpr (FIRST) {
      if (have_prev && !X) {
        __builtin_amdgcn_sched_barrier(0);
        epi_tiles(I1{}, I0{}, pm0, pn0, par ^ 1);
      }
    }
    seg_sync_b();
    // phase 2: (m0, n1)
    read_b(I1{}, 0);
    if constexpr (!FIRST) stage(I3{}, ca, cw, t0 + 1, 1);  // W1 of t0 + 1
    // the previous tile's second M half (final after its phase 8) leaves in phases 2 and 3
    if constexpr (FIRST) {
      if (have_prev && X) epi_tiles(I1{}, I0{}, pm0, pn0, par ^ 1);
    }
    seg_sync_a();
    mfma_quadrant(I0{}, I1{}, std::integral_constant<bool, FIRST>{});
    if constexpr (FIRST) {
      if (have_prev && !X) {
        __builtin_amdgcn_sched_barrier(0);
        epi_tiles(I1{}, I2{}, pm0, pn0, par ^ 1);
      }
    }
    seg_sync_b();
    // phase 3: (m1, n1)
    read_a(I1{}, 0);
    if constexpr (FIRST) {
      if (have_prev && X) epi_tiles(I1{}, I2{}, pm0, pn0, par ^ 1);
    }
    seg_sync_a();
    mfma_quadrant(I1{}, I1{}, std::integral_constant<bool, FIRST>{});
    seg_sync_b();
    // phase 4: (m1, n0)
    read_b(I0{}, 0);
    stage(I0{}, a2, w2, k2, 0);  // A0 of t0 + 2 (buffer 0's W halves are read until this phase: the n0 fragments again)
    // K-tile t0 + 1 has landed (this wave's pieces; the barrier does the rest); younger: A0 of t0 + 2 and the 8 stores
    if (FIRST && have_prev) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    seg_sync_a();
    mfma_quadrant(I1{}, I0{}, std::integral_constant<bool, FIRST>{});
    seg_sync_b();
    // ---- K-tile t0 + 1, buffer 1 ----
    // phase 5: (m0, n0)
    read_b(I0{}, 1);
    read_a(I0{}, 1);
    stage(I1{}, a2, w2, k2, 0);  // A1 and W0 of t0 + 2
    stage(I2{}, a2, w2, k2, 0);
    seg_sync_a();
    mfma_quadrant(I0{}, I0{}, F{});
    seg_sync_b();
    // phase 6: (m0, n1)
    read_b(I1{}, 1);
    stage(I3{}, a2, w2, k2, 0);  // W1 of t0 + 2
    seg_sync_a();
    mfma_quadrant(I0{}, I1{}, F{});
    if constexpr (LAST) {
      if (!X) {  // first M half (final with this segment), m-tiles 0, 1
        __builtin_amdgcn_sched_barrier(0);
        epi_tiles(I0{}, I0{}, m0, n0, par);
      }
    }
    seg_sync_b();
    // phase 7: (m1, n1)
    read_a(I1{}, 1);
    if constexpr (LAST) {
      if (X) epi_tiles(I0{}, I0{}, m0, n0, par);
    }
    seg_sync_a();
    mfma_quadrant(I1{}, I1{}, F{});
    if constexpr (LAST) {
      if (!X) {  // ... m-tiles 2, 3
        __builtin_amdgcn_sched_barrier(0);
        epi_tiles(I0{}, I2{}, m0, n0, par);
      }
    }
    seg_sync_b();
    // phase 8: (m1, n0)
    read_b(I0{}, 1);
    stage(I0{}, a2, w2, k3, 1);  // A0 of t0 + 3
    // K-tile t0 + 2 has landed: younger than its last piece are A0 of t0 + 3 (2) and, in a LAST pair, the stores issued since
    // phase 6's staging (X: the 4 of phase 7; Y: all 8)
    if constexpr (LAST) {
      if (X) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    if constexpr (LAST) {
      if (X) epi_tiles(I0{}, I2{}, m0, n0, par);
    }
    seg_sync_a();
    mfma_quadrant(I1{}, I0{}, F{});
    seg_sync_b();
  };

  for (;;) {
    const int Ln = L + nwg;
    const bool more = Ln < p.ntiles;
    // (no next tile: the stream re-loads this tile's first K-tiles into buffers nobody reads again)
    tile_bases(more ? Ln : L, na, nw_, nm0, nn0);
    pair(T{}, F{}, 0);
    for (int t0 = 2; t0 < nk - 2; t0 += 2) pair(F{}, F{}, t0);
    pair(F{}, T{}, nk - 2);
    have_prev = true;
    pm0 = m0;
    pn0 = n0;
    par ^= 1;
    if (!more) break;
    L = Ln;
    ca = na;
    cw = nw_;
    m0 = nm0;
    n0 = nn0;
  }
  // the last tile's second M half; then the first M half waits for the second's extra barrier
  epi_tiles(I1{}, I0{}, pm0, pn0, par ^ 1);
  epi_tiles(I1{}, I2{}, pm0, pn0, par ^ 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wr == 0) __builtin_amdgcn_s_barrier();
}

}  // namespace

// whether tile variant 31 takes this launch (and is expected to pay: many tiles, a write-once bf16 output)
// the tile-count rule alone: one workgroup per CU walks tiles / CUs rounds -- at least two, and the last one reasonably full
// (a 2.3-round launch runs 3 rounds long: measured on ViT-g/14's qkv, 594 tiles: 114 us against ring4's 107; a launch of
// less than one round -- one MedSAM slice, M = 4096: 144 / 192 tiles -- pays the pipeline fill and the four epilogue slots once
// per 12 K-tiles: qkv 27.1 -> 28.9 us, fc1 31.4 -> 32.4 against the 128-row ring4 tiles)
bool gemm_8p_shape_ok(int64_t M, int N) {
  if (M <= 0 || (N & 255)) return false;
  int n_cu = device_cu_count(current_device_index());
  if (n_cu <= 0) n_cu = 256;
  const int64_t tiles = ((M + 255) / 256) * (int64_t)(N / 256), slots = n_cu & ~7;
  if (tiles < 2 * slots) return false;
  const int64_t rounds = (tiles + slots - 1) / slots;
  return (double)tiles >= 0.85 * (double)(rounds * slots);
}

bool gemm_8p_eligible(const GemmArgs& a, int epi) {
  if (epi != EPI_BIAS && epi != EPI_BIAS_GELU) return false;
  if (a.out_f32 || a.win_ws || a.a_rpg || a.patch_p || a.ln_part || a.ln_cpart || a.w_interleaved || a.resid32 || a.C32) return false;
  if (a.omap.rpg < (1 << 30) || a.a_scale || a.w_scale || a.c_scale) return false;
  if (a.M <= 0 || (a.N & 255) || (a.K & 127) || a.K < 256) return false;
  const int64_t Mr = (a.M + 255) & ~(int64_t)255;
  if (Mr != a.M && a.a_rows < Mr) return false;  // a ragged last tile reads 256 rows of A (and of the row statistics)
  if (a.lda < a.K || a.ldw < a.K || a.ldc < a.N) return false;
  if ((a.ln_stats != nullptr) != (a.colsum != nullptr)) return false;
  if ((double)Mr * a.ldc * 2.0 >= 2147483648.0 || (double)Mr * a.lda * 2.0 >= 4294967296.0 || (double)a.N * a.ldw * 2.0 >= 4294967296.0)
    return false;  // 32-bit store offsets / lane offsets
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15) return false;
  if ((a.lda | a.ldw | a.ldc) & 7) return false;
  return gemm_8p_shape_ok(a.M, a.N);
}

hipError_t launch_gemm_8p(const GemmArgs& a, int epi, hipStream_t s) {
  if (!gemm_8p_eligible(a, epi)) return hipErrorInvalidValue;
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  static float* zeros[VDR_MAX_DEVICES] = {};  // a zero bias of the largest N seen (bias == NULL), per device
  static int zeros_n[VDR_MAX_DEVICES] = {};
  const float* bias = a.bias;
  if (!bias) {
    if (zeros_n[dev] < a.N) {
      // (load-time path in practice: every Linear of the models has a bias; kept correct for vdr_op_linear callers)
      if (zeros[dev]) (void)hipFree(zeros[dev]);
      hipError_t e = hipMalloc(&zeros[dev], (size_t)a.N * 4);
      if (e != hipSuccess) return e;
      e = hipMemset(zeros[dev], 0, (size_t)a.N * 4);
      if (e != hipSuccess) return e;
      zeros_n[dev] = a.N;
    }
    bias = zeros[dev];
  }
  G8 g{(const bf16_t*)a.A, (const bf16_t*)a.W, bias, a.colsum, a.ln_stats, (bf16_t*)a.C, (int)a.M, a.N, a.K,
       (int)a.lda, (int)a.ldw, (int)a.ldc, a.N / 256, (int)(((a.M + 255) / 256) * (a.N / 256)), 0};
  g.nt_store = (double)a.M * (double)a.ldc * 2.0 >= 128e6 ? 1 : 0;  // as launch_cfg: outputs larger than half the Infinity Cache
  int n_cu = device_cu_count(dev);
  if (n_cu <= 0) n_cu = 256;
  const int grid = n_cu & ~7;
  const bool fold = a.ln_stats != nullptr;
#define VDR_L8(E, FO)                                                                                              \
  {                                                                                                                \
    auto fn = gemm_8p_kernel<E, FO>;                                                                               \
    static PerDeviceFlag attr;                                                                                     \
    if (!attr.done[dev]) {                                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);  \
      if (e != hipSuccess) return e;                                                                               \
      attr.done[dev] = true;                                                                                       \
    }                                                                                                              \
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), LDS_TOTAL, s, g);                                                \
  }
  if (epi == EPI_BIAS) {
    if (fold) VDR_L8(EPI_BIAS, true) else VDR_L8(EPI_BIAS, false)
  } else {
    if (fold) VDR_L8(EPI_BIAS_GELU, true) else VDR_L8(EPI_BIAS_GELU, false)
  }
#undef VDR_L8
  return hipGetLastError();
}

}  // namespace vdr
