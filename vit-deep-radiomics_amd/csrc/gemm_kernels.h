// bf16 MFMA GEMM kernels with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)
// (kernel templates + the launch helper; instantiated per tile variant by gemm.hip and gemm_ring4.hip so the library
// builds in parallel)
//
// Replaces the nn.Linear calls under nn.MultiheadAttention / nn.TransformerEncoderLayer
// (reference src/models_archs.py:130-135) and attn.qkv / attn.proj / mlp.fc1 / mlp.fc2 of the
// frozen ViTs called at src/tfds_dense_descriptor.py:123, plus the patchify conv as a GEMM
// (src/tfds_dense_descriptor.py:128).
//
// Common structure: A and W tiles go global -> LDS with 16-byte global_load_lds into a ring of LDS slots (counted
// s_waitcnt vmcnt, raw s_barrier: the loads stay in flight across barriers); the 16-byte chunks of a row are
// XOR-swizzled on the per-lane SOURCE address and on the ds_read_b128 address, which makes the fragment reads
// bank-conflict free.  The MFMA (v_mfma_f32_16x16x32_bf16) is issued transposed (W fragment as the A operand,
// activation fragment as the B operand) so that a lane owns one output ROW and 4 consecutive output columns per
// register group; the epilogue goes through LDS (epilogue_lds) so that all its global traffic is whole 128-B lines.
//
// Weight layouts (GemmArgs::w_interleaved):
//   0: PyTorch Linear layout [N][K], K contiguous.  A 32-deep unit of 16 rows is 16 x 64 B = HALF cache lines.
//   1: "pair-interleaved": [N/2][K/32][2][32] -- one 128-B line holds the 32-deep K block kb of rows 2i and 2i+1.
//      The same wave-instruction (16 rows x 64 B) now touches 8 WHOLE lines.  Measured (tools/micro/glds_shape.hip,
//      L2-resident source, 2 workgroups x 8 waves per CU): 66 GB/s per CU for half lines, 112 GB/s for whole lines.
//      Weights are packed once at load (w_interleave_kernel), so this costs nothing per forward.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "gemm_epi.h"

namespace vdr {

// 16-B chunk swizzle of an LDS image with 64-B rows (one 32-deep unit: 4 chunks per row), for the 16x16x32 MFMA fragment
// read, where lane (r = lane & 15, q = lane >> 4) takes chunk q of row r.  A ds_read_b128 is served in four groups of 16
// lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS) and four consecutive 64-B rows fill
// the 256-B bank window, so within a group rows r, r+4, r+8, r+12 of one residue class must land on four different
// chunks although two of them read chunk q and two chunk q+1.  chunk ^ (row >> 2) & 3 does NOT do that (rows 0 and 4
// collide: measured SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE in ring3, every fragment read 2-way);
// chunk ^ g[(row >> 2) & 3] with g = {0, 3, 2, 1} = -(row >> 2) & 3 does ({g0, g3, 1^g1, 1^g2} and {g1, g2, 1^g0, 1^g3}
// are both {0, 1, 2, 3}).  Same involution on the global_load_lds source address and on the read address.
VDR_DEV int swz64(int row) { return (-(row >> 2)) & 3; }

// source address of (row gr, logical 16-B chunk c) of the FIRST 32-deep unit of W
VDR_DEV const bf16_t* w_unit_src(const GemmK& p, int gr, int c) {
  return p.w_il ? p.W + (int64_t)(gr >> 1) * ((int64_t)(p.K >> 5) * 64) + (gr & 1) * 32 + c * 8
                : p.W + (int64_t)gr * p.ldw + c * 8;
}

// -------------------------------------------------------------------------------------------------
// Ring variant 3: round 1's ring pipeline (32-deep units of both operands, 32x32x16 MFMA) on the 16x16x32 MFMA shape.  Same LDS image, same bytes read per unit
// (8 ds_read_b128 per wave), same MFMA cycles (16 instructions of 4 passes instead of 8 of 8) -- but the chip
// holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back item 7): measured here, same box,
// alternating processes, -5 ... -10 % on every GEMM of the forward.
// A wave owns a 64 x 64 tile = 4 x 4 MFMA tiles; one MFMA consumes the unit's whole K = 32, so a step is split
// by A row tiles instead of k-steps:
//   top of step s :  B[0..3] (weight tiles) and Alo (activation row tiles 0, 1) of unit s are in registers
//       read Ahi <- unit s                      | 8 MFMAs  B[jt] x Alo
//       lgkmcnt(0); vmcnt: retire unit s+1; s_barrier; global_load_lds unit s+NST -> slot of unit s
//       read Alo <- unit s+1                    | 8 MFMAs  B[jt] x Ahi, each B[jt] re-read from unit s+1 as
//                                                 soon as its last MFMA of step s has been issued
// -------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int NST, int EPI>
VDR_DEV void gemm_ring3_body(const GemmK& p, const int64_t m0, const int n0, char* smem) {
  // the wave tile is 64 x 64 = 4 x 4 MFMA tiles
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * 64;
  constexpr int BN = WAVES_N * 64;
  constexpr int UNIT = (BM + BN) * 64;
  constexpr int NA = BM / 16 / NW;
  constexpr int NB = BN / 16 / NW;
  constexpr int G = NA + NB;
  static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile/wave mismatch");
  static_assert((NST - 1) * G <= 63, "vmcnt range");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int srow = lane >> 2;
  const int spc = lane & 3;
  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int r = (wave * NA + q) * 16 + srow;
    const int c = spc ^ swz64(r);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    const int64_t aoff = p.a_rpg > 0 ? (gr / p.a_rpg) * p.a_gs + (gr % p.a_rpg) * p.a_is : gr * p.lda;
    a_src[q] = p.A + aoff + c * 8;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 16 + srow;
    const int c = spc ^ swz64(r);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    b_src[q] = w_unit_src(p, gr, c);
  }
  const int bstep = p.w_il ? 64 : 32;  // elements between consecutive 32-deep units of one W row

  // fragment of MFMA tile t (16 rows): lane (r = lane & 15, q = lane >> 4) reads the 16-B chunk q of row r; the
  // chunk swizzle swz64(row) of the LDS image; tile bases are multiples of 16 rows
  const int r15 = lane & 15;
  const int chq = ((lane >> 4) ^ swz64(lane)) * 16;  // rows of a 16-row tile: (row >> 2) & 3 == (lane >> 2) & 3
  const int a_base = (wm * 64 + r15) * 64 + chq;
  const int b_base = BM * 64 + (wn * 64 + r15) * 64 + chq;

  Acc16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.t[j][i][e] = 0.0f;

  const int nsteps = p.K >> 5;
  const bool do_epi = !VDR_ABL(p, 1);
  const bool skip_loads = VDR_ABL(p, 4);  // diagnostic (tuning builds): the ring is filled once, then reused
  auto stage = [&](int slot) {
    char* d = smem + slot * UNIT;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      glds16(a_src[q], d + (wave * NA + q) * 1024);
      a_src[q] += 32;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      glds16(b_src[q], d + BM * 64 + (wave * NB + q) * 1024);
      b_src[q] += bstep;
    }
  };
  auto ld = [&](int slot, int off) { return *reinterpret_cast<const bf16x8*>(smem + slot * UNIT + off); };
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
  bf16x8 fb[4], alo[2], ahi[2];

  // LayerNorm fold, statistics in the GEMM: the first BM threads fetch the (sum, sumsq) partials of one row each
  // BEFORE the ring fill is issued (so the counted vmcnt waits of the ring see them as older operations) and reduce
  // them after it, while the ring's first units are in flight
  constexpr int MAXG = 16;
  float2 pv[MAXG];
  const bool fold_here = __builtin_amdgcn_readfirstlane(p.ln_cpart != nullptr) && tid < BM;
  if (fold_here) {
    int64_t r = m0 + tid;
    r = r < p.M ? r : p.M - 1;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
      const int gg = g < p.ln_groups ? g : p.ln_groups - 1;
      pv[g] = *reinterpret_cast<const float2*>(p.ln_cpart + ((int64_t)gg * p.ln_cstride + r) * 2);
    }
  }

  int issued = -1;
#pragma unroll
  for (int u = 0; u < NST; ++u)
    if (u < nsteps) {
      stage(u);
      issued = u;
    }
  if (fold_here) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < p.ln_groups) {
        s1 += (double)pv[g].x;
        s2 += (double)pv[g].y;
      }
    const double mean = s1 * (double)p.ln_inv_d;
    double var = s2 * (double)p.ln_inv_d - mean * mean;
    var = var > 0.0 ? var : 0.0;
    float2 o;
    o.x = (float)mean;
    o.y = (float)(1.0 / sqrt(var + (double)p.ln_eps));
    reinterpret_cast<float2*>(smem + p.stats_off)[tid] = o;
  }
  retire(0, issued);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = ld(0, b_base + j * 1024);
#pragma unroll
  for (int i = 0; i < 2; ++i) alo[i] = ld(0, a_base + i * 1024);
  int slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int nslot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc.t[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], alo[i], acc.t[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) ahi[i] = ld(slot, a_base + (2 + i) * 1024);  // lands under the MFMAs just issued
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave holds every fragment of unit s it still needs
    if (s + 1 < nsteps) {
      retire(s + 1, issued);
      __builtin_amdgcn_s_barrier();
      if (s + NST < nsteps && !skip_loads) {
        stage(slot);
        issued = s + NST;
      }
    }
    // unit s+1 (after the last unit these reads fetch stale, unused bytes: an unconditional read costs nothing,
    // a conditional one a second register set)
#pragma unroll
    for (int i = 0; i < 2; ++i) alo[i] = ld(nslot, a_base + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc.t[j][2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], ahi[i], acc.t[j][2 + i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fb[j] = ld(nslot, b_base + j * 1024);  // B[j] of unit s is dead: fetch unit s+1's under the remaining MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = nslot;
  }

  if (!do_epi && acc.t[0][0][0] != 12345.678f) return;  // ablation: no epilogue (keeps acc live)
  __syncthreads();  // every wave is done with the ring: its memory becomes the staging area
  epilogue_tile<EPI>(p, acc, smem, wave, m0 + wm * 64, n0 + wn * 64, lane,
                     p.ln_cpart ? reinterpret_cast<const float2*>(smem + p.stats_off) + wm * 64 : nullptr);
}

// -------------------------------------------------------------------------------------------------
// ring3 with the K loop split across two wave groups of one workgroup (small problems: a single 1024^2 SAM slice is
// M = 4096, i.e. fewer 128 x 128 tiles than CUs, every tile a serial chain of K/32 units on ONE wave per SIMD).
// 8 waves per 128 x 128 tile: group g = wave >> 2 consumes unit 2s + g of "super-unit" s (64 K), so the serial chain
// halves and every SIMD holds two waves whose MFMA and LDS phases interleave; group 1 hands its accumulators to group
// 0 through LDS at the end (deterministic: fixed order, no atomics), group 0 runs the usual LDS-staged epilogue.
// -------------------------------------------------------------------------------------------------
template <int NST, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_ring3k_kernel(GemmK p) {
  constexpr int TM = 2, TN = 2;
  constexpr int BM = 128, BN = 128;
  constexpr int UNIT = (BM + BN) * 64;  // 16 KB: one 32-K unit
  constexpr int SUPER = 2 * UNIT;       // one 64-K super-unit: unit of group 0, then unit of group 1
  constexpr int G = 4;                  // DMA instructions per wave and super-unit
  static_assert((NST - 1) * G <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wg = xcd_remap(blockIdx.x, p.nwg);
  int tm, tn;
  tile_of(p, wg, tm, tn);
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;

  // staging: wave w brings 16 rows of A and 16 rows of W of BOTH units of a super-unit
  const int srow = lane >> 2, spc = lane & 3;
  const bf16_t* a_src;
  const bf16_t* b_src;
  {
    const int r = wave * 16 + srow;
    const int c = spc ^ swz64(r);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    const int64_t aoff = p.a_rpg > 0 ? (gr / p.a_rpg) * p.a_gs + (gr % p.a_rpg) * p.a_is : gr * p.lda;
    a_src = p.A + aoff + c * 8;
    int gn = n0 + r;
    gn = gn < p.N ? gn : p.N - 1;
    b_src = w_unit_src(p, gn, c);
  }
  const int bstep = p.w_il ? 64 : 32;
  const int r15 = lane & 15;
  const int chq = ((lane >> 4) ^ swz64(lane)) * 16;  // rows of a 16-row tile: (row >> 2) & 3 == (lane >> 2) & 3
  const int a_base = kg * UNIT + (wm * 64 + r15) * 64 + chq;
  const int b_base = kg * UNIT + BM * 64 + (wn * 64 + r15) * 64 + chq;

  Acc16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.t[j][i][e] = 0.0f;

  const int nsteps = p.K >> 6;  // super-units
  auto stage = [&](int slot) {
    char* d = smem + slot * SUPER;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      glds16(a_src + g * 32, d + g * UNIT + wave * 1024);
      glds16(b_src + g * bstep, d + g * UNIT + BM * 64 + wave * 1024);
    }
    a_src += 64;
    b_src += 2 * bstep;
  };
  auto ld = [&](int slot, int off) { return *reinterpret_cast<const bf16x8*>(smem + slot * SUPER + off); };
  auto retire = [&](int u, int issued_upto) {
    const int younger = issued_upto - u;
    if (younger >= NST - 1) {
      wait_vmcnt<(NST - 1) * G>();
    } else if (younger == NST - 2 && NST >= 3) {
      wait_vmcnt<(NST - 2) * G>();
    } else {
      wait_vmcnt<0>();
    }
  };
  bf16x8 fb[4], alo[2], ahi[2];
  int issued = -1;
#pragma unroll
  for (int u = 0; u < NST; ++u)
    if (u < nsteps) {
      stage(u);
      issued = u;
    }
  retire(0, issued);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = ld(0, b_base + j * 1024);
#pragma unroll
  for (int i = 0; i < 2; ++i) alo[i] = ld(0, a_base + i * 1024);
  int slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int nslot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc.t[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], alo[i], acc.t[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) ahi[i] = ld(slot, a_base + (2 + i) * 1024);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (s + 1 < nsteps) {
      retire(s + 1, issued);
      __builtin_amdgcn_s_barrier();
      if (s + NST < nsteps) {
        stage(slot);
        issued = s + NST;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) alo[i] = ld(nslot, a_base + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc.t[j][2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], ahi[i], acc.t[j][2 + i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fb[j] = ld(nslot, b_base + j * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = nslot;
  }

  // K reduction across the two groups: group 1 -> LDS -> group 0 (layout [register][lane] f32x4: conflict-free)
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem) + w4 * (16 * 64);
  if (kg == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(j * 4 + i) * 64 + lane] = acc.t[j][i];
  }
  __syncthreads();
  if (kg == 1) return;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 o = red[(j * 4 + i) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.t[j][i][e] += o[e];
    }
  epilogue_lds<EPI, TM, TN>(p, acc, smem + 65536 + w4 * (32 * 272), m0 + wm * 64, n0 + wn * 64, lane);
}

template <int WAVES_M, int WAVES_N, int NST, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 4) void gemm_ring3_kernel(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wg = xcd_remap(blockIdx.x, p.nwg);
  int tm, tn;
  tile_of(p, wg, tm, tn);
  gemm_ring3_body<WAVES_M, WAVES_N, NST, EPI>(p, (int64_t)tm * (WAVES_M * 64), tn * (WAVES_N * 64), smem);
}


// -------------------------------------------------------------------------------------------------
// Ring variant 4: ring3's compute schedule with BOTH operands staged in whole 128-B lines.
//   W: pair-interleaved layout (above), 32-deep units of BN rows x 64 B, NSTW = 3 slots (as ring3).
//   A: row-major activations, staged as 64-deep PIECES of BM rows x 128 B (8 rows x 128 B per wave-instruction),
//      2 slots; piece q serves units 2q (bytes 0..63 of every row) and 2q+1 (bytes 64..127).  LDS image [row][128 B],
//      16-B chunk index XOR (row >> 1) & 7: conflict-free for the 16x16x32 fragment read (lane = (row & 15, chunk q)).
// LDS for a 128 x 256 tile: 2 x 16 KB + 3 x 16 KB = 80 KB = exactly half a CU, so two workgroups stay resident.
// Issue schedule (after the mid-step barrier of step s): W unit s+3 every step, A piece (s+3)/2 on odd steps; the
// counted vmcnt of step s leaves exactly what step s-1 issued in flight (NB, plus NA4 when s is even).
// -------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int EPI, bool OPAQUE_TID = false>
VDR_DEV void gemm_ring4_body(const GemmK& p, const int64_t m0, const int n0, char* smem) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * 64;
  constexpr int BN = WAVES_N * 64;
  constexpr int APIECE = BM * 128;   // bytes of one 64-deep A piece
  constexpr int WUNIT = BN * 64;     // bytes of one 32-deep W unit
  constexpr int WBASE = 2 * APIECE;  // the W ring sits behind the two A slots
  constexpr int NA4 = BM / 8 / NW;   // global_load_lds per wave per A piece (8 rows x 128 B each)
  constexpr int NB = BN / 16 / NW;   // per wave per W unit (16 rows x 64 B each = 8 row pairs x 128 B)
  static_assert(BM % (8 * NW) == 0 && BN % (16 * NW) == 0, "tile/wave mismatch");

  int tid = threadIdx.x;
  if (OPAQUE_TID) asm volatile("" : "+v"(tid));  // (persistent form: keeps the lane-derived addresses from being hoisted out of the tile loop)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
#ifdef VDR_GEMM_STAMPS
  unsigned long long st[8];
  st[0] = __builtin_readcyclecounter();
  st[6] = wall_clock64();
#define VDR_GSTAMP(i) st[i] = __builtin_readcyclecounter()
#else
#define VDR_GSTAMP(i)
#endif

  const bf16_t* a_src[NA4];
  const bf16_t* b_src[NB];
  // im2col-free patchify (EPI_PATCH, p.pg_ps != 0): row = token (b, py, px), the 16-byte chunk c of 64-deep piece 0 is
  // pixels kx0 .. kx0+7 of patch row ky = 8c / p of channel 0; piece pc adds a wave-uniform offset (see stage_a)
  const bool patch_gather = EPI == EPI_PATCH && __builtin_amdgcn_readfirstlane(p.pg_ps) != 0;
#pragma unroll
  for (int q = 0; q < NA4; ++q) {
    const int r = (wave * NA4 + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    if (patch_gather) {
      const int P = 1 << p.pg_ps, G = p.pg_g, side = G << p.pg_ps;
      const int b = (int)(gr / (G * G));
      const int t = (int)(gr - (int64_t)b * (G * G));
      const int py = t / G, px = t - py * G;
      const int ky = (c * 8) >> p.pg_ps, kx0 = (c * 8) & (P - 1);
      a_src[q] = p.A + ((int64_t)b * p.pg_C * side + py * P + ky) * side + px * P + kx0;
    } else {
      a_src[q] = p.A + gr * p.lda + c * 8;
    }
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz64(r);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    b_src[q] = w_unit_src(p, gr, c);
  }
  const int bstep = p.w_il ? 64 : 32;

  // fragment of MFMA tile t (16 rows): lane (r = lane & 15, q = lane >> 4) reads the 16-B chunk q of row r of the
  // current 32-deep half
  const int r15 = lane & 15, q4 = lane >> 4;
  const int a_row = (wm * 64 + r15) * 128;
  const int a_ch0 = a_row + ((q4 ^ ((r15 >> 1) & 7)) * 16);        // even units: chunks 0..3 of the 128-B row
  const int a_ch1 = a_row + (((4 + q4) ^ ((r15 >> 1) & 7)) * 16);  // odd units: chunks 4..7
  const int b_base = WBASE + (wn * 64 + r15) * 64 + ((q4 ^ swz64(lane)) * 16);

  Acc16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.t[j][i][e] = 0.0f;

  const int nsteps = p.K >> 5;  // even: K % 64 == 0
  const bool do_epi = !VDR_ABL(p, 1);
  const bool skip_loads = VDR_ABL(p, 4);
  int a_piece = 0;  // (patch gather) 64-deep piece the next stage_a fetches
  auto stage_a = [&](int aslot) {
    char* d = smem + aslot * APIECE;
    if (patch_gather) {
      // k = 64 a_piece .. + 63: channel k / p^2, first patch row (k % p^2) / p -- the same for every lane (p^2 % 64 == 0)
      const int kq = a_piece << 6, side = p.pg_g << p.pg_ps;
      const int ch = kq >> (2 * p.pg_ps), kyb = (kq & ((1 << (2 * p.pg_ps)) - 1)) >> p.pg_ps;
      const int64_t off = ((int64_t)ch * side + kyb) * side;
#pragma unroll
      for (int q = 0; q < NA4; ++q) glds16(a_src[q] + off, d + (wave * NA4 + q) * 1024);
      ++a_piece;
      return;
    }
#pragma unroll
    for (int q = 0; q < NA4; ++q) {
      glds16(a_src[q], d + (wave * NA4 + q) * 1024);
      a_src[q] += 64;
    }
  };
  auto stage_w = [&](int wslot) {
    char* d = smem + WBASE + wslot * WUNIT;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      glds16(b_src[q], d + (wave * NB + q) * 1024);
      b_src[q] += bstep;
    }
  };
  auto ld = [&](int off) { return *reinterpret_cast<const bf16x8*>(smem + off); };
  bf16x8 fb[4], alo[2], ahi[2];

  // LayerNorm fold, statistics in the GEMM (see gemm_ring3_body): the (mean, rstd) of row tid stays in two registers
  // through the main loop and goes to LDS (behind the epilogue staging area) once the ring is dead
  constexpr int MAXG = 16;
  float2 pv[MAXG];
  float2 my_stats = {0.0f, 1.0f};
  const bool fold_here = __builtin_amdgcn_readfirstlane(p.ln_cpart != nullptr) && tid < BM;
  if (fold_here) {
    int64_t r = m0 + tid;
    r = r < p.M ? r : p.M - 1;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
      const int gg = g < p.ln_groups ? g : p.ln_groups - 1;
      pv[g] = *reinterpret_cast<const float2*>(p.ln_cpart + ((int64_t)gg * p.ln_cstride + r) * 2);
    }
  }

  // prologue: [A0, W0] [W1] [A1, W2]
  stage_a(0);
  stage_w(0);
  stage_w(1);
  if (nsteps > 2) {
    stage_a(1);
    stage_w(2);
  }
  if (fold_here) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < p.ln_groups) {
        s1 += (double)pv[g].x;
        s2 += (double)pv[g].y;
      }
    const double mean = s1 * (double)p.ln_inv_d;
    double var = s2 * (double)p.ln_inv_d - mean * mean;
    var = var > 0.0 ? var : 0.0;
    my_stats.x = (float)mean;
    my_stats.y = (float)(1.0 / sqrt(var + (double)p.ln_eps));
  }
  VDR_GSTAMP(1);  // addresses computed, ring fill issued
  if (nsteps > 2) wait_vmcnt<NB + NA4 + NB>();
  else wait_vmcnt<NB>();
  __builtin_amdgcn_s_barrier();
  VDR_GSTAMP(2);  // first units landed
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = ld(b_base + j * 1024);
#pragma unroll
  for (int i = 0; i < 2; ++i) alo[i] = ld(a_ch0 + i * 2048);

  int wslot = 0;  // W slot of unit s
  int aoff = 0;   // byte offset of the A slot of unit s
  // one step; ODD: unit s is the second half of its A piece
  auto step = [&](int s, auto odd_tag) {
    constexpr bool ODD = decltype(odd_tag)::value;
    const int a_cur = aoff + (ODD ? a_ch1 : a_ch0);
    // unit s+1 lives in the same piece (even s) or in the other slot (odd s)
    const int a_nxt = ODD ? (aoff ^ APIECE) + a_ch0 : aoff + a_ch1;
    const int nwslot = wslot + 1 == 3 ? 0 : wslot + 1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc.t[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], alo[i], acc.t[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) ahi[i] = ld(a_cur + (2 + i) * 2048);  // lands under the MFMAs just issued
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave holds every fragment of unit s it still needs
    if (s + 1 < nsteps) {
      // everything except what step s-1 issued has landed: W unit s+1 and the A piece of unit s+1
      if (s + 2 < nsteps) {
        if constexpr (ODD) wait_vmcnt<NB>();
        else wait_vmcnt<NB + NA4>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (s + 3 < nsteps && !skip_loads) {
        stage_w(wslot);                                 // W unit s+3 -> the slot unit s just left
        if constexpr (ODD) stage_a(aoff ? 1 : 0);       // A piece (s+3)/2 -> the slot whose two units are now consumed
      }
    }
    // unit s+1 (after the last unit these reads fetch stale, unused bytes)
#pragma unroll
    for (int i = 0; i < 2; ++i) alo[i] = ld(a_nxt + i * 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc.t[j][2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], ahi[i], acc.t[j][2 + i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fb[j] = ld(b_base + nwslot * WUNIT + j * 1024);  // B[j] of unit s is dead: fetch unit s+1's under the remaining MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    wslot = nwslot;
    if constexpr (ODD) aoff ^= APIECE;
  };
  for (int s = 0; s < nsteps; s += 2) {
    step(s, std::false_type{});
    step(s + 1, std::true_type{});
  }

  if (!do_epi && acc.t[0][0][0] != 12345.678f) return;  // ablation (tuning builds): no epilogue, accumulators stay live
#ifdef VDR_GEMM_STAMPS
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc.t[j][i]));
#endif
  VDR_GSTAMP(3);  // main loop done (accumulators complete)
  if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
    if (!p.out_f32 && p.win_ws == 0 && !p.ln_part && !p.ln_cpart) {
      // write-once output through the bf16 epilogue: its constants are requested NOW, into the registers the operand
      // fragments have just left, and used after the barrier (see EpiPre)
      int lane_here = lane;  // opaque: the loads depend on nothing in the loop, hipcc would hoist them above it
      asm volatile("" : "+v"(lane_here)::"memory");
      const EpiPre pre = epilogue_bf16_prefetch(p, m0 + wm * 64, n0 + wn * 64, lane_here);
      __syncthreads();
      VDR_GSTAMP(4);
      epilogue_bf16<EPI>(p, acc, smem + wave * 8192, m0 + wm * 64, n0 + wn * 64, lane, nullptr, pre);
#ifdef VDR_GEMM_STAMPS
      asm volatile("" ::: "memory");
      VDR_GSTAMP(5);
      st[7] = wall_clock64();
      if (lane == 0 && p.stamps) {
        unsigned long long* d = p.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = st[i];
      }
#endif
      return;
    }
  }
  __syncthreads();  // every wave is done with the ring: its memory becomes the staging area
  VDR_GSTAMP(4);
  constexpr int STATS_OFF = NW * 32 * 272;  // behind the wave-private staging images
  if (p.ln_cpart) {
    if (tid < BM) reinterpret_cast<float2*>(smem + STATS_OFF)[tid] = my_stats;
    __syncthreads();
  }
  epilogue_tile<EPI>(p, acc, smem, wave, m0 + wm * 64, n0 + wn * 64, lane,
                     p.ln_cpart ? reinterpret_cast<const float2*>(smem + STATS_OFF) + wm * 64 : nullptr);
#ifdef VDR_GEMM_STAMPS
  asm volatile("" ::: "memory");
  VDR_GSTAMP(5);  // epilogue instructions issued (stores may still be in flight)
  st[7] = wall_clock64();
  if (lane == 0 && p.stamps) {
    unsigned long long* d = p.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
    for (int i = 0; i < 8; ++i) d[i] = st[i];
  }
#endif
#undef VDR_GSTAMP
}

// Producer-side finalisation of the LayerNorm statistics (GemmK::fin_stats; residual epilogue of the ring4 kernels).  A block
// of BM tile rows gets its (sum, sumsq) partials from tiles_n workgroups -- one per tile column -- wherever on the chip they
// run.  Each of them, after its own partial stores are acknowledged (agent-scope stores in epilogue_resid_impl, counted out
// by vmcnt(0)), bumps the block's counter; the one that sees tiles_n - 1 is the last, reads all partials of the block's rows
// back with agent-scope loads, writes (mean, rstd) -- ln_finalize_kernel's arithmetic: double, groups in order, so the
// statistics are bitwise those of the separate launch -- and zeroes the counter for the next launch.  No fence: a release
// fence at agent scope writes the XCD's whole L2 back (measured on a stream-K hand-off: fc2 214 -> 298 us).
// Takes the ln_finalize launch between a residual GEMM and the fold's consumer away (23 launches, 0.15 ms by the profiler in
// the headline step) -- and measures EQUAL in the forward (8.73 vs 8.71-8.74 ms; ViT-L/14@336 25.96 vs 25.99): the wait for the
// stores' acknowledgement and the counter's round trip stand at the end of every tile of every workgroup.  Hence an option
// (vdr_config.ln_fin_fused), off by default.
template <int BM>
VDR_DEV void finalize_rows_if_last(const GemmK& q, int tm, char* smem) {
  if (!q.fin_stats) return;  // (kernel argument: uniform)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();  // every thread's partial stores are out; the staging area is read out
  if (threadIdx.x == 0) {
    const uint32_t seen = __hip_atomic_fetch_add(q.fin_cnt + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<volatile uint32_t*>(smem) = seen;
  }
  __syncthreads();
  const uint32_t seen = *reinterpret_cast<volatile uint32_t*>(smem);
  if (seen + 1 == (uint32_t)q.tiles_n) {
    const int64_t row = (int64_t)tm * BM + threadIdx.x;
    if ((int)threadIdx.x < BM && row < q.M) {
      double s1 = 0.0, s2 = 0.0;
      for (int g0 = 0; g0 < q.fin_groups; g0 += 16) {
        uint64_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {  // 16 independent loads in flight
          const int g = g0 + j < q.fin_groups ? g0 + j : q.fin_groups - 1;
          v[j] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(q.ln_part + ((int64_t)g * q.part_stride + row) * 2),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (g0 + j < q.fin_groups) {
            s1 += (double)__uint_as_float((uint32_t)v[j]);
            s2 += (double)__uint_as_float((uint32_t)(v[j] >> 32));
          }
      }
      const double mean = s1 * (double)q.fin_inv_d;
      double var = s2 * (double)q.fin_inv_d - mean * mean;
      var = var > 0.0 ? var : 0.0;
      float2 o;
      o.x = (float)mean;
      o.y = (float)(1.0 / sqrt(var + (double)q.fin_eps));
      *reinterpret_cast<float2*>(q.fin_stats + row * 2) = o;
    }
    if (threadIdx.x == 0) __hip_atomic_store(q.fin_cnt + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// TAG: no effect on the code -- a second symbol for the same instantiation, so that a profile separates the two residual
// GEMMs of a block (TAG 1 = K > N: fc2; TAG 0: the out-projection and everything else)
template <int WAVES_M, int WAVES_N, int EPI, int TAG = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 4) void gemm_ring4_kernel(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wg = xcd_remap(blockIdx.x, p.nwg);
  int tm, tn;
  tile_of(p, wg, tm, tn);
  gemm_ring4_body<WAVES_M, WAVES_N, EPI>(p, (int64_t)tm * (WAVES_M * 64), tn * (WAVES_N * 64), smem);
  if constexpr (epi_base(EPI) == EPI_BIAS_RESID) finalize_rows_if_last<WAVES_M * 64>(p, tm, smem);
}

// Persistent form of ring4: as many workgroups as the chip holds at once (two per CU), each walking the tile list with
// that stride.  A workgroup slot is held until its last output store has drained and a fresh workgroup has been
// launched into it -- stamped (tools/micro/gemm_stamps.hip): the 512 slots are occupied 81-85 % of a launch -- and the
// loop saves the relaunch.  Measured at M = 50432 (tools/kbench.py in alternating processes, VDR_GEMM_PERSISTENT=0 / 1):
// fc2 243 -> 235 us, proj 83 -> 82; whole ViT-B forward 10.515 -> 10.445 ms (tools/ab_forward.py).  qkv / fc1 LOSE 4-6 %
// in this form -- with their output stores removed altogether as well, so it is not the store drain in front of the next
// tile's loads; the loop's price is the barrier between a tile's epilogue and the next tile's ring fill -- and keep one
// tile per workgroup: the persistent form is used for the residual epilogue only.  Two things made an earlier attempt lose 13 %: (i) kept live across the loop, the ~75 SGPRs of the
// argument block and everything loop-invariant derived from them spill (100 SGPRs, 29 VGPRs to scratch) -- the block is
// re-read per tile through the kernarg pointer made opaque; (ii) hipcc hoists the lane-derived addresses of the body out
// of the loop (+20 VGPRs at a 128-register budget) -- the body takes an opaque copy of threadIdx.x.
template <int WAVES_M, int WAVES_N, int EPI, int TAG = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 4) void gemm_ring4p_kernel(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef const __attribute__((address_space(4))) GemmK* kernarg_ptr;
  kernarg_ptr pk = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();  // (p is the only argument)
  const int nwg = p.nwg, stride = gridDim.x;
  for (int id = blockIdx.x; id < nwg; id += stride) {
    asm volatile("" : "+s"(pk));
#ifdef __HIP_DEVICE_COMPILE__
    GemmK q;
    {
      constexpr int NWORDS = sizeof(GemmK) / 4;
      uint32_t w[NWORDS];
      const __attribute__((address_space(4))) uint32_t* src = (const __attribute__((address_space(4))) uint32_t*)pk;
#pragma unroll
      for (int i = 0; i < NWORDS; ++i) w[i] = src[i];
      __builtin_memcpy(&q, w, sizeof(GemmK));
      // The copy went through integers, and with it the knowledge that these point to global memory: every access of the
      // tile became a FLAT instruction (slower, and counted in both vmcnt and lgkmcnt, so each one is waited out).  The
      // pointers are taken from the argument itself instead -- hipcc promotes those to global -- and only the scalars are
      // re-read per tile (the pointers a given epilogue uses are a dozen SGPRs, not the 75 of the whole block).
#define VDR_GLOBAL(f) q.f = p.f
      VDR_GLOBAL(A);
      VDR_GLOBAL(W);
      VDR_GLOBAL(bias);
      VDR_GLOBAL(resid);
      VDR_GLOBAL(resid32);
      VDR_GLOBAL(C32);
      VDR_GLOBAL(gamma);
      VDR_GLOBAL(pos);
      VDR_GLOBAL(C);
      VDR_GLOBAL(ln_stats);
      VDR_GLOBAL(colsum);
      VDR_GLOBAL(ln_cpart);
      VDR_GLOBAL(ln_part);
      VDR_GLOBAL(fin_stats);
      VDR_GLOBAL(fin_cnt);
      VDR_GLOBAL(sA);
      VDR_GLOBAL(sW);
      VDR_GLOBAL(sC);
#undef VDR_GLOBAL
    }
#else
    const GemmK q = p;
#endif
    const int wg = xcd_remap(id, nwg);
    int tm, tn;
    tile_of(q, wg, tm, tn);
    gemm_ring4_body<WAVES_M, WAVES_N, EPI, true>(q, (int64_t)tm * (WAVES_M * 64), tn * (WAVES_N * 64), smem);
    if constexpr (epi_base(EPI) == EPI_BIAS_RESID) finalize_rows_if_last<WAVES_M * 64>(q, tm, smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area is read out: the next tile's ring fill may overwrite it
    __syncthreads();
  }
}

// PIPE: 3x = ring3 with x LDS slots, 4x = ring3k with x super-slots, 50 = ring4
template <int WAVES_M, int WAVES_N, int PIPE, int E, int TAG = 0>
static auto launch_pick() -> void (*)(GemmK) {
  if constexpr (PIPE >= 50)
    return gemm_ring4_kernel<WAVES_M, WAVES_N, E, TAG>;
  else if constexpr (PIPE >= 40)
    return gemm_ring3k_kernel<PIPE - 40, E>;
  else
    return gemm_ring3_kernel<WAVES_M, WAVES_N, PIPE - 30, E>;
}

#ifdef VDR_GEMM_STAMPS
inline unsigned long long* g_gemm_stamps = nullptr;  // tools/micro/gemm_stamps.hip
#endif
// the persistent form exists for the residual epilogue of ring4 (tuning builds: for every epilogue, variant 2xx)
template <int WAVES_M, int WAVES_N, int PIPE, int E, int TAG = 0>
static auto launch_pick_persistent() -> void (*)(GemmK) {
#ifdef VDR_TUNING
  if constexpr (PIPE >= 50) return gemm_ring4p_kernel<WAVES_M, WAVES_N, E, TAG>;
#else
  if constexpr (PIPE >= 50 && epi_base(E) == EPI_BIAS_RESID) return gemm_ring4p_kernel<WAVES_M, WAVES_N, E, TAG>;
#endif
  return launch_pick<WAVES_M, WAVES_N, PIPE, E, TAG>();
}

inline int g_gemm_ablation = 0;  // tuning builds only (variant / 100 of vdr_op_linear)
inline int g_gemm_gn = -1;       // tuning builds only: column-group width override (variant / 1000 - 1)

// integer tuning knob from the environment: read in tuning builds (-DVDR_TUNING, `make tuning`) only; the shipped
// library has no environment dependence
static inline int tuning_env(const char* name, int dflt) {
#ifdef VDR_TUNING
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

template <int WAVES_M, int WAVES_N, int PIPE>
static hipError_t launch_cfg(const GemmArgs& a, int epi, hipStream_t s) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  constexpr int NWV = PIPE >= 40 && PIPE < 50 ? 8 : WAVES_M * WAVES_N;
  GemmK k;
  k.A = (const bf16_t*)a.A;
  k.W = (const bf16_t*)a.W;
  k.w_il = a.w_interleaved;
  k.bias = a.bias;
  k.resid = (const bf16_t*)a.resid;
  k.resid32 = a.resid32;
  k.C32 = a.C32;
  if (a.resid32 || a.C32) {  // the fp32 residual stream: its own instantiation of the residual kernels
    if (epi != EPI_BIAS_RESID || !a.resid32 || !a.C32 || a.win_ws || (PIPE >= 40 && PIPE < 50)) return hipErrorInvalidValue;
    epi = EPI_BIAS_RESID32;
  }
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
    // column-group width: tiles are walked in groups of gn tile columns (tile_of) so that a group's W panels (BN x K
    // bf16 each) stay in the XCD's 4 MB L2 next to the A row panels in flight.  Measured:
    //   per launch, M = 50432, interleaved rounds (tools/kbench.py --variants (gn+1)*1000+26): qkv (9 columns, panel
    //   393 KB) 0.192 ms row-major, 0.174 / 0.171 / 0.175 / 0.176 for gn = 2 / 3 / 4 / 5; fc1 (12 columns) 0.276
    //   row-major, 0.266 / 0.265 / 0.266 for gn = 2 / 4 / 5, 0.275 for 6;
    //   whole forward, same device (tools/ab_forward.py): ViT-B (K = 768) 10.54 ms row-major, 10.37 / 10.40 / 10.40 for
    //   gn = 3 / 4 / 5; ViT-L/14 (K = 1024) 26.91 row-major, 26.47 / 26.64 for gn = 2 / 3; ViT-g/14 (K = 1536, 32 tile
    //   columns in w12) 22.79 row-major, 20.09 / 20.07 for gn = 2 / 3 (-12 %).
    // Rule: about 1.7 MB of W per group, i.e. gn = 4 / 3 / 2 for K = 768 / 1024 / 1536.  (Round 1 grouped only when W
    // as a whole exceeded the L2; with the whole-line operand loads the L2 misses weigh more and qkv gains too.)
    // K >= 3072 panels (>= 1.6 MB) give gn = 1: row-major, which is all a 3-column fc2 can use anyway.
    VDR_KNOB int gn_env = tuning_env("VDR_GEMM_GN", -1);
    const size_t panel = (size_t)BN * a.K * 2;
    int gn = (int)((1700u << 10) / panel);
    if (gn < 2 || gn >= k.tiles_n) gn = 0;  // a single column at a time re-reads A once per column: never better than row-major
    k.gn = g_gemm_gn >= 0 ? g_gemm_gn : gn_env >= 0 ? gn_env : gn;
  }
  const int64_t nwg = tiles_m * k.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffff) return hipErrorInvalidValue;
  k.nwg = (int)nwg;
  k.win_ws = a.win_ws;
  k.win_g = a.win_g;
  k.a_rpg = a.a_rpg;
  k.a_gs = a.a_gs;
  k.a_is = a.a_is;
  k.out_f32 = a.out_f32;
  if (a.patch_p) {
    const int P = a.patch_p;
    if (epi != EPI_PATCH || PIPE != 50 || (P != 8 && P != 16 && P != 32) || a.patch_g <= 0 || a.patch_C <= 0 || a.K != a.patch_C * P * P ||
        a.M % ((int64_t)a.patch_g * a.patch_g) || a.a_rpg || ((uintptr_t)a.A & 15))
      return hipErrorInvalidValue;
    k.pg_ps = P == 8 ? 3 : P == 16 ? 4 : 5;
    k.pg_g = a.patch_g;
    k.pg_C = a.patch_C;
  }
  {
    VDR_KNOB int nt_env = tuning_env("VDR_GEMM_NT", -1);
    const bool big = (double)a.M * (double)a.ldc * 2.0 >= 128e6 && !a.resid;  // write-once output larger than half the Infinity Cache
    k.nt_store = nt_env >= 0 ? nt_env : (big ? 1 : 0);
  }
  k.ln_stats = a.ln_stats;
  k.colsum = a.colsum;
  k.ln_part = a.ln_part;
  k.part_stride = a.part_stride;
  k.ln_fold = a.ln_stats || a.ln_cpart;
  if (a.fin_stats) {  // producer-side finalisation: ring4 kernels, residual epilogue, rows stored where they are computed
    if (PIPE < 50 || epi_base(epi) != EPI_BIAS_RESID || !a.ln_part || !a.fin_cnt || a.win_ws || (a.N & 63)) return hipErrorInvalidValue;
    k.fin_stats = a.fin_stats;
    k.fin_cnt = a.fin_cnt;
    k.fin_groups = a.N / 64;
    k.fin_inv_d = a.fin_inv_d;
    k.fin_eps = a.fin_eps;
  }
  if (a.ldc >= ((int64_t)1 << 24)) return hipErrorInvalidValue;  // (epilogue_bf16 addresses a wave tile with 32-bit byte offsets)
  // (the residual epilogue, epilogue_resid: bf16 in place or out of place, no consumer-side fold, 32-bit row numbers)
  if (epi_base(epi) == EPI_BIAS_RESID && (k.ln_fold || a.out_f32 || a.M >= ((int64_t)1 << 31))) return hipErrorInvalidValue;
  if (a.ln_cpart) {
    if ((PIPE >= 40 && PIPE < 50) || a.ln_groups < 1 || a.ln_groups > 16 || a.ln_stats) return hipErrorInvalidValue;
    k.ln_cpart = a.ln_cpart;
    k.ln_groups = a.ln_groups;
    k.ln_cstride = a.ln_cstride;
    k.ln_inv_d = a.ln_inv_d;
    k.ln_eps = a.ln_eps;
  }
  k.abl = g_gemm_ablation;
#ifdef VDR_GEMM_STAMPS
  k.stamps = g_gemm_stamps;
#endif
  if (PIPE >= 50 && a.a_rpg) return hipErrorInvalidValue;  // the two-stride A gather stays on ring3

  dim3 grid((unsigned)k.nwg), block(NWV * 64);
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  bool persistent = PIPE >= 50 && epi_base(epi) == EPI_BIAS_RESID;  // (see gemm_ring4p_kernel)
#ifdef VDR_TUNING
  {
    VDR_KNOB int pers_env = tuning_env("VDR_GEMM_PERSISTENT", -1);
    if (pers_env >= 0) persistent = persistent && pers_env;
    if (pers_env == 2 && PIPE >= 50) persistent = true;  // every epilogue (experiments)
    if ((k.abl & 2) && PIPE >= 50) persistent = true;
  }
#endif
  const size_t staging = (size_t)WAVES_M * WAVES_N * 32 * 272;  // epilogue images (ring3 / ring4: one per wave)
  size_t lds;
  if (PIPE >= 50) {
    lds = (size_t)2 * BM * 128 + (size_t)3 * BN * 64;
    const size_t need = staging + (a.ln_cpart ? (size_t)BM * 8 : 0);
    if (need > lds) lds = need;
  } else if (PIPE >= 40) {
    lds = (size_t)(BM + BN) * 64 * 2 * (PIPE - 40);
    if (lds < (size_t)65536 + 4 * 32 * 272) lds = (size_t)65536 + 4 * 32 * 272;  // K reduction + staging
  } else {
    lds = (size_t)(BM + BN) * 64 * (PIPE - 30);
    if (lds < staging) lds = staging;
    if (a.ln_cpart) {  // (mean, rstd) of the tile's BM rows, behind the ring / staging area
      k.stats_off = (int)lds;
      lds += (size_t)BM * 8;
    }
  }
#ifdef VDR_TUNING
  {  // tools/: extra dynamic LDS per workgroup (e.g. 40000 on ring4: one workgroup per CU instead of two)
    const int pad = tuning_env("VDR_GEMM_LDS_PAD", 0);
    if (pad > 0) lds += (size_t)pad;
  }
#endif
#define VDR_LAUNCH(E) VDR_LAUNCH_T(E, 0)
#define VDR_LAUNCH_T(E, T)                                                                             \
  case E + 100 * T: {                                                                                  \
    auto fn = launch_pick<WAVES_M, WAVES_N, PIPE, E, T>();                                             \
    if (persistent && launch_pick_persistent<WAVES_M, WAVES_N, PIPE, E, T>() != fn) {                  \
      auto pfn = launch_pick_persistent<WAVES_M, WAVES_N, PIPE, E, T>();                               \
      static int slots_dev[VDR_MAX_DEVICES] = {}; /* workgroups of this instantiation the chip holds at once */ \
      int& slots = slots_dev[dev];                                                                     \
      if (!slots) {                                                                                    \
        int per_cu = 0;                                                                                \
        const int n_cu = device_cu_count(dev);                                                         \
        if (n_cu <= 0 ||                                                                               \
            hipFuncSetAttribute((const void*)pfn, hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                (int)(lds > 65536 ? lds : 65536)) != hipSuccess ||                     \
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)pfn, (int)block.x, lds) != hipSuccess || \
            per_cu <= 0)                                                                               \
          return hipErrorUnknown;                                                                      \
        slots = per_cu * n_cu;                                                                         \
      }                                                                                                \
      if (k.nwg > slots) {                                                                             \
        fn = pfn;                                                                                      \
        grid = dim3((unsigned)slots);                                                                  \
      }                                                                                                \
    }                                                                                                  \
    static size_t lds_set[VDR_MAX_DEVICES][2] = {}; /* per kernel and device: the attribute is raised once, not per launch */ \
    size_t& lset = lds_set[dev][grid.x != (unsigned)k.nwg];                                            \
    if (lds > 65536 && lds > lset) {                                                                   \
      hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                         (int)lds);                                                    \
      if (e != hipSuccess) return e;                                                                   \
      lset = lds;                                                                                      \
    }                                                                                                  \
    hipLaunchKernelGGL(fn, grid, block, lds, s, k);                                                    \
    break;                                                                                             \
  }
  // (the residual GEMM with K > N -- fc2 -- launches the TAG 1 symbol of the same code: profiles tell it from the out-projection)
  switch (epi + (PIPE >= 50 && epi_base(epi) == EPI_BIAS_RESID && a.K > a.N ? 100 : 0)) {
    VDR_LAUNCH(EPI_BIAS)
    VDR_LAUNCH(EPI_BIAS_GELU)
    VDR_LAUNCH(EPI_BIAS_RESID)
    VDR_LAUNCH_T(EPI_BIAS_RESID, 1)
    VDR_LAUNCH(EPI_BIAS_RESID32)
    VDR_LAUNCH_T(EPI_BIAS_RESID32, 1)
    VDR_LAUNCH(EPI_SWIGLU)
    VDR_LAUNCH(EPI_PATCH)
    default:
      return hipErrorInvalidValue;
  }
#undef VDR_LAUNCH
#undef VDR_LAUNCH_T
  return hipGetLastError();
}

}  // namespace vdr
