// bf16 MFMA GEMM kernels with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)
// (all kernel templates + the launch helper; instantiated per tile variant by gemm.hip = ring3 defaults,
// gemm_ring2.hip and gemm_legacy.hip, so the library builds in parallel)
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
#pragma once
#include <cstdlib>

#include "gemm_epi.h"

namespace vdr {


template <int WAVES_M, int WAVES_N, int TM, int TN, int PIPE, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(GemmK p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;  // one K-tile of A and W in LDS
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
  auto stage = [&](int buf) {
    char* dA = sA + buf * STAGE_BYTES;
    char* dB = sB + buf * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      glds16(a_src[q], dA + (wave * NA + q) * 1024);
      a_src[q] += 64;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      glds16(b_src[q], dB + (wave * NB + q) * 1024);
      b_src[q] += 64;
    }
  };
  auto compute = [&](int buf) {
    const char* cA = sA + buf * STAGE_BYTES;
    const char* cB = sB + buf * STAGE_BYTES;
    // fragments double-buffered in registers: the ds_reads of k-step ks+1 are in flight under the
    // MFMAs of k-step ks (the compiler turns the dependency into a counted lgkmcnt)
    bf16x8 af[2][TM], bf[2][TN];
    auto load_frags = [&](int ks, int set) {
      const int ch = ((2 * ks + h) ^ swz) * 16;
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const bf16x8*>(cB + b_off[j] + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const bf16x8*>(cA + a_off[i] + ch);
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this k-step's MFMAs
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], af[ks & 1][i], acc[j][i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (PIPE == 0) {
    // single LDS stage, two barriers per K-tile; latency hidden by 3 co-resident workgroups per CU
    for (int kt = 0; kt < nk; ++kt) {
      stage(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  } else {
    // two LDS stages, ONE barrier per K-tile: the global_load_lds of tile kt+1 is issued right after
    // the barrier and lands under the MFMAs of tile kt
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // tile kt has landed for every wave; everyone is done reading the other stage
      if (kt + 1 < nk) stage((kt + 1) & 1);
      compute(kt & 1);
    }
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


// -------------------------------------------------------------------------------------------------
// Ring-pipelined variant.  K is consumed in 32-deep units; each unit (A rows + W rows, 64-B rows,
// 16-B chunks XOR-swizzled by (row>>2)&3) lives in one of NST LDS slots.  The global_load_lds of
// unit s+NST-1 is issued right after the barrier of step s and stays in flight ACROSS the next
// barriers: the only wait is a counted s_waitcnt vmcnt that retires unit s alone, and the barrier
// is a raw s_barrier (a __syncthreads() would drain the whole ring).  This keeps (NST-1) units per
// workgroup streaming from L2 at all times, which is what the per-CU L2->LDS path needs to reach its
// rate (the earlier variants issue a burst, drain it, and sit at ~half of it).
// -------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int TM, int TN, int NST, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (WAVES_M * WAVES_N == 4 ? 2 : 2)) void gemm_ring_kernel(GemmK p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int UNIT = (BM + BN) * 64;  // bytes of one 32-deep unit
  constexpr int NA = BM / 16 / NW;      // global_load_lds per wave per unit (16 rows x 64 B each)
  constexpr int NB = BN / 16 / NW;
  constexpr int G = NA + NB;
  static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile/wave mismatch");
  static_assert((NST - 2) * G <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];

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

  const int srow = lane >> 2;  // row within the 16-row piece
  const int spc = lane & 3;    // physical 16-B chunk within the 64-B row
  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int r = (wave * NA + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    a_src[q] = p.A + gr * p.lda + c * 8;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    b_src[q] = p.W + (int64_t)gr * p.ldw + c * 8;
  }

  const int swz = (lane >> 2) & 3;  // == (row >> 2) & 3: tile row bases are multiples of 32
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (wm * TM * 32 + i * 32 + l31) * 64;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = BM * 64 + (wn * TN * 32 + j * 32 + l31) * 64;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  const int nsteps = p.K >> 5;
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
      b_src[q] += 32;
    }
  };
  auto compute = [&](int slot) {
    const char* c0 = smem + slot * UNIT;
    bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((2 * ks + h) ^ swz) * 16;
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(c0 + b_off[j] + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(c0 + a_off[i] + ch);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][j], af[ks][i], acc[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: units 0 .. NST-2 in flight
#pragma unroll
  for (int u = 0; u < NST - 1; ++u)
    if (u < nsteps) stage(u);
  int slot = 0, stage_slot = NST - 1;
  for (int s = 0; s < nsteps; ++s) {
    const int younger = nsteps - 1 - s;  // units issued after unit s (capped by the ring depth)
    if (younger >= NST - 2) {
      wait_vmcnt<(NST - 2) * G>();
    } else if (NST > 3 && younger == 1) {
      wait_vmcnt<G>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // unit s landed for every wave; slot of unit s-1 is free
    if (s + NST - 1 < nsteps && !(p.abl & 4)) stage(stage_slot);
    if (!(p.abl & 2)) compute(slot);
    slot = slot + 1 == NST ? 0 : slot + 1;
    stage_slot = stage_slot + 1 == NST ? 0 : stage_slot + 1;
  }

  if ((p.abl & 1) && acc[0][0][0] != 12345.678f) return;  // ablation: no epilogue (keeps acc live)
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


template <int EPI, int TM, int TN>
VDR_DEV void epilogue_direct(const GemmK& p, f32x16 (&acc)[TN][TM], int64_t m_base, int n_base, int lane);  // defined below

// -------------------------------------------------------------------------------------------------
// Ring variant 2: the barrier of a step sits in the MIDDLE of its MFMAs.
//   top of step s :  X holds the k-step-0 fragments of unit s (read during step s-1)
//       read Y <- unit s, k-step 1             | 8 MFMAs on X          (LDS reads under MFMAs)
//       lgkmcnt(0); vmcnt: retire unit s+1; s_barrier               (unit s is now dead)
//       global_load_lds unit s+NST -> slot of unit s
//       read X <- unit s+1, k-step 0           | 8 MFMAs on Y
// so no wave ever waits on an LDS read right after a barrier, and the loader runs NST-1 units ahead.
// The epilogue goes through LDS (epilogue_lds) so that all its global traffic is whole lines.
// -------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int TM, int TN, int NST, int EPI>
VDR_DEV void gemm_ring2_body(const GemmK& p, const int64_t m0, const int n0, char* smem) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
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
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int srow = lane >> 2;
  const int spc = lane & 3;
  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int r = (wave * NA + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    const int64_t aoff = p.a_rpg > 0 ? (gr / p.a_rpg) * p.a_gs + (gr % p.a_rpg) * p.a_is : gr * p.lda;
    a_src[q] = p.A + aoff + c * 8;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    b_src[q] = p.W + (int64_t)gr * p.ldw + c * 8;
  }

  const int swz = (lane >> 2) & 3;
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (wm * TM * 32 + i * 32 + l31) * 64;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = BM * 64 + (wn * TN * 32 + j * 32 + l31) * 64;
  const int ch0 = ((0 + h) ^ swz) * 16;  // k-step 0: chunks 0,1
  const int ch1 = ((2 + h) ^ swz) * 16;  // k-step 1: chunks 2,3

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  const int nsteps = p.K >> 5;
  const bool do_epi = __builtin_amdgcn_readfirstlane(p.abl & 1) == 0;
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
      b_src[q] += 32;
    }
  };
  bf16x8 xa[TM], xb[TN], ya[TM], yb[TN];
  auto read_frags = [&](bf16x8 (&fa)[TM], bf16x8 (&fb)[TN], int slot, int ch) {
    const char* c0 = smem + slot * UNIT;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(c0 + b_off[j] + ch);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(c0 + a_off[i] + ch);
  };
  auto mfmas = [&](bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
  };
  // retire unit u: every global_load_lds of units <= u issued by this wave has landed
  auto retire = [&](int u, int issued_upto) {
    int younger = issued_upto - u;  // units issued after unit u
    if (younger >= NST - 1) {
      wait_vmcnt<(NST - 1) * G>();
    } else if (younger == NST - 2 && NST >= 3) {
      wait_vmcnt<(NST - 2) * G>();
    } else if (younger == 2 && NST >= 5) {
      wait_vmcnt<2 * G>();
    } else if (younger == 1 && NST >= 4) {
      wait_vmcnt<G>();
    } else {
      wait_vmcnt<0>();
    }
  };

  // prologue: fill the ring
  int issued = -1;
#pragma unroll
  for (int u = 0; u < NST; ++u)
    if (u < nsteps) {
      stage(u);
      issued = u;
    }
  retire(0, issued);
  __builtin_amdgcn_s_barrier();
  read_frags(xa, xb, 0, ch0);
  int slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    mfmas(xa, xb);                       // k-step 0 of unit s (fragments read during step s-1)
    __builtin_amdgcn_sched_barrier(0);
    read_frags(ya, yb, slot, ch1);       // k-step 1 of unit s: lands under the MFMAs just issued
    const int nslot = slot + 1 == NST ? 0 : slot + 1;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): unit s fully consumed from LDS by this wave
    if (s + 1 < nsteps) {
      retire(s + 1, issued);
      __builtin_amdgcn_s_barrier();
      if (s + NST < nsteps) {
        stage(slot);
        issued = s + NST;
      }
      read_frags(xa, xb, nslot, ch0);    // k-step 0 of unit s+1: lands under the next MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(ya, yb);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  }

  if (!do_epi && acc[0][0][0] != 12345.678f) return;  // ablation: no epilogue (keeps acc live)
  if (p.epi_lds) {
    __syncthreads();  // every wave is done with the ring: its memory becomes the staging area
    epilogue_lds<EPI, TM, TN>(p, acc, smem + wave * (32 * 272), m0 + wm * TM * 32, n0 + wn * TN * 32, lane);
  } else {
    epilogue_direct<EPI, TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane);
  }
}


// -------------------------------------------------------------------------------------------------
// Ring variant 3: the ring2 pipeline on the 16x16x32 MFMA shape.  Same LDS image, same bytes read per unit
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
  constexpr int TM = 2, TN = 2;  // in units of 32: the wave tile is 64 x 64
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
    const int c = spc ^ ((r >> 2) & 3);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    const int64_t aoff = p.a_rpg > 0 ? (gr / p.a_rpg) * p.a_gs + (gr % p.a_rpg) * p.a_is : gr * p.lda;
    a_src[q] = p.A + aoff + c * 8;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int r = (wave * NB + q) * 16 + srow;
    const int c = spc ^ ((r >> 2) & 3);
    int gr = n0 + r;
    gr = gr < p.N ? gr : p.N - 1;
    b_src[q] = p.W + (int64_t)gr * p.ldw + c * 8;
  }

  // fragment of MFMA tile t (16 rows): lane (r = lane & 15, q = lane >> 4) reads the 16-B chunk q of row r; the
  // chunk swizzle (row >> 2) & 3 of the LDS image equals (lane >> 2) & 3 because tile bases are multiples of 16
  const int r15 = lane & 15;
  const int chq = ((lane >> 4) ^ ((lane >> 2) & 3)) * 16;
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
  const bool do_epi = __builtin_amdgcn_readfirstlane(p.abl & 1) == 0;
  const bool skip_loads = __builtin_amdgcn_readfirstlane(p.abl & 4) != 0;  // diagnostic: the ring is filled once, then reused
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
      b_src[q] += 32;
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
  epilogue_lds<EPI, TM, TN>(p, acc, smem + wave * (32 * 272), m0 + wm * 64, n0 + wn * 64, lane,
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
    const int c = spc ^ ((r >> 2) & 3);
    int64_t gr = m0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    const int64_t aoff = p.a_rpg > 0 ? (gr / p.a_rpg) * p.a_gs + (gr % p.a_rpg) * p.a_is : gr * p.lda;
    a_src = p.A + aoff + c * 8;
    int gn = n0 + r;
    gn = gn < p.N ? gn : p.N - 1;
    b_src = p.W + (int64_t)gn * p.ldw + c * 8;
  }
  const int r15 = lane & 15;
  const int chq = ((lane >> 4) ^ ((lane >> 2) & 3)) * 16;
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
      glds16(b_src + g * 32, d + g * UNIT + BM * 64 + wave * 1024);
    }
    a_src += 64;
    b_src += 64;
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

// Workgroups [0, p.nwg_big) compute BM x BN tiles of rows [0, p.m_split); the remaining workgroups
// compute (BM/2) x BN tiles of rows [p.m_split, M).  The hardware dispatches workgroups in index
// order, so the half-height tiles form the last, partial round: a tile count that leaves the final
// round x % full costs x/2 % of a round instead of a whole one (ViT-B proj / fc2: 2.31 rounds of
// 128 x 256 tiles -> 2 rounds + one round of 64 x 256 tiles).

// Direct epilogue, no LDS: in the transposed-MFMA accumulator a row's columns 8g..8g+3 sit in lane r
// and 8g+4..8g+7 in lane r+32.  One v_permlane32_swap per register pair (groups g, g+1) leaves lanes
// 0-31 with the 8 consecutive columns 8g..8g+7 and lanes 32-63 with 8(g+1)..8(g+1)+7, so the whole
// epilogue (bias, LayerNorm fold, GELU, residual, row statistics) runs on 8-column octets with 16-byte
// global accesses.  A store wave-instruction of 32 rows x 32 B costs the CU's store path the same as
// one of 8 rows x 128 B (measured 28.4 vs 29.3 GB/s per CU), so nothing is lost against the LDS-staged
// form, and the LDS ring stays untouched (no barrier between the main loop and the epilogue).
VDR_DEV void swap_halves(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

template <int EPI, int TM, int TN>
VDR_DEV void epilogue_direct(const GemmK& p, f32x16 (&acc)[TN][TM], int64_t m_base, int n_base, int lane) {
  static_assert(TN % 2 == 0, "column tiles are processed in pairs");
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t m = m_base + i * 32 + l31;
    float mu = 0.0f, rs = 1.0f;
    if (p.ln_stats) {
      const int64_t mm = m < p.M ? m : p.M - 1;
      const float2 t = *reinterpret_cast<const float2*>(p.ln_stats + 2 * mm);
      mu = t.x;
      rs = t.y;
    }
#pragma unroll
    for (int jp = 0; jp < TN / 2; ++jp) {
      float s1 = 0.0f, s2 = 0.0f;
      int64_t orow = -1;
      if (EPI != EPI_SWIGLU) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float v[8], u[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a = acc[j][i][8 * gp + e], b = acc[j][i][8 * gp + 4 + e];
              swap_halves(a, b);
              v[e] = u[e] = a;
              v[4 + e] = u[4 + e] = b;
            }
            float t1, t2;
            orow = epi_oct<EPI>(p, v, u, m, n_base + j * 32 + 8 * (2 * gp + h), t1, t2, mu, rs);
            s1 += t1;
            s2 += t2;
          }
        }
        if (p.ln_part) {
          // this lane pair (r, r+32) covered the row's 64 columns of this block
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (h == 0 && orow >= 0 && n_base + jp * 64 < p.N) {
            float* dst = p.ln_part + ((int64_t)((n_base + jp * 64) >> 6) * p.part_stride + orow) * 2;
            dst[0] = s1;
            dst[1] = s2;
          }
        }
      } else {
        // gate pairs: tile 2jp holds x1, tile 2jp+1 the matching x2 columns
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          float v[8], u[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = acc[2 * jp][i][8 * gp + e], b = acc[2 * jp][i][8 * gp + 4 + e];
            swap_halves(a, b);
            v[e] = a;
            v[4 + e] = b;
            float c = acc[2 * jp + 1][i][8 * gp + e], d = acc[2 * jp + 1][i][8 * gp + 4 + e];
            swap_halves(c, d);
            u[e] = c;
            u[4 + e] = d;
          }
          float t1, t2;
          epi_oct<EPI>(p, v, u, m, n_base + 2 * jp * 32 + 8 * (2 * gp + h), t1, t2, mu, rs);
        }
      }
    }
  }
}

// waves per SIMD the register allocator must leave room for: 64-register accumulators (2x2 MFMA tiles
// per wave) run 4 waves per SIMD (16 per CU) -- the per-CU load rate scales with the number of waves
// that issue vector-memory instructions (measured: 35 GB/s with 4 waves, 75-79 GB/s with 8)
template <int TILES>
constexpr int ring2_min_waves() { return TILES <= 4 ? 4 : 2; }

template <int WAVES_M, int WAVES_N, int TM, int TN, int NST, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, ring2_min_waves<TM * TN>()) void gemm_ring2_kernel(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  if ((int)blockIdx.x < p.nwg_big) {
    const int wg = xcd_remap(blockIdx.x, p.nwg_big);
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    gemm_ring2_body<WAVES_M, WAVES_N, TM, TN, NST, EPI>(p, (int64_t)tm * BM, tn * BN, smem);
  } else {
    if constexpr (WAVES_M == 2 && (TN % 4) == 0) {
      // same wave count and BN, half the rows: waves laid out 1 x (2*WAVES_N), wave tile (TM*32) x (TN/2*32)
      const int wg = xcd_remap(blockIdx.x - p.nwg_big, p.nwg - p.nwg_big);
      const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
      gemm_ring2_body<1, WAVES_N * 2, TM, TN / 2, NST, EPI>(p, p.m_split + (int64_t)tm * (BM / 2), tn * BN, smem);
    }
  }
}

inline int g_gemm_ablation = 0;
inline int g_gemm_split = 0;  // VDR_GEMM_SPLIT=1 enables the mixed-height last round (measured: no gain, the
                       // dispatcher already back-fills CUs as workgroups retire; kept for A/B)

template <int WAVES_M, int WAVES_N, int TM, int TN, int PIPE, int E>
static auto launch_pick() -> void (*)(GemmK) {
  if constexpr (PIPE >= 40)
    return gemm_ring3k_kernel<PIPE - 40, E>;
  else if constexpr (PIPE >= 30)
    return gemm_ring3_kernel<WAVES_M, WAVES_N, PIPE - 30, E>;
  else if constexpr (PIPE >= 20)
    return gemm_ring2_kernel<WAVES_M, WAVES_N, TM, TN, PIPE - 20, E>;
  else if constexpr (PIPE >= 10)
    return gemm_ring_kernel<WAVES_M, WAVES_N, TM, TN, PIPE - 10, E>;
  else
    return gemm_kernel<WAVES_M, WAVES_N, TM, TN, PIPE, E>;
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int PIPE>
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
  k.tiles_m = (int)tiles_m;
  {
    // column-group width (ring3 kernels): as many W panels (BN x K bf16) as fit in half of an XCD's 4 MB L2, and only
    // when W as a whole exceeds that L2 (ViT-B: fc1 4.7 MB yes, qkv 3.5 MB no -- measured: fc1 HBM-side reads 790 ->
    // 446 MB and -2.6 % time, qkv +3 % time).  VDR_GEMM_GN overrides (0 = row-major).
    static const int gn_env = [] { const char* e = getenv("VDR_GEMM_GN"); return e && *e ? atoi(e) : -1; }();
    const size_t panel = (size_t)BN * a.K * 2, whole = (size_t)a.N * a.K * 2;
    int gn = whole > (4u << 20) ? (int)((2u << 20) / panel) : 0;
    if (gn < 2) gn = 0;  // a single column at a time re-reads A once per column: never better than row-major
    k.gn = gn_env >= 0 ? gn_env : gn;
  }
  const int64_t nwg = tiles_m * k.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffff) return hipErrorInvalidValue;
  {
    static bool once = false;
    if (!once) {
      const char* e = getenv("VDR_GEMM_SPLIT");
      if (e && *e) g_gemm_split = atoi(e);
      once = true;
    }
  }
  k.nwg = (int)nwg;
  k.nwg_big = (int)nwg;
  k.m_split = a.M;
  if (PIPE >= 20 && PIPE < 30 && WAVES_M == 2 && (TN % 4) == 0 && g_gemm_split != 0) {
    // mixed tile heights: finish with one round of half-height tiles when that is shorter than a
    // partial round of full ones
    static int n_cu = 0;
    if (!n_cu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
      if (n_cu <= 0) n_cu = 256;
    }
    const size_t lds_b = (size_t)(BM + BN) * 64 * (PIPE - 20);
    int per_cu = (int)(160 * 1024 / lds_b);
    const int by_waves = 8 / (WAVES_M * WAVES_N);  // launch bound: 2 waves per SIMD
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    const int64_t slots = (int64_t)n_cu * per_cu;
    const int64_t full = nwg / slots;
    const int64_t rem = nwg - full * slots;
    if (full >= 1 && rem > 0) {
      const int64_t panels_big = full * slots / k.tiles_n;
      const int64_t m_split = panels_big * BM;
      const int64_t small_tiles = (a.M - m_split + BM / 2 - 1) / (BM / 2) * k.tiles_n;
      if (m_split < a.M && small_tiles <= slots) {
        k.nwg_big = (int)(panels_big * k.tiles_n);
        k.m_split = m_split;
        k.nwg = k.nwg_big + (int)small_tiles;
      }
    }
  }
  k.win_ws = a.win_ws;
  k.win_g = a.win_g;
  k.a_rpg = a.a_rpg;
  k.a_gs = a.a_gs;
  k.a_is = a.a_is;
  k.out_f32 = a.out_f32;
  {
    static const int nt_env = [] { const char* e = getenv("VDR_GEMM_NT"); return e && *e ? atoi(e) : -1; }();
    const bool big = (double)a.M * (double)a.ldc * 2.0 >= 128e6 && !a.resid;  // write-once output larger than half the Infinity Cache
    k.nt_store = nt_env >= 0 ? nt_env : (big ? 1 : 0);
  }
  k.ln_stats = a.ln_stats;
  k.colsum = a.colsum;
  k.ln_part = a.ln_part;
  k.part_stride = a.part_stride;
  k.ln_fold = a.ln_stats || a.ln_cpart;
  if (a.ln_cpart) {
    if (!(PIPE >= 30 && PIPE < 40) || a.ln_groups < 1 || a.ln_groups > 16 || a.ln_stats) return hipErrorInvalidValue;
    k.ln_cpart = a.ln_cpart;
    k.ln_groups = a.ln_groups;
    k.ln_cstride = a.ln_cstride;
    k.ln_inv_d = a.ln_inv_d;
    k.ln_eps = a.ln_eps;
  }
  k.abl = g_gemm_ablation;
  k.stagger = 0;
  {
    static int epi_lds = -1;
    if (epi_lds < 0) {
      const char* e = getenv("VDR_GEMM_EPI_LDS");
      epi_lds = e && *e ? atoi(e) : 1;  // measured: the LDS-staged form is 3-20 % faster in this (non-persistent) kernel
    }
    k.epi_lds = epi_lds;
  }

  const dim3 grid((unsigned)k.nwg), block(PIPE >= 40 ? 512 : WAVES_M * WAVES_N * 64);  // ring3k: two wave groups per tile
  constexpr int RING_SLOTS = PIPE >= 40 ? 2 * (PIPE - 40) : PIPE >= 30 ? PIPE - 30 : PIPE - 20;
  const size_t lds_ring2 = (size_t)(BM + BN) * 64 * RING_SLOTS > (size_t)WAVES_M * WAVES_N * 32 * 272
                               ? (size_t)(BM + BN) * 64 * RING_SLOTS
                               : (size_t)WAVES_M * WAVES_N * 32 * 272;
  const size_t lds_k = lds_ring2 > (size_t)65536 + 4 * 32 * 272 ? lds_ring2 : (size_t)65536 + 4 * 32 * 272;  // ring3k: reduction + staging
  size_t lds = PIPE >= 40 ? lds_k : PIPE >= 20 ? lds_ring2 : PIPE >= 10 ? (size_t)(BM + BN) * 64 * (PIPE - 10) : (size_t)(BM + BN) * 128 * (PIPE ? 2 : 1);
  if (a.ln_cpart) {  // (mean, rstd) of the tile's BM rows, behind the ring / staging area
    k.stats_off = (int)lds;
    lds += (size_t)BM * 8;
  }
#define VDR_LAUNCH(E)                                                                             \
  case E: {                                                                                       \
    auto fn = launch_pick<WAVES_M, WAVES_N, TM, TN, PIPE, E>();                                   \
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

}  // namespace vdr
