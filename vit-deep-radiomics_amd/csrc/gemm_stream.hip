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
};

constexpr int ST_BM = 128, ST_BN = 256;
constexpr int ST_STAGE = (ST_BM + ST_BN) * 128;  // 48 KB
constexpr int ST_WOFF = ST_BM * 128;             // W image behind the A image of a stage
constexpr int ST_CONST = 3 * ST_STAGE;           // per-tile constants: 2 x 4 KB (by tile parity)
constexpr int ST_LDS = ST_CONST + 2 * 4096;      // 155648 B
constexpr int ST_EU = 10;                        // unrolled head steps of a tile (epilogue slices in the first 8)

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

  // ---- loader: wave w brings rows 8 w + 64 j + (lane >> 3) of A (j < 2) and of W (j < 4), 128 B each ---------------
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
  auto issue_stage = [&](int slot) {
    const uint32_t d = lds0 + (uint32_t)slot * ST_STAGE + (uint32_t)wave * 1024;
    const bf16_t* ab = p.A + dk * 64;
    const bf16_t* wb = p.W + dk * 64;
    const int ar0 = dt.m0 + 8 * wave, wr0 = dt.n0 + 8 * wave;
    if (dt.m0 + ST_BM <= p.M && dt.n0 + ST_BN <= p.N) {
#pragma unroll
      for (int j = 0; j < 2; ++j) dma16(ab + (int64_t)(ar0 + 64 * j) * p.lda, a_voff, d + j * 8192);
#pragma unroll
      for (int j = 0; j < 4; ++j) dma16(wb + (int64_t)(wr0 + 64 * j) * p.ldw, w_voff, d + ST_WOFF + j * 8192);
    } else {
      // edge tile: rows past M / N are clamped (their products are never stored): the piece's base row and the lane's
      // row inside the piece
      int l2 = lane;
      asm volatile("" : "+v"(l2));  // (keeps these address terms out of the registers that live across the loop)
      const int e3 = l2 >> 3, ec = l2 & 7;
      const int a_ch = (ec ^ ((4 * wave + (e3 >> 1)) & 7)) << 4, w_ch = (ec ^ (((e3 >> 1) & 1) | ((wave & 3) << 1))) << 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r0 = ar0 + 64 * j, left = p.M - r0;  // rows of the piece that exist
        const int lim = left >= 8 ? 7 : (left > 0 ? left - 1 : 0);
        dma16(ab + (int64_t)(r0 < p.M ? r0 : p.M - 1) * p.lda, (uint32_t)(min(e3, lim) * p.lda * 2 + a_ch), d + j * 8192);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r0 = wr0 + 64 * j, left = p.N - r0;
        const int lim = left >= 8 ? 7 : (left > 0 ? left - 1 : 0);
        dma16(wb + (int64_t)(r0 < p.N ? r0 : p.N - 1) * p.ldw, (uint32_t)(min(e3, lim) * p.ldw * 2 + w_ch), d + ST_WOFF + j * 8192);
      }
    }
    if (++dk == nk) {
      dk = 0;
      crossed_into_new = dn.valid;
      if (dn.valid) dt = dn;
    }
  };
  // per-tile constants -> LDS (2 operations per wave): waves 0-3 bias, 4-7 colsum (or gamma), then row statistics
  auto issue_consts = [&](const StTile& t, int par) {
    const uint32_t d = lds0 + ST_CONST + (uint32_t)par * 4096 + (uint32_t)wave * 256;
    int col = t.n0 + (wave & 3) * 64 + lane;
    const bool cok = col < p.N;
    col = cok ? col : 0;
    const float* src;
    if (wave < 4) {
      src = p.bias && cok ? p.bias + col : g_stream_zero + lane;
    } else if (EPI == EPI_BIAS_RESID) {
      src = p.gamma && cok ? p.gamma + col : g_stream_one + lane;
    } else {
      src = FOLD && cok ? p.colsum + col : g_stream_zero + lane;
    }
    dma4v(src, d);
    // (mean, rstd) of rows m0 + 32 (wave & 3) + lane / 2 (waves 4-7 write a duplicate behind it)
    int row = t.m0 + (wave & 3) * 32 + (lane >> 1);
    row = row < p.M ? row : p.M - 1;
    const float* ssrc = FOLD ? p.ln_stats + 2 * (int64_t)row + (lane & 1) : g_stream_zero + lane;
    dma4v(ssrc, d + 2048);
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

  // one epilogue slice: accumulator tile (it, jt) of `prev`; pk carries the packed pair until its store
  uint32_t pk[2];
  auto epi_slice = [&](auto it_tag, auto jt_tag) {
    constexpr int it = decltype(it_tag)::value, jt = decltype(jt_tag)::value;
    const uint32_t cst = lds0 + ST_CONST + (uint32_t)epar * 4096;
    const int colq = wn * 64 + 32 * (jt >> 1) + 8 * q4 + 4 * (jt & 1);
    const f32x4 b4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(cst + colq * 4));
    float o[4];
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
      if constexpr (FOLD) {
        const f32x4 c4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(cst + 1024 + colq * 4));
        const f32x2 st = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((uintptr_t)(cst + 2048 + (wm * 64 + it * 16 + r15) * 8));
        const float rs = st[1], nrm = -st[1] * st[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(rs, prev[jt][it][e], fmaf(nrm, c4[e], b4[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = prev[jt][it][e] + b4[e];
      }
      if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_erf(o[e]);
      }
    }
    bf16x4 ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) ob[e] = (bf16_t)o[e];
    const u32x2 w2 = __builtin_bit_cast(u32x2, ob);
    if constexpr ((jt & 1) == 0) {
      pk[0] = w2[0];
      pk[1] = w2[1];
      asm volatile("" : "+v"(pk[0]), "+v"(pk[1]));  // (computed HERE, under this half's MFMAs: hipcc sinks it to the store otherwise)
    } else {
      u32x4 v;
      v[0] = pk[0];
      v[1] = pk[1];
      v[2] = w2[0];
      v[3] = w2[1];
      const uint32_t vo = c_voff;
      const uint32_t so = (uint32_t)(it * 16) * ldc2 + (uint32_t)((jt >> 1) * 64);
      // (row step in the VGPR offset: the hardware's range check covers voffset + immediate, rows past M are dropped)
      const uint32_t voff = vo + so;  // (a masked lane's 0x7fffffff stays out of range: so < 2^31)
      __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, voff, 0, NT ? 2 : 0);
    }
  };

  // ---- prologue: stages 0..2 of the first tile ---------------------------------------------------------------------
  issue_stage(0);
  issue_stage(1);
  issue_stage(2);
  StTile ct = dt;  // compute tile == first tile (nk >= ST_EU + 1 > 3: the loader has not left it)
  int cpar = 0;
  wait_vmcnt<12>();  // stage 0 landed (this wave's pieces)
  asm volatile("s_barrier" ::: "memory");

  bf16x8 fa[2][4], fb[2][4];
  int slot = 0;
  auto rd_frags = [&](int sl, int h, bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    const uint32_t base = lds0 + (uint32_t)sl * ST_STAGE;
    // (second 32-deep half of the 128-B rows: chunk + 4 = byte offset ^ 64, the images are 128-B aligned)
    const uint32_t aa = base + (a_rd0 ^ (uint32_t)(h << 6));
    const uint32_t bb = base + (b_rd0 ^ (uint32_t)(h << 6));
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = lds_rd(bb + JOFF[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = lds_rd(aa + i * 2048);
  };
  rd_frags(0, 0, fa[0], fb[0]);
  wait_vmcnt<6>();  // stage 1 landed: the first step reads its first half before the first mid-step barrier
  asm volatile("s_barrier" ::: "memory");

  // One 64-deep step.  MODE 0: body, 1: first step of a tile (accumulators start from zero), 2: last step of a tile.
  // QA / QB: epilogue slices of the two halves (-1: none).  NV: vmcnt of the mid-step wait.  CONSTS: issue the tile's
  // constants behind the stage.
  auto step = [&](auto mode_tag, auto qa_tag, auto qb_tag, auto nv_tag, auto consts_tag) {
    constexpr int MODE = decltype(mode_tag)::value, QA = decltype(qa_tag)::value, QB = decltype(qb_tag)::value;
    constexpr int NV = decltype(nv_tag)::value;
    constexpr bool CONSTS = decltype(consts_tag)::value;
    const int nslot = slot == 2 ? 0 : slot + 1;
    // ---- half 0: MFMAs on (fa[0], fb[0]); fragments of half 1 come in underneath
    rd_frags(slot, 1, fa[1], fb[1]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (MODE == 1) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], acc[j][i], 0, 0, 0);
      }
    if constexpr (QA >= 0) epi_slice(std::integral_constant<int, (QA >> 2)>{}, std::integral_constant<int, (QA & 3)>{});
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < 4 || (i >= 8 && !(i & 1))) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (QA >= 0) __builtin_amdgcn_sched_group_barrier(0x002, EPI == EPI_BIAS_GELU ? 4 : 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- middle: this wave has read stage `slot` completely; its pieces of the next stage have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<NV>();
    asm volatile("s_barrier" ::: "memory");
    issue_stage(slot);
    if constexpr (CONSTS) issue_consts(ct, cpar);
    __builtin_amdgcn_sched_barrier(0);
    // ---- half 1: MFMAs on (fa[1], fb[1]); first half of the next step's fragments underneath
    rd_frags(nslot, 0, fa[0], fb[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (MODE == 2) prev[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[j][i], 0, 0, 0);
        else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[j][i], 0, 0, 0);
      }
    if constexpr (QB >= 0) epi_slice(std::integral_constant<int, (QB >> 2)>{}, std::integral_constant<int, (QB & 3)>{});
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < 4 || (i >= 8 && !(i & 1))) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (QB >= 0) __builtin_amdgcn_sched_group_barrier(0x002, EPI == EPI_BIAS_GELU ? 4 : 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  };
#define ST_I(x) std::integral_constant<int, (x)> {}
#define ST_B(x) std::integral_constant<bool, (x)> {}

  // vmcnt of the mid-step wait of tile step s = operations issued behind the stage it retires (issued at step s - 2):
  // stores of steps s-2 and s-1 (one per step 0..7), the stage of step s-1 (6), the constants of step 0 (2)
  for (;;) {
    step(ST_I(1), ST_I(0), ST_I(1), ST_I(6), ST_B(true));
    step(ST_I(0), ST_I(2), ST_I(3), ST_I(9), ST_B(false));
    step(ST_I(0), ST_I(4), ST_I(5), ST_I(10), ST_B(false));
    step(ST_I(0), ST_I(6), ST_I(7), ST_I(8), ST_B(false));
    step(ST_I(0), ST_I(8), ST_I(9), ST_I(8), ST_B(false));
    step(ST_I(0), ST_I(10), ST_I(11), ST_I(8), ST_B(false));
    step(ST_I(0), ST_I(12), ST_I(13), ST_I(8), ST_B(false));
    step(ST_I(0), ST_I(14), ST_I(15), ST_I(8), ST_B(false));
    step(ST_I(0), ST_I(-1), ST_I(-1), ST_I(8), ST_B(false));
    step(ST_I(0), ST_I(-1), ST_I(-1), ST_I(7), ST_B(false));
    for (int s = ST_EU; s < nk - 1; ++s) step(ST_I(0), ST_I(-1), ST_I(-1), ST_I(6), ST_B(false));
    step(ST_I(2), ST_I(-1), ST_I(-1), ST_I(6), ST_B(false));
    // tile boundary: `prev` holds tile ct
    et = ct;
    epar = cpar;
    epi_setup();
    if (!crossed_into_new) break;  // the loader never left this tile: it was the last
    ct = dt;
    cpar ^= 1;
    dn = fetch_tile();
  }
  // ---- flush: epilogue of the last tile, nothing to hide it under ---------------------------------------------------
  wait_vmcnt<0>();
  asm volatile("s_barrier" ::: "memory");
#define ST_Q(q) epi_slice(ST_I((q) >> 2), ST_I((q) & 3));
  ST_Q(0) ST_Q(1) ST_Q(2) ST_Q(3) ST_Q(4) ST_Q(5) ST_Q(6) ST_Q(7)
  ST_Q(8) ST_Q(9) ST_Q(10) ST_Q(11) ST_Q(12) ST_Q(13) ST_Q(14) ST_Q(15)
#undef ST_Q
#undef ST_I
#undef ST_B
}

hipError_t launch_gemm_stream(const GemmArgs& a, int epi, hipStream_t s) {
  if (a.K % 64 || a.K / 64 < ST_EU + 1 || a.N % 64 || a.M <= 0) return hipErrorInvalidValue;
  if (a.w_interleaved || a.out_f32 || a.win_ws || a.a_rpg || a.patch_p || a.ln_cpart || a.ln_part) return hipErrorInvalidValue;
  if (epi != EPI_BIAS && epi != EPI_BIAS_GELU) return hipErrorInvalidValue;
  if (a.M >= (1 << 30) || a.ldc >= (1 << 24) || a.lda >= (1 << 24) || a.ldw >= (1 << 24)) return hipErrorInvalidValue;
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

  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  static int n_cu[64] = {};
  if (!n_cu[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount;
  }
  const int grid = k.ntiles < n_cu[dev] ? k.ntiles : n_cu[dev];
#define ST_LAUNCH2(E, F, NTV)                                                                                            \
  {                                                                                                                 \
    static bool attr[64] = {};                                                                                      \
    if (!attr[dev]) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_stream_kernel<E, F, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS); \
      if (e != hipSuccess) return e;                                                                                \
      attr[dev] = true;                                                                                             \
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
