// EXPERIMENT, not built into the library (round 3).  To try it again: copy to csrc/, add it to the Makefile's SRCS, give
// launch_cfg a PIPE == 51 hook that calls launch_ring4d before its launch switch, and map a variant number to
// launch_cfg<2, 4, 51>.
// Result on one MI355X, M = 50432, interleaved with ring4 (variant 26, packed weights), bitwise equal to ring4 on every case of
// tools/stream_check.py and, as the fc1 of the whole forward, on tools/ab_stream_forward.py (LayerNorm fold on and off):
//     every tile drawn (first form):   qkv 176.4 -> 180.3 us    fc1 (erf-GELU) 254.4 -> 257.7    N 768 / K 3072 222.2 -> 223.1
//     static head + dynamic tail:      qkv 164.8 -> 171.7       fc1 245.8 -> 234.8 (-4.5 %)       N 768 / K 3072 213.0 -> 213.6
//     in the forward (fc1 only, with the LayerNorm fold: 12 more constants per lane, the second column pair's requested
//     behind the fill): gemm_ring4d_kernel<1, true> 250 us per launch against ring4's 249-258; step 9.048 -> 9.019 ms.
// i.e. hiding the prologue behind the epilogue pays where the epilogue is long (GELU) and the output not the larger part of
// the traffic; for qkv the 16-row x 64-B store instructions in place of whole 128-B lines cost more than the prologue (the
// CU's address path is what both kernels wait on: DESIGN.md 4.7), and with the fold's constants the fc1 gain is gone too.
// Findings worth keeping:
//   * A HARDWARE HAZARD hipcc (ROCm 7.2) does not cover: `buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen` -- an SGPR
//     soffset -- followed IMMEDIATELY by a VALU write of v[a] can store the NEW value of that register (seen in lanes 12-15 of
//     every 16, in the second wave of a SIMD, in ~1 % of the tiles of the K = 3072 launch: the first dword of the 16-byte
//     store held the next block's unconverted fp32 sum).  LLVM's hazard recognizer inserts the wait state only when the
//     soffset is NOT a register.  One wait state is enough (a build with an s_and_b64 between the two: 40 clean launches;
//     with nothing between them ~500 wrong elements per launch).  R4D_STORE_NOP below ties an s_nop to the store's data
//     registers so that the scheduler cannot move the write in front of it; tools/hazard_scan.py looks for the
//     pattern in the library's ISA (none: the shipped stores with an SGPR soffset are followed by other stores or the end).
//   * with the draw as an opaque `global_atomic_add v127 ... sc0` (v127 kept out of hipcc's hands by amdgpu_num_vgpr(127))
//     the kernel raised a memory access fault on its first multi-tile launch (gpurun_out/r3/r4d_{2,3,8}.log); with atomicAdd
//     it does not.  CAUSE (round 4, from the ISA of that build -- commit 6cc6aa2 with R4D_BUILTIN_DRAW undefined, hipcc -S --
//     not from re-running it): NOT v127 -- the kernel descriptors read .amdhsa_next_free_vgpr 128, .amdhsa_accum_offset 128,
//     .agpr_count 0, so v127 is the last architected VGPR of the wave's allocation.  It is the atomic's BASE ADDRESS.  At its
//     128-register budget the persistent loop spills SGPRs to lanes of v126, and the in-loop draw compiles to
//         v_readlane_b32 s78, v126, 4 / v_readlane_b32 s79, v126, 5 / v_mov_b32 v0, s30 / v_mov_b32 v102, 1
//         ;;#ASMSTART  global_atomic_add v127, v0, v102, s[78:79] sc0  ;;#ASMEND
//     `VALU writes an SGPR -> a vector-memory instruction reads that SGPR` needs 5 wait states on gfx9 / CDNA and hipcc's
//     hazard recognizer does not look INSIDE an asm statement: with 2-3 wait states the atomic went to whatever s[78:79]
//     held before (scalar temporaries of the tile-index divisions) -- a wild address, on the first launch whose loop runs
//     (the one-tile integer cases only execute the prologue draw, whose pointer pair comes straight from s_load).  Wait states
//     BEHIND the instruction, which is what was tried, cannot help; `s_nop 4` as the first line of the asm statement (or the
//     pointer handed over as a VGPR pair) does.  An "=v" output instead of v127 would not have.  tools/hazard_scan.py now
//     flags the pattern (6 of the 8 asm atomics of that build) and tests/test_abi_cpu.py keeps the library's own asm memory
//     instructions (the opaque LDS-DMA of the attention kernels: 180, none behind a VALU write of their SGPRs) clean.
//     The builtin costs thread 0's wave a vmcnt(0) right behind the draw (hipcc's atomic optimizer reads the result back at
//     once), i.e. that wave's fill is not overlapped;
//   * the launcher once asked for 80 KB + 16 B of LDS: one workgroup per CU instead of two, 215 / 306 / 260 us.
//
// ring4d: the ring4 main loop (gemm_kernels.h: 128 x 256 tile, 8 waves, two workgroups per CU) as PERSISTENT workgroups
// whose next tile's ring fill runs under the current tile's epilogue, for the write-once linears (EPI_BIAS = attn.qkv,
// EPI_BIAS_GELU = mlp.fc1; reference src/models_archs.py:130-135, the nn.Linear calls of the frozen ViTs reached from
// src/tfds_dense_descriptor.py:123).  Variant 31 of vdr_op_linear, vdr_config.stream_gemm = 2.
//
// What the round-2 stamps of ring4 left on the table (profiles/r02_gemm_stamps.txt): 8-9 % of a tile's life is the
// prologue (addresses, ring fill, the first wait), and a workgroup slot is occupied only 81-85 % of a launch (relaunch into
// a drained slot).  The ring cannot be refilled early there because the epilogue stages its outputs through the ring's LDS.
// Here:
//   * the outputs never touch LDS: the W rows of a wave tile are assigned to MFMA row slots by the permutation the stream
//     kernel introduced (gemm_stream.hip: slot i of column tile jt holds column 32 (jt >> 1) + 8 (i >> 2) + 4 (jt & 1) + (i & 3)),
//     under which a lane's accumulators of tiles (jt, jt + 1) are 8 consecutive columns of one row -- 16-byte buffer stores
//     straight from the accumulator layout (16 rows x 64 B per instruction).  Only the fragment read address and the
//     chunk swizzle of the W image change (chunk ^ -(row >> 3) & 3: the 16-lane groups a ds_read_b128 is served in now
//     hold rows 8a + b instead of 4a + b);
//   * so the barrier that ends a tile's main loop frees the whole ring: the next tile's [A0 W0] [W1] [A1 W2] fill is issued
//     right behind it and lands while this tile's epilogue (LayerNorm fold, bias, erf-GELU, bf16, 8 stores per wave) runs;
//     the next main loop starts with one counted vmcnt that leaves the 8 stores in flight;
//   * the loads are the LDS-DMA builtin and the stores buffer-store builtins, so hipcc counts every vector-memory
//     operation itself (its wait for the epilogue constants leaves the fill in flight) and the written vmcnt values are exact;
//   * tiles are handed out dynamically: the grid is the number of workgroups the chip holds at once; workgroup b starts
//     with tile id b and then draws ids from one counter per XCD (ids congruent to b mod 8, which xcd_remap turns into a
//     contiguous tile range per XCD exactly as for the non-persistent kernels), one atomic per tile, fetched a whole tile
//     ahead.  The draw that returns the last value of its XCD resets the counter: no memset between launches.  (Static
//     striding lost 13 % on qkv in round 2: the hardware dispatcher's back-fill is a dynamic queue, and this is one.)
// Same products in the same order and the same epilogue formulas as ring4: outputs are bitwise equal (tests/test_ops_gpu.py).
#define R4D_STORE_NOP 1
#include <mutex>

#include "gemm_kernels.h"

namespace vdr {

constexpr int R4D_SLOTS = 64;                    // streams (per device) that can have a ring4d launch in flight
__device__ int g_r4d_queue[R4D_SLOTS][8];        // [stream slot][XCD]: next draw; zero between launches

VDR_DEV int swzd(int row) { return (-(row >> 3)) & 3; }


template <int EPI, bool FOLD>
__global__ __launch_bounds__(512, 4) void gemm_ring4d_kernel(GemmK p, int* __restrict__ queue) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU, "write-once outputs only");
  constexpr int WAVES_N = 4;
  constexpr int BM = 128, BN = 256;
  constexpr int APIECE = BM * 128, WUNIT = BN * 64, WBASE = 2 * APIECE;
  constexpr int NA4 = 2, NB = 2;                 // LDS-DMA pieces per wave per A piece / W unit
  constexpr int NSTORE = 8;                      // output stores per wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the next tile id travels through the A slot that is dead during a tile's last step (two workgroups x 80 KB fill the CU:
  // there is no LDS beyond the ring)

  const int G = __builtin_amdgcn_readfirstlane(gridDim.x);  // a multiple of 8 (launcher)
  const int xcd = __builtin_amdgcn_readfirstlane(blockIdx.x & 7);
  const int nsteps = p.K >> 5;  // >= 8, even (launcher)
  const int bstep = p.w_il ? 64 : 32;

  const bf16_t* a_src[NA4];
  const bf16_t* b_src[NB];
  int64_t m0 = 0;
  int n0 = 0;
  // operand addresses of tile `id` (lane-derived parts rebuilt per tile: kept live across the loop they cost 20 registers
  // at a 128-register budget, as in gemm_ring4p_kernel)
  auto setup = [&](int id) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_of(p, xcd_remap(id, p.nwg), tm, tn);
    m0 = (int64_t)tm * BM;
    n0 = tn * BN;
#pragma unroll
    for (int q = 0; q < NA4; ++q) {
      const int r = (wave * NA4 + q) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int64_t gr = m0 + r;
      gr = gr < p.M ? gr : p.M - 1;
      a_src[q] = p.A + gr * p.lda + c * 8;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int r = (wave * NB + q) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ swzd(r);
      int gr = n0 + r;
      gr = gr < p.N ? gr : p.N - 1;
      b_src[q] = w_unit_src(p, gr, c);
    }
  };
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto stage_a = [&](int aslot) {
    char* d = smem + aslot * APIECE;
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
  auto fill = [&]() {
    stage_a(0);
    stage_w(0);
    stage_w(1);
    stage_a(1);
    stage_w(2);
  };
  // Tile sequence of workgroup b: a STATIC head, ids b, b + G, ..., b + (n_static - 1) G (every workgroup can work that out
  // for itself: no draw, no mailbox, one barrier between tiles), then a DYNAMIC tail drawn from this XCD's counter over the
  // ids from n_static G on (what evens out the end of the launch: static striding alone lost 13 % on qkv in round 2).  A
  // draw goes through atomicAdd, behind which hipcc's atomic optimizer waits vmcnt(0) -- thread 0's wave then sits out its
  // own fill -- so it is confined to the last tiles of a workgroup.
  const int n_static = max(1, p.nwg / G - 1);
  const int dyn_base = n_static * (G >> 3);                   // draws number the ids xcd + 8 (dyn_base + v)
  const int ids_here = (p.nwg - xcd + 7) >> 3;                // tile ids congruent to xcd (mod 8)
  const int last_v = ids_here - dyn_base + (G >> 3) - 1;      // the last value this XCD's counter ever hands out
  int drawn = 0;  // (thread 0 only) the last value drawn: the id of the tile after the next one
  int t = 0;      // index of the current tile in this workgroup's sequence

  int id = blockIdx.x;
  setup(id);
  fill();
  if (1 >= n_static && threadIdx.x == 0) drawn = atomicAdd(queue + xcd, 1);
  bool first = true;
  for (;;) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r15 = lane & 15, q4 = lane >> 4;
    const int a_row = (wm * 64 + r15) * 128;
    const int a_ch0 = a_row + ((q4 ^ ((r15 >> 1) & 7)) * 16);
    const int a_ch1 = a_row + (((4 + q4) ^ ((r15 >> 1) & 7)) * 16);
    // W fragment of MFMA tile j: row slot r15 reads row 32 (j >> 1) + 8 (r15 >> 2) + 4 (j & 1) + (r15 & 3) of the wave's 64
    const int b_base = WBASE + (wn * 64 + 8 * (r15 >> 2) + (r15 & 3)) * 64 + ((q4 ^ ((-(r15 >> 2)) & 3)) * 16);
    auto b_off = [](int j) { return (32 * (j >> 1) + 4 * (j & 1)) * 64; };
    auto ld = [&](int off) { return *reinterpret_cast<const bf16x8*>(smem + off); };

    Acc16 acc;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc.t[j][i][e] = 0.0f;
    bf16x8 fb[4], alo[2], ahi[2];

    // [A0 W0] have landed; [W1] [A1 W2] and (from the second tile on) the previous tile's stores and thread 0's draw may be in flight
    const bool first_tile = first;
    first = false;
    if (first_tile) wait_vmcnt<NB + NA4 + NB>();
    else wait_vmcnt<NB + NA4 + NB + NSTORE>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = ld(b_base + b_off(j));
#pragma unroll
    for (int i = 0; i < 2; ++i) alo[i] = ld(a_ch0 + i * 2048);

    int wslot = 0, aoff = 0;
    auto step = [&](int s, auto odd_tag) {
      constexpr bool ODD = decltype(odd_tag)::value;
      const int a_cur = aoff + (ODD ? a_ch1 : a_ch0);
      const int a_nxt = ODD ? (aoff ^ APIECE) + a_ch0 : aoff + a_ch1;
      const int nwslot = wslot + 1 == 3 ? 0 : wslot + 1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc.t[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], alo[i], acc.t[j][i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) ahi[i] = ld(a_cur + (2 + i) * 2048);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      if (s + 1 < nsteps) {
        // counts as in ring4: everything but what step s-1 issued has landed.  The previous tile's 8 stores sit in the queue
        // between the fill and the pieces issued from step 0 on: steps 0 and 1 wait for fill pieces only and let them pass
        if (s + 2 < nsteps) {
          if constexpr (ODD) {
            if (s == 1 && !first_tile) wait_vmcnt<NB + NSTORE>();
            else wait_vmcnt<NB>();
          } else {
            if (s == 0 && !first_tile) wait_vmcnt<NB + NA4 + NSTORE>();
            else wait_vmcnt<NB + NA4>();
          }
        } else {
          wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (s + 3 < nsteps) {
          stage_w(wslot);
          if constexpr (ODD) stage_a(aoff ? 1 : 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) alo[i] = ld(a_nxt + i * 2048);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc.t[j][2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], ahi[i], acc.t[j][2 + i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fb[j] = ld(b_base + nwslot * WUNIT + b_off(j));
        __builtin_amdgcn_sched_barrier(0);
      }
      wslot = nwslot;
      if constexpr (ODD) aoff ^= APIECE;
    };
    for (int s = 0; s < nsteps; s += 2) {
      step(s, std::false_type{});
      step(s + 1, std::true_type{});
    }

    int r15e, q4e;
    // ---- the tile's constants, requested into the registers the fragments have left (nothing else is in flight: the last
    //      step waited vmcnt(0)); used after the fill of the next tile has been issued behind them -------------------------
    const int64_t mw = m0 + wm * 64;
    const int nw = n0 + wn * 64;
    {
      // (opaque lane: the loads depend on nothing the main loop computes, hipcc would issue them ahead of it and keep 40
      // registers live across it)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e)::"memory");
      r15e = lane_e & 15;
      q4e = lane_e >> 4;
    }
    // LATE (erf-GELU): the constants of the second column pair are requested behind the fill and used half an epilogue
    // (about 2 us) later -- with all 40 live next to the accumulators and the next tile's addresses the kernel spills
    constexpr bool LATE = EPI == EPI_BIAS_GELU;
    f32x4 bias[4], csum[4];
    float2 stats[4];
    auto load_consts = [&](int jp) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        int n = nw + 32 * jp + 8 * q4e + 4 * h;
        n = n < p.N ? n : 0;  // (columns past N are never stored)
        bias[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (FOLD) csum[j] = *reinterpret_cast<const f32x4*>(p.colsum + n);
      }
    };
    load_consts(0);
    if (!LATE) load_consts(1);
    if (FOLD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t m = mw + i * 16 + r15e;
        m = m < p.M ? m : p.M - 1;
        stats[i] = *reinterpret_cast<const float2*>(p.ln_stats + 2 * m);
      }
    }
    const bool next_static = t + 1 < n_static;  // wave-uniform
    int nid;
    const int last_draw = drawn;
    if (next_static) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done with the ring
      asm volatile("" ::: "memory");
      nid = blockIdx.x + (t + 1) * G;
    } else {
      // the id of the next tile (drawn a tile ago) goes to every wave through LDS: written into the A slot nobody reads any
      // more (every wave is past the barrier of step nsteps - 2), read behind the barrier that frees the ring, and a second
      // barrier keeps a fast wave's fill from overwriting it before a slow wave has read it
      // (an LDS-typed pointer: through a generic `volatile int*` hipcc emitted flat_store / flat_load with vmcnt(0) waits)
      typedef __attribute__((address_space(3))) int lds_int;
      lds_int* s_next = (lds_int*)(smem + ((((nsteps >> 1) - 1) & 1) ^ 1) * APIECE);
      if (threadIdx.x == 0) *s_next = xcd + 8 * (dyn_base + drawn);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nid_v = *s_next;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nid_v) : : "memory");
      nid = __builtin_amdgcn_readfirstlane(nid_v);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const bool more = (uint32_t)nid < (uint32_t)p.nwg;  // wave-uniform (unsigned: whatever the mailbox held, never an id outside the tile list)
    const int64_t mw_this = mw;
    if (more) {
      setup(nid);
      fill();
      if (t + 2 >= n_static && threadIdx.x == 0) drawn = atomicAdd(queue + xcd, 1);  // the tile after the next one is a dynamic one
    } else if (threadIdx.x == 0 && last_draw == last_v) {
      // this workgroup's draw failed; it was the last draw its XCD will ever make: the counter goes back to zero
      atomicExch(queue + xcd, 0);
    }
    if (LATE) load_consts(1);

    // ---- epilogue in the accumulator layout: lane (r15, q4) owns rows 16 i + r15 and, for tile pair jp, columns
    //      32 jp + 8 q4 .. + 7 of the wave tile (same formulas as epilogue_bf16) ---------------------------------------
    {
      int64_t mb;
      {
        const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)mw_this);
        const int hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)mw_this >> 32));
        mb = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
      }
      const int nb = __builtin_amdgcn_readfirstlane(nw);
      const int64_t left = p.M - mb;
      const int valid = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
      const uint32_t ldc2 = (uint32_t)p.ldc * 2u;
      const __amdgpu_buffer_rsrc_t rc =
          __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + mb * p.ldc + nb), 0, (int)((uint32_t)valid * ldc2), 0x00020000);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int col = 32 * jp + 8 * q4e;
        const uint32_t voff = nb + col < p.N ? (uint32_t)r15e * ldc2 + (uint32_t)col * 2u : 0x7fffffffu;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mu = FOLD ? stats[i].x : 0.0f, rs = FOLD ? stats[i].y : 1.0f;
          const float nrm = -rs * mu;
          bf16x8 o;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * jp + h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc.t[j][i][e];
              v = fmaf(rs, v, fmaf(nrm, FOLD ? csum[j][e] : 0.0f, bias[j][e]));
              if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
              o[4 * h + e] = (bf16_t)v;
            }
          }
          if (p.nt_store) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, voff, (uint32_t)(i * 16) * ldc2, 2 /* nt */);
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, voff, (uint32_t)(i * 16) * ldc2, 0);
#ifdef R4D_STORE_NOP
          asm volatile("s_nop 1" : "+v"(o) : : "memory");  // hazard test: VALU write of the store's first data register right behind a store with an SGPR soffset
#endif

        }
        if (jp == 0) __builtin_amdgcn_sched_barrier(0);  // (the second pair's arithmetic stays behind the first pair's stores)
      }
    }
    if (!more) break;
    id = nid;
    ++t;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
bool ring4d_eligible(const GemmK& k, int epi, int slots) {
  if (epi != EPI_BIAS && epi != EPI_BIAS_GELU) return false;
  if (k.out_f32 || k.win_ws || k.ln_part || k.ln_cpart || k.off || (int64_t)k.rpg < k.M || k.a_rpg || k.pg_ps || k.sA || k.sC || !k.bias) return false;
  if (k.ln_fold && (!k.ln_stats || !k.colsum)) return false;
  if (k.K < 256 || (k.K & 63) || (k.N & 7) || k.abl) return false;
  return slots >= 8 && k.nwg >= 2 * slots;  // enough tiles for every resident workgroup to overlap at least one fill
}

// queue row of (device, stream): kernels of one stream run one after another, so they can share a row; 64 streams per device
static int* ring4d_queue_row(int dev, hipStream_t s) {
  static std::mutex mu;
  static hipStream_t owner[VDR_MAX_DEVICES][R4D_SLOTS];
  static int used[VDR_MAX_DEVICES] = {};
  static int* base[VDR_MAX_DEVICES] = {};
  std::lock_guard<std::mutex> g(mu);
  if (!base[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_r4d_queue)) != hipSuccess || !ptr) return nullptr;
    base[dev] = (int*)ptr;
  }
  for (int i = 0; i < used[dev]; ++i)
    if (owner[dev][i] == s) return base[dev] + i * 8;
  if (used[dev] == R4D_SLOTS) return nullptr;
  owner[dev][used[dev]] = s;
  return base[dev] + (used[dev]++) * 8;
}

// returns hipErrorNotSupported when the launch is not this kernel's (the caller falls back to ring4)
hipError_t launch_ring4d(const GemmK& k, int epi, hipStream_t s) {
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  const size_t lds = (size_t)2 * 128 * 128 + (size_t)3 * 256 * 64;  // exactly half a CU: the next-tile mailbox lives inside the ring
  const bool fold = k.ln_fold != 0;
  const void* fn = epi == EPI_BIAS ? (fold ? (const void*)gemm_ring4d_kernel<EPI_BIAS, true> : (const void*)gemm_ring4d_kernel<EPI_BIAS, false>)
                                   : (fold ? (const void*)gemm_ring4d_kernel<EPI_BIAS_GELU, true> : (const void*)gemm_ring4d_kernel<EPI_BIAS_GELU, false>);
  const int which = (epi == EPI_BIAS ? 0 : 2) + (fold ? 1 : 0);
  static int slots_dev[VDR_MAX_DEVICES][4] = {};
  int& slots = slots_dev[dev][which];
  if (!slots) {
    int per_cu = 0;
    const int n_cu = device_cu_count(dev);
    if (n_cu <= 0 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, lds) != hipSuccess || per_cu <= 0)
      return hipErrorUnknown;
    slots = (per_cu * n_cu) & ~7;
  }
  if (!ring4d_eligible(k, epi, slots)) return hipErrorNotSupported;
  int* row = ring4d_queue_row(dev, s);
  if (!row) return hipErrorNotSupported;
  const dim3 grid((unsigned)slots), block(512);
  switch (which) {
    case 0: hipLaunchKernelGGL((gemm_ring4d_kernel<EPI_BIAS, false>), grid, block, lds, s, k, row); break;
    case 1: hipLaunchKernelGGL((gemm_ring4d_kernel<EPI_BIAS, true>), grid, block, lds, s, k, row); break;
    case 2: hipLaunchKernelGGL((gemm_ring4d_kernel<EPI_BIAS_GELU, false>), grid, block, lds, s, k, row); break;
    default: hipLaunchKernelGGL((gemm_ring4d_kernel<EPI_BIAS_GELU, true>), grid, block, lds, s, k, row); break;
  }
  return hipGetLastError();
}

}  // namespace vdr
