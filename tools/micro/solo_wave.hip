// Can ONE wave per SIMD keep the gfx950 matrix pipe busy from a hipcc-scheduled loop when its tile is 128 x 128?
// (hipBLASLt's kernel for the ViT-B shapes is MT256x256x64, 4 waves, one workgroup per CU: tools/blaslt_names.py.)
// One 4-wave workgroup per CU, 2 x 2 waves of 128 x 128 over a 256 x 256 tile; per 64-deep step a wave does 128 MFMAs
// (16x16x32) = 2048 pipe cycles, 32 ds_read_b128 of fragments and, optionally, its 16 of the step's 64 LDS-DMA pieces and
// one s_barrier.  Nothing is computed that means anything: LDS holds whatever it holds, the loads go to the other stage.
//   hipcc --offload-arch=gfx950 -O3 -o solo_wave solo_wave.hip && ./solo_wave <loads 0|1|2> <barrier 0|1> [steps] [wait 0|1|2]   (loads 2 / 3: global_load_dwordx4 -> VGPR -> ds_write_b128 one step later, as C++ / as opaque instructions)
// Prints cycles per step per wave (s_memtime) against the 2048 of the pipe.
// (loads 2 -- register-staged as C++ -- is NOT a measurement of that path: with 256 accumulator registers in AGPRs hipcc keeps
// the 16 staged registers there too and rotates them through v_accvgpr_mov / read / write chains in the loop: 0.7 PF.
// loads 3 issues the same loads and ds_writes as opaque instructions with a counted vmcnt(15): the staged data stays put,
// but with 64 more live VGPRs hipcc renames accumulator tuples through 64 v_accvgpr_mov + s_nop per step; 1.24-1.34 PF
// without and 0.96-1.17 PF with the barrier -- below the LDS-DMA form (1.63-1.72 / 1.26-1.42) even allowing for that.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int STAGE = 512 * 128;  // [256 rows A | 256 rows W] x 128 B

__device__ __forceinline__ void dma_piece(const void* g, unsigned lds_addr) {
  // 64 lanes x 16 B -> 1 KB of LDS at M0 (wave-uniform), lane l lands at +16 l
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory");
}

template <int LOADS, bool BARRIER, int WAIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const char* __restrict__ src, long src_bytes,
                                                                                    int steps, float* sink,
                                                                                    unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;
  // fragment address of (row tile i, k half kk): row 16 i + r of the wave's 128 rows, chunk (4 kk + q) ^ (row & 7)
  const unsigned a_base = (unsigned)(uintptr_t)lds + (128 * wm + r) * 128;
  const unsigned w_base = (unsigned)(uintptr_t)lds + (256 + 128 * wn + r) * 128;
  const unsigned sw0 = (unsigned)(((0 + q) ^ (r & 7)) * 16), sw1 = (unsigned)(((4 + q) ^ (r & 7)) * 16);
  f32x4 acc[8][8];
  static_for<64>([&](auto m) { acc[m / 8][m % 8] = f32x4{0.f, 0.f, 0.f, 0.f}; });
  bf16x8 fa[2][8], fb[2][8];
  auto read_frag = [&](auto set, auto idx, unsigned stage_off, int kk) {
    // idx 0..7: A row tile idx; 8..15: W row tile idx - 8
    constexpr int t = idx % 8;
    const unsigned base = (idx < 8 ? a_base : w_base) + stage_off + t * 16 * 128 + (kk ? sw1 : sw0);
    bf16x8 v = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)base);
    if constexpr (idx < 8) fa[set][t] = v;
    else fb[set][t] = v;
  };
  const char* g = src + ((long)blockIdx.x * 65536 + wave * 16384 + lane * 16) % (src_bytes - (1 << 20));
  i32x4 stg[16];
  if (LOADS == 2) static_for<16>([&](auto i) { stg[i] = *reinterpret_cast<const i32x4*>(g + i * 1024); });
  if (LOADS == 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[i]) : "v"(g + i * 1024) : "memory");
  }
  static_for<16>([&](auto i) { read_frag(std::integral_constant<int, 0>{}, i, 0u, 0); });
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    const unsigned cur = (s & 1) ? STAGE : 0, nxt = (s & 1) ? 0 : STAGE;
    static_for<2>([&](auto kk) {
      constexpr int set = kk, oset = 1 - kk;
      static_for<16>([&](auto grp) {
        // the fragment set of the next k half (the next step's first half from the other stage when kk == 1)
        read_frag(std::integral_constant<int, oset>{}, grp, kk == 0 ? cur : nxt, kk == 0 ? 1 : 0);
        if constexpr (LOADS == 1 && (grp % 2 == 0)) {
          constexpr int piece = kk * 8 + grp / 2;  // 16 pieces per wave per step
          dma_piece(g + piece * 1024, (unsigned)(uintptr_t)lds + nxt + (wave * 16 + piece) * 1024);
        }
        if constexpr (LOADS == 3 && (grp % 2 == 0)) {
          // register-staged, every memory instruction opaque: hipcc cannot move the staged data anywhere.  The piece loaded
          // one step ago is the oldest of the 16 loads in flight: vmcnt(15)
          constexpr int piece = kk * 8 + grp / 2;
          const unsigned dst = (unsigned)(uintptr_t)lds + nxt + (wave * 16 + piece) * 1024 + lane * 16;
          asm volatile("s_waitcnt vmcnt(15)\n\tds_write_b128 %0, %1" : : "v"(dst), "v"(stg[piece]) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[piece]) : "v"(g + piece * 1024) : "memory");
        }
        if constexpr (LOADS == 2 && (grp % 2 == 0)) {
          // register-staged: the piece loaded ONE step ago goes to LDS (ds_write_b128), its register takes the next load
          constexpr int piece = kk * 8 + grp / 2;
          *reinterpret_cast<__attribute__((address_space(3))) i32x4*>((uintptr_t)((unsigned)(uintptr_t)lds + nxt + (wave * 16 + piece) * 1024 + lane * 16)) = stg[piece];
          stg[piece] = *reinterpret_cast<const i32x4*>(g + piece * 1024);
        }
        static_for<4>([&](auto e) {
          constexpr int m = grp * 4 + e, i = m / 8, j = m % 8;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if (LOADS) {
      g += 16384 * 4;
      if (g + 65536 > src + src_bytes) g -= (src_bytes - (2 << 20));
      if (LOADS == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the ds_writes are opaque: hipcc does not wait for them)
      // WAIT 0: everything issued in this step has landed (a 2-stage ring with no slack); 1: everything issued in the
      // PREVIOUS step has (the 16 pieces of this step may be in flight: one step of slack); 2: no wait (the address path alone)
      if (LOADS == 1 && WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (LOADS == 1 && WAIT == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    if (BARRIER) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  static_for<64>([&](auto m) { sum += acc[m / 8][m % 8][0] + acc[m / 8][m % 8][3]; });
  if (LOADS >= 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<16>([&](auto i) { sum += (float)stg[i][0]; });
  }
  if (sum == 123456.789f) sink[lane] = sum;
  if (lane == 0 && blockIdx.x == 7) cyc[wave] = t1 - t0;
}

int main(int argc, char** argv) {
  const int loads = argc > 1 ? atoi(argv[1]) : 0, barrier = argc > 2 ? atoi(argv[2]) : 0, steps = argc > 3 ? atoi(argv[3]) : 2000;
  const long src_bytes = 512l << 20;
  char* src;
  float* sink;
  unsigned long long* cyc;
  CK(hipMalloc(&src, src_bytes));
  CK(hipMemset(src, 0, src_bytes));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&cyc, 64));
  auto run = [&](auto kern) -> int {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 2 * STAGE, 0, src, src_bytes, steps, sink, cyc);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[4];
      CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
      const double tf = 256.0 * 4 * steps * 128 * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
      printf("loads %d barrier %d wait %d: %.1f us, %.0f TFLOP/s; cycles per step per wave:", loads, barrier, argc > 4 ? atoi(argv[4]) : 0, ms * 1e3, tf);
      for (int w = 0; w < 4; ++w) printf(" %.0f", (double)h[w] / steps);
      printf("  (pipe: 2048)\n");
    }
    return 0;
  };
  const int wait = argc > 4 ? atoi(argv[4]) : 0;
  if (loads == 3) return barrier ? run(k<3, true, 2>) : run(k<3, false, 2>);  // register-staged, opaque instructions
  if (loads == 2) return barrier ? run(k<2, true, 2>) : run(k<2, false, 2>);  // register-staged loads (hipcc counts the waits itself)
  if (loads && barrier) return wait == 0 ? run(k<1, true, 0>) : wait == 1 ? run(k<1, true, 1>) : run(k<1, true, 2>);
  if (loads) return wait == 0 ? run(k<1, false, 0>) : wait == 1 ? run(k<1, false, 1>) : run(k<1, false, 2>);
  if (barrier) return run(k<0, true, 0>);
  return run(k<0, false, 0>);
}
