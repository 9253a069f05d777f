// What does the per-CU L2 -> LDS path deliver for the access SHAPES a GEMM operand ring can use?
// Every workgroup streams K-steps of a [rows][K] bf16 matrix (row stride K*2 bytes) into an LDS ring with
// global_load_lds_dwordx4, a counted vmcnt keeping `ahead` steps in flight, exactly like gemm_ring3_body.
//   shape 0: 16 rows x  64 B per wave-instruction (the 32-deep units of a row-major operand, ring3: half cache lines)
//   shape 1:  8 rows x 128 B per wave-instruction (64-deep units: whole 128-B lines)
//   shape 2:  4 rows x 256 B per wave-instruction (128-deep units)
//   shape 3: 1 KB contiguous (upper bound)
// Workgroups that share blockIdx % 8 (one XCD under round-robin placement) read the same row window, so the source
// is served by that XCD's L2 after the first touch (window: `rows` x K x 2 B, default 1024 x 768 = 1.5 MB per XCD).
// Reports GB/s per CU for 1 or 2 workgroups per CU of 4 / 8 / 16 waves.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/glds_shape.hip -o tools/micro/glds_shape   (the binary is not tracked)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// G wave-instructions per wave and step; the ring holds 3 steps, 2 in flight
template <int SHAPE, int G>
__global__ __launch_bounds__(1024) void glds_kernel(const char* src, int K2 /*row bytes*/, int rows, int iters, int private_window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  constexpr int RPI = SHAPE == 0 ? 16 : SHAPE == 1 ? 8 : SHAPE == 2 ? 4 : 0;   // rows per instruction
  constexpr int SEG = SHAPE == 0 ? 64 : SHAPE == 1 ? 128 : SHAPE == 2 ? 256 : 1024;  // bytes per row segment
  const size_t window = (size_t)rows * K2;
  const char* base = src + (private_window ? (size_t)blockIdx.x : (size_t)(blockIdx.x & 7)) * window;
  const int steps_per_row_sweep = K2 / SEG;       // K-steps until the row block is exhausted
  const int rows_per_step = SHAPE == 3 ? 0 : RPI * G * nw;  // rows one step of the workgroup covers
  int rb = (blockIdx.x >> 3) * 64 % rows;        // different workgroups start at different row blocks
  int ks = 0;
  size_t lin = (size_t)(blockIdx.x >> 3) * 65536 % window;
  auto issue = [&](int slot) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const char* s;
      if (SHAPE == 3) {
        s = base + (lin + (size_t)(wave * G + q) * 1024 + lane * 16) % window;
      } else {
        const int r = (rb + (wave * G + q) * RPI + lane / (SEG / 16)) % rows;
        s = base + (size_t)r * K2 + ks * SEG + (lane % (SEG / 16)) * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(smem + ((slot * nw + wave) * G + q) * 1024), 16, 0, 0);
    }
    if (SHAPE == 3) {
      lin = (lin + (size_t)nw * G * 1024) % window;
    } else if (++ks == steps_per_row_sweep) {
      ks = 0;
      rb = (rb + rows_per_step) % rows;
    }
  };
  issue(0);
  issue(1);
  int slot = 2;
  for (int it = 0; it < iters; ++it) {
    issue(slot);
    slot = slot == 2 ? 0 : slot + 1;
    wait_vmcnt<2 * G>();
    __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
}

template <int SHAPE, int G>
static int run(const char* src, int K2, int rows, int waves, int wg_per_cu, int priv, const char* label) {
  const int iters = 4000;
  const size_t lds = (size_t)3 * waves * G * 1024;
  CK(hipFuncSetAttribute((const void*)glds_kernel<SHAPE, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((glds_kernel<SHAPE, G>), dim3(256 * wg_per_cu), dim3(waves * 64), lds, 0, src, K2, rows, iters, priv);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double bytes = (double)256 * wg_per_cu * (iters + 2) * waves * G * 1024;
  printf("%-28s waves/WG %2d  WG/CU %d  G %d  LDS %3zu KB  window %s : %7.3f ms  %6.2f TB/s  %6.1f GB/s/CU\n", label, waves, wg_per_cu, G,
         lds / 1024, priv ? "private(MALL/HBM)" : "per-XCD(L2)", best, bytes / best / 1e9, bytes / best / 1e6 / 256);
  return 0;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 768;
  const int rows = argc > 2 ? atoi(argv[2]) : 1024;
  const int K2 = K * 2;
  char* src;
  const size_t bytes = (size_t)512 * rows * K2;  // private windows for up to 512 workgroups
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(src, 1, bytes));
  printf("K = %d (row %d B), window %d rows = %.2f MB\n", K, K2, rows, rows * (double)K2 / 1e6);
  for (int priv = 0; priv < 2; ++priv) {
    // one 16-wave workgroup per CU (the 256 x 256 tile: 32 KB per 32-deep unit -> G = 2 per wave)
    run<0, 2>(src, K2, rows, 16, 1, priv, "16r x 64B  (BK32)");
    run<1, 2>(src, K2, rows, 16, 1, priv, " 8r x 128B (BK64)");
    run<2, 2>(src, K2, rows, 16, 1, priv, " 4r x 256B (BK128)");
    run<3, 2>(src, K2, rows, 16, 1, priv, "contiguous 1 KB");
    // two 8-wave workgroups per CU (the 128 x 256 tile: 24 KB per unit -> G = 3 per wave)
    run<0, 3>(src, K2, rows, 8, 2, priv, "16r x 64B  (BK32)");
    run<1, 3>(src, K2, rows, 8, 2, priv, " 8r x 128B (BK64)");
    run<2, 3>(src, K2, rows, 8, 2, priv, " 4r x 256B (BK128)");
    run<3, 3>(src, K2, rows, 8, 2, priv, "contiguous 1 KB");
    // one 8-wave workgroup per CU, deeper per-wave issue (48 KB per step -> G = 6; 3 steps = 144 KB of LDS)
    run<0, 6>(src, K2, rows, 8, 1, priv, "16r x 64B  (BK32)");
    run<1, 6>(src, K2, rows, 8, 1, priv, " 8r x 128B (BK64)");
    run<3, 6>(src, K2, rows, 8, 1, priv, "contiguous 1 KB");
    // one 4-wave workgroup per CU
    run<0, 8>(src, K2, rows, 4, 1, priv, "16r x 64B  (BK32)");
    run<1, 8>(src, K2, rows, 4, 1, priv, " 8r x 128B (BK64)");
    run<3, 8>(src, K2, rows, 4, 1, priv, "contiguous 1 KB");
  }
  return 0;
}
