// Do matrix-pipe cycles and vector-issue cycles overlap on a gfx950 SIMD, or do they add?
// One workgroup per CU.  Roles per wave:
//   M: `iters` rounds of 8 independent v_mfma (32x32x16 bf16: 8 passes, or 16x16x32 bf16: 4 passes), no memory
//   V: `iters` rounds of VPM independent v_fma_f32 (8 chains), no memory
//   X: both in ONE wave, interleaved in program order (8 MFMAs + VPM fmas per round)
// Modes (argv[1]):
//   0  4 waves, all M           (one wave per SIMD: the matrix pipe alone)
//   1  4 waves, all V           (the vector port alone)
//   2  8 waves, 0-3 M and 4-7 V (two waves per SIMD, one of each role)
//   3  8 waves, all M           (two M waves per SIMD)
//   4  8 waves, all V
//   5  4 waves, all X           (one wave per SIMD doing both)
//   6  8 waves, all X
//   7  as 2, the V waves at s_setprio 3      8  as 2, the M waves at s_setprio 3
//   9  as 6, waves 4-7 at s_setprio 3
// Prints the kernel time and, per SIMD, cycles per round (shader clock from s_memtime around the loop of wave 0).
//   hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue <mode> <shape 0|1> <vpm>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int VPM, bool DO_M, bool DO_V>
__device__ __forceinline__ void body(int iters, float* sink, int lane) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(0.001f * (lane + e));
    b[e] = (__bf16)(0.002f * (lane - e));
  }
  f32x16 acc32[SHAPE == 0 ? 8 : 1];
  f32x4 acc16[SHAPE == 1 ? 8 : 1];
  for (auto& x : acc32)
    for (int e = 0; e < 16; ++e) x[e] = 0.0f;
  for (auto& x : acc16)
    for (int e = 0; e < 4; ++e) x[e] = 0.0f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + 0.001f * (lane + j);
  const float c0 = 0.999f, c1 = 0.0005f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (DO_M) {
        if (SHAPE == 0) acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[j], 0, 0, 0);
        else acc16[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[j], 0, 0, 0);
      }
      if (DO_V) {
#pragma unroll
        for (int k = 0; k < VPM / 8; ++k)  // (inline asm: as C the chains are packed into v_pk_fma_f32)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j + k) & 7]) : "v"(c0), "v"(c1));
      }
    }
  }
  float s = 0.0f;
  for (auto& x : acc32) s += x[0] + x[15];
  for (auto& x : acc16) s += x[0] + x[3];
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 123456.789f) sink[lane] = s;
}

template <int SHAPE, int VPM>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* sink, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const bool is_v = mode == 1 || mode == 4 || ((mode == 2 || mode == 7 || mode == 8) && wave >= 4);
  const bool is_x = mode == 5 || mode == 6 || mode == 9;
  if ((mode == 7 && wave >= 4) || (mode == 8 && wave < 4) || (mode == 9 && wave >= 4)) __builtin_amdgcn_s_setprio(3);
  if (is_x) body<SHAPE, VPM, true, true>(iters, sink, lane);
  else if (is_v) body<SHAPE, VPM, false, true>(iters, sink, lane);
  else body<SHAPE, VPM, true, false>(iters, sink, lane);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// transcendental rate: `iters` rounds of 32 independent v_exp_f32 (8 chains), 4 or 8 waves per workgroup
__global__ __launch_bounds__(512) void kexp(int iters, float* sink, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = -0.001f * (lane + j);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 123456.789f) sink[lane] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE, int VPM>
int run(int mode, int iters) {
  float* sink;
  unsigned long long* cyc;
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&cyc, 64));
  CK(hipMemset(cyc, 0, 64));
  const int waves = (mode == 0 || mode == 1 || mode == 5) ? 4 : 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SHAPE, VPM>), dim3(256), dim3(waves * 64), 0, 0, mode, iters, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8];
  CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
  printf("mode %d shape %s vpm %3d waves %d: %.3f ms; s_memtime ticks per round (8 MFMA + %d fma):", mode, SHAPE ? "16x16x32" : "32x32x16", VPM,
         waves, ms, VPM);
  for (int w = 0; w < waves; ++w) printf(" %.1f", (double)h[w] / iters);
  printf("\n");
  return 0;
}

int main(int argc, char** argv) {
  const int iters = 20000;
  for (int waves = 4; waves <= 8; waves += 4) {
    float* sink;
    unsigned long long* cyc;
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&cyc, 64));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kexp, dim3(256), dim3(waves * 64), 0, 0, iters, sink, cyc);
      CK(hipDeviceSynchronize());
    }
    unsigned long long h[8];
    CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    printf("v_exp_f32 x32 per round, %d waves: ticks per round", waves);
    for (int w = 0; w < waves; ++w) printf(" %.1f", (double)h[w] / iters);
    printf("\n");
  }
  for (int shape = 0; shape < 2; ++shape)
    for (int mode = 0; mode <= 9; ++mode) {
      int rc = 0;
      if (shape == 0) {
        rc |= run<0, 32>(mode, iters);
        if (mode == 2 || mode >= 5) rc |= run<0, 64>(mode, iters);
      } else {
        rc |= run<1, 16>(mode, iters);
        if (mode == 2 || mode >= 5) rc |= run<1, 32>(mode, iters);
      }
      if (rc) return rc;
    }
  return 0;
}
