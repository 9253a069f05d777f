// Issue cost of the vector instructions the epilogues and the softmax are made of, on one gfx950 SIMD:
// cycles per wave-instruction for v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32, v_exp_f32, v_max3_f32,
// v_cvt_pk_bf16_f32, with one and with two waves per SIMD (8 independent chains each, no memory).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int OP>
__global__ __launch_bounds__(512) void k(int iters, float* sink, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x2 v[8];
  for (int j = 0; j < 8; ++j) v[j] = f32x2{1.0f + 0.001f * (lane + j), 0.5f + 0.002f * (lane - j)};
  const f32x2 c0 = {0.999f, 0.998f}, c1 = {0.0005f, 0.0004f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j][0]) : "v"(c0[0]), "v"(c1[0]));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c0), "v"(c1));
        if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1));
        if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c0));
        if (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j][0]));
        if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[j][0]) : "v"(c0[0]), "v"(c1[0]));
        if (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[j][0]) : "v"(v[j][1]), "v"(c0[0]));
        if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j][0]) : "v"(c1[0]));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
  if (s == 123456.789f) sink[lane] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
  float* sink;
  unsigned long long* cyc;
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&cyc, 64));
  const int iters = 4000;
  const char* names[8] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_exp_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_add_f32"};
  void (*ks[8])(int, float*, unsigned long long*) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>};
  for (int op = 0; op < 8; ++op)
    for (int waves : {4, 8}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(ks[op], dim3(256), dim3(waves * 64), 0, 0, iters, sink, cyc);
        CK(hipDeviceSynchronize());
      }
      unsigned long long h[8];
      CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
      printf("%-18s %d wave(s) per SIMD: %.2f cycles per instruction per wave, %.2f per SIMD\n", names[op], waves / 4,
             (double)h[0] / (iters * 32.0), (double)h[0] / (iters * 32.0) / (waves / 4));
    }
  return 0;
}
