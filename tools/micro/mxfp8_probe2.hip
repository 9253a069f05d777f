// which C entries change when ONE lane's scale byte changes?  (all data = 1.0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void probe(const int* sa, const int* sb, float* c) {
  const int lane = threadIdx.x;
  v8i av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = 0x38383838; bv[i] = 0x38383838; }
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa[lane], 0, sb[lane]);
  for (int i = 0; i < 16; ++i) c[lane * 16 + i] = acc[i];
}
int main() {
  int *dsa, *dsb; float* dc;
  hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 4096);
  std::vector<float> c(1024);
  for (int which = 0; which < 2; ++which)
    for (int L : {0, 3, 35, 40}) {
      std::vector<int> sa(64, 0x7f7f7f7f), sb(64, 0x7f7f7f7f);
      (which ? sb : sa)[L] = 0x7f7f7f80;  // byte 0 -> 2x
      hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
      hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, 1, 64, 0, 0, dsa, dsb, dc);
      hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost);
      printf("%s lane %d x2: ", which ? "scale_b" : "scale_a", L);
      int n = 0;
      for (int lane = 0; lane < 64; ++lane)
        for (int reg = 0; reg < 16; ++reg) {
          const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          const float v = c[lane * 16 + reg];
          if (v != 64.f && n < 6) { printf("C[%d][%d]=%g ", row, col, v); ++n; }
          else if (v != 64.f) ++n;
        }
      printf(" (%d entries differ from 64)\n", n);
    }
  return 0;
}
