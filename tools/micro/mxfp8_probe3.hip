// which of a lane's 8 operand registers belong to scale block 0 (k < 32) and which to block 1?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void probe(const int* a, const int* sa, float* c) {
  const int lane = threadIdx.x;
  v8i av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a[lane * 8 + i]; bv[i] = 0x38383838; }
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa[lane], 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) c[lane * 16 + i] = acc[i];
}
int main() {
  int *da, *dsa; float* dc;
  hipMalloc(&da, 2048); hipMalloc(&dsa, 256); hipMalloc(&dc, 4096);
  std::vector<float> c(1024);
  for (int sl : {0, 32})
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < 8; ++i) {
        std::vector<int> a(512, 0x38383838), sa(64, 0x7f7f7f7f);
        sa[sl] = 0x7f7f7f80;   // the scale held by lane sl (row 0) doubled
        a[(32 * h) * 8 + i] = 0;  // row 0, lane half h, register i zeroed
        hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, 1, 64, 0, 0, da, dsa, dc);
        hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost);
        printf("scale lane %2d doubled; zero lane-half %d reg %d: C[0][0] = %g  -> register in the %s block\n", sl, h, i, c[0],
               c[0] == 88.f ? "DOUBLED" : "other");
      }
  return 0;
}
