// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950: checks the hypothesis
//   C[r][c] = sum_{blk=0,1} 2^(sa[lane r + 32 blk] - 127) * 2^(sb[lane c + 32 blk] - 127)
//                           * sum_{h=0,1} sum_{j<16} A_lane(r,h)[16 blk + j] * B_lane(c,h)[16 blk + j]
// i.e. scale block `blk` = bytes [16 blk, 16 blk + 16) of BOTH lane halves of a row (found with
// mxfp8_probe2/3), its scale byte is supplied by lane r + 32 blk, and opsel picks the byte of that VGPR.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int OPA, int OPB>
__global__ void probe(const int* a, const int* b, const int* sa, const int* sb, float* c, int use_scale) {
  const int lane = threadIdx.x;
  v8i av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = a[lane * 8 + i];
    bv[i] = b[lane * 8 + i];
  }
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (use_scale)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, OPA, sa[lane], OPB, sb[lane]);
  else
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0, 0, 0);
  for (int i = 0; i < 16; ++i) c[lane * 16 + i] = acc[i];
}

static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

int main() {
  std::vector<int> a(512), b(512), sa(64), sb(64);
  std::vector<float> c(1024);
  srand(7);
  auto rnd8 = []() {  // finite e4m3 with small exponents (exact in fp32 sums)
    uint8_t e = 5 + rand() % 5, m = rand() % 8, s = rand() % 2;
    return (uint8_t)((s << 7) | (e << 3) | m);
  };
  for (auto* v : {&a, &b})
    for (auto& w : *v) w = rnd8() | (rnd8() << 8) | (rnd8() << 16) | ((uint32_t)rnd8() << 24);
  for (auto* v : {&sa, &sb})
    for (auto& w : *v) w = (125 + rand() % 5) | ((125 + rand() % 5) << 8) | ((125 + rand() % 5) << 16) | ((125 + rand() % 5) << 24);
  int *da, *db, *dsa, *dsb;
  float* dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 4096);
  hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  auto check = [&](int use_scale, int opa, int opb, const char* tag) {
    hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int reg = 0; reg < 16; ++reg) {
        const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        // transposed-issue convention of this repo: first operand indexes the C row?  try both
        double e0 = 0, e1 = 0;
        for (int h = 0; h < 2; ++h) {  // h = scale block here
          double d0 = 0, d1 = 0;
          for (int half = 0; half < 2; ++half)
            for (int j = 0; j < 16; ++j) {
              const uint8_t* pa_r = (const uint8_t*)&a[(row + 32 * half) * 8];
              const uint8_t* pb_c = (const uint8_t*)&b[(col + 32 * half) * 8];
              const uint8_t* pa_c = (const uint8_t*)&a[(col + 32 * half) * 8];
              const uint8_t* pb_r = (const uint8_t*)&b[(row + 32 * half) * 8];
              d0 += (double)e4m3(pa_r[16 * h + j]) * e4m3(pb_c[16 * h + j]);
              d1 += (double)e4m3(pa_c[16 * h + j]) * e4m3(pb_r[16 * h + j]);
            }
          double s0 = 1, s1 = 1;
          if (use_scale) {
            s0 = ldexp(1.0, ((sa[row + 32 * h] >> (8 * opa)) & 255) - 127) * ldexp(1.0, ((sb[col + 32 * h] >> (8 * opb)) & 255) - 127);
            s1 = ldexp(1.0, ((sa[col + 32 * h] >> (8 * opa)) & 255) - 127) * ldexp(1.0, ((sb[row + 32 * h] >> (8 * opb)) & 255) - 127);
          }
          e0 += d0 * s0;
          e1 += d1 * s1;
        }
        const double got = c[lane * 16 + reg];
        const double err = fmin(fabs(got - e0), 1e30);
        (void)e1;
        worst = fmax(worst, err / (fabs(e0) + 1e-3));
      }
    printf("%s: A-row/B-col hypothesis worst rel err %.3e\n", tag, worst);
  };
  hipLaunchKernelGGL((probe<0, 0>), 1, 64, 0, 0, da, db, dsa, dsb, dc, 0);
  hipDeviceSynchronize();
  check(0, 0, 0, "unscaled      ");
  hipLaunchKernelGGL((probe<0, 0>), 1, 64, 0, 0, da, db, dsa, dsb, dc, 1);
  hipDeviceSynchronize();
  check(1, 0, 0, "scaled op 0,0 ");
  hipLaunchKernelGGL((probe<1, 2>), 1, 64, 0, 0, da, db, dsa, dsb, dc, 1);
  hipDeviceSynchronize();
  check(1, 1, 2, "scaled op 1,2 ");
  hipLaunchKernelGGL((probe<3, 3>), 1, 64, 0, 0, da, db, dsa, dsb, dc, 1);
  hipDeviceSynchronize();
  check(1, 3, 3, "scaled op 3,3 ");
  return 0;
}
