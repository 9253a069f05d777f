// Phase timeline of the ring4 GEMM (128 x 256 tile, 8 waves, two workgroups per CU) at a headline shape: every wave stamps
// s_memtime at entry, after its ring fill is issued, when the first units have landed, at the end of the main loop, after
// the barrier that frees the ring, and when its epilogue instructions are issued (-DVDR_GEMM_STAMPS hooks in
// csrc/gemm_kernels.h; the shipped library has none).  Prints the shader clock, when workgroups start and end in real
// time (s_memrealtime), how many run at once, and the mean duration of each phase.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVDR_GEMM_STAMPS -I vit-deep-radiomics_amd/csrc tools/micro/gemm_stamps.hip -o tools/micro/gemm_stamps
//   tools/micro/gemm_stamps [shape = fc1 | fc2 | qkv | proj]
#include "gemm.hip"
#include "gemm_ring4.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const char* shape = argc > 1 ? argv[1] : "fc1";
  const int64_t M = 50432;
  int N = 3072, K = 768, epi = vdr::EPI_BIAS_GELU;
  if (!strcmp(shape, "fc2")) N = 768, K = 3072, epi = vdr::EPI_BIAS_RESID;
  if (!strcmp(shape, "qkv")) N = 2304, K = 768, epi = vdr::EPI_BIAS;
  if (!strcmp(shape, "proj")) N = 768, K = 768, epi = vdr::EPI_BIAS_RESID;
  auto fill = [](std::vector<uint16_t>& h, float scale) {
    uint32_t x = 777;
    for (auto& v : h) {
      x = x * 1664525u + 1013904223u;
      const float f = ((int)(x >> 8) % 4096 - 2048) / 2048.0f * scale;
      v = (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
    }
  };
  std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K), hR((size_t)M * N);
  fill(hA, 1.0f);
  fill(hW, 0.05f);
  fill(hR, 1.0f);
  void *A, *W, *Wp, *C, *R;
  float* bias;
  CK(hipMalloc(&A, hA.size() * 2));
  CK(hipMalloc(&W, hW.size() * 2));
  CK(hipMalloc(&Wp, hW.size() * 2));
  CK(hipMalloc(&C, (size_t)M * N * 2));
  CK(hipMalloc(&R, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, hR.data(), hR.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, N * 4));
  CK(vdr::launch_w_interleave(W, Wp, N, K, K, 0));
  const int tiles = (int)((M + 127) / 128) * ((N + 255) / 256), NW = 8;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)tiles * NW * 8 * 8));
  CK(hipMemset(stamps, 0, (size_t)tiles * NW * 8 * 8));
  vdr::g_gemm_stamps = stamps;
  vdr::GemmArgs a{};
  a.A = A;
  a.W = Wp;
  a.w_interleaved = 1;
  a.bias = bias;
  a.resid = epi == vdr::EPI_BIAS_RESID ? R : nullptr;
  a.C = C;
  a.M = M;
  a.N = N;
  a.K = K;
  a.lda = K;
  a.ldw = K;
  a.ldc = N;
  a.ldr = N;
  a.omap = vdr::RowMap{1, 1, 0};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    CK(vdr::launch_gemm(a, epi, 26, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s launch %d: %.1f us  (%.0f TFLOP/s)\n", shape, rep, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  }
  std::vector<unsigned long long> st((size_t)tiles * NW * 8);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  auto at = [&](int wg, int w) { return &st[((size_t)wg * NW + w) * 8]; };
  unsigned long long r0 = ~0ull, r1 = 0;
  double clk = 0;
  for (int wg = 0; wg < tiles; ++wg) {
    const unsigned long long* d = at(wg, 0);
    r0 = std::min(r0, d[6]);
    r1 = std::max(r1, d[7]);
    clk += (double)(d[5] - d[0]) / (double)(d[7] - d[6]);
  }
  printf("%d workgroups (tiles); shader clock %.0f MHz; first entry .. last exit %.1f us (launch %.1f us by events)\n", tiles,
         clk / tiles * 100.0, (double)(r1 - r0) / 100.0, ms * 1e3);
  {  // concurrency: workgroups alive per microsecond
    const int T = (int)((r1 - r0) / 100) + 1;
    std::vector<int> alive(T + 1, 0);
    for (int wg = 0; wg < tiles; ++wg) {
      const unsigned long long* d = at(wg, 0);
      const int a0 = (int)((d[6] - r0) / 100), a1 = (int)((d[7] - r0) / 100);
      for (int t = a0; t <= a1 && t <= T; ++t) ++alive[t];
    }
    printf("workgroups alive at t us:");
    for (int t = 0; t <= T; t += std::max(1, T / 24)) printf(" %d:%d", t, alive[t]);
    printf("\n");
    double life = 0;
    for (int wg = 0; wg < tiles; ++wg) life += (double)(at(wg, 0)[7] - at(wg, 0)[6]) / 100.0;
    printf("mean workgroup lifetime %.1f us; sum of lifetimes / (512 slots x span) = %.2f\n", life / tiles,
           life / (512.0 * (double)(r1 - r0) / 100.0));
  }
  const char* names[5] = {"addresses + ring fill issue", "wait for the first units + barrier", "main loop", "barrier after the loop",
                          "epilogue (issue)"};
  double sum[5] = {0}, tot = 0;
  size_t n = 0;
  for (int wg = 0; wg < tiles; ++wg)
    for (int w = 0; w < NW; ++w) {
      const unsigned long long* d = at(wg, w);
      for (int i = 0; i < 5; ++i) sum[i] += (double)(d[i + 1] - d[i]);
      tot += (double)(d[5] - d[0]);
      ++n;
    }
  printf("mean per wave: %.0f ticks per tile (%d K-steps of 32)\n", tot / n, K / 32);
  for (int i = 0; i < 5; ++i) printf("  %-36s %8.0f ticks  %5.1f %%\n", names[i], sum[i] / n, 100.0 * sum[i] / tot);
  return 0;
}
