// Where a step of the stream GEMM (csrc/gemm_stream.hip, variant 30) spends its time: every wave sums, over all its
// 64-deep steps, the cycles from one mid-step barrier to the next stamp (the two MFMA halves), the wait for its last
// fragment reads, the counted vmcnt wait for the stage two steps ahead, and the barrier (-DVDR_STREAM_STAMPS hooks; the
// shipped library has none).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DVDR_STREAM_STAMPS -I vit-deep-radiomics_amd/csrc tools/micro/stream_stamps.hip -o tools/micro/stream_stamps
//   tools/micro/stream_stamps [shape = fc1 | qkv | widek]
#ifdef ST_MFMA32
#include "gemm_stream_mfma32.hip"  // the 32x32x16 experiment (this directory)
#else
#include "gemm_stream.hip"
#endif
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
  if (!strcmp(shape, "qkv")) N = 2304, K = 768, epi = vdr::EPI_BIAS;
  if (!strcmp(shape, "widek")) N = 768, K = 3072, epi = vdr::EPI_BIAS;
  auto fill = [](std::vector<uint16_t>& h, float scale) {
    uint32_t x = 777;
    for (auto& v : h) {
      x = x * 1664525u + 1013904223u;
      const float f = ((int)(x >> 8) % 4096 - 2048) / 2048.0f * scale;
      v = (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
    }
  };
  std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
  fill(hA, 1.0f);
  fill(hW, 0.05f);
  void *A, *W, *C;
  float* bias;
  CK(hipMalloc(&A, hA.size() * 2));
  CK(hipMalloc(&W, hW.size() * 2));
  CK(hipMalloc(&C, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, N * 4));
  const int NWG = 256, NW = 8;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)NWG * NW * 8 * 8));
  CK(hipMemset(stamps, 0, (size_t)NWG * NW * 8 * 8));
  vdr::g_stream_stamps = stamps;
  vdr::GemmArgs a{};
  a.A = A;
  a.W = W;
  a.bias = bias;
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
    CK(vdr::launch_gemm_stream(a, epi, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s launch %d: %.1f us  (%.0f TFLOP/s)\n", shape, rep, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  }
  std::vector<unsigned long long> st((size_t)NWG * NW * 8);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double sum[4] = {0}, steps = 0, flush = 0, life = 0;
  unsigned long long r0 = ~0ull, r1 = 0;
  double mn_steps = 1e9, mx_steps = 0;
  for (int i = 0; i < NWG * NW; ++i) {
    const unsigned long long* d = &st[(size_t)i * 8];
    for (int k = 0; k < 4; ++k) sum[k] += (double)d[k];
    steps += (double)d[4];
    mn_steps = std::min(mn_steps, (double)d[4]);
    mx_steps = std::max(mx_steps, (double)d[4]);
    flush += (double)d[5];
    r0 = std::min(r0, d[6]);
    r1 = std::max(r1, d[7]);
    life += (double)(d[7] - d[6]) / 100.0;
  }
  const double tot = sum[0] + sum[1] + sum[2] + sum[3];
  printf("steps per wave: %.0f .. %.0f; first entry .. last exit %.1f us; mean wave lifetime %.1f us\n", mn_steps, mx_steps,
         (double)(r1 - r0) / 100.0, life / (NWG * NW));
  printf("shader clock ~ %.0f MHz (cycles summed over steps / wave lifetime)\n", tot / (NWG * NW) / (life / (NWG * NW)));
  const char* names[4] = {"two MFMA halves (barrier .. last slot issued)", "wait for the last fragment reads (lgkmcnt 0)",
                          "counted vmcnt wait (stage + 1 landed)", "s_barrier"};
  printf("mean per step: %.0f cycles\n", tot / steps);
  for (int k = 0; k < 4; ++k) printf("  %-48s %7.0f cycles  %5.1f %%\n", names[k], sum[k] / steps, 100.0 * sum[k] / tot);
  printf("flush of the last tile's epilogue: %.0f cycles per wave\n", flush / (NWG * NW));
  printf("per wave index (mean cycles per step: halves / lgkm / vmcnt / barrier):\n");
  for (int w = 0; w < NW; ++w) {
    double s4[4] = {0}, n = 0;
    for (int g = 0; g < NWG; ++g) {
      const unsigned long long* d = &st[((size_t)g * NW + w) * 8];
      for (int k = 0; k < 4; ++k) s4[k] += (double)d[k];
      n += (double)d[4];
    }
    printf("  wave %d: %6.0f %5.0f %5.0f %6.0f\n", w, s4[0] / n, s4[1] / n, s4[2] / n, s4[3] / n);
  }
  return 0;
}
