// Phase timeline of the persistent attention kernel (attn_persist_kernel<7, LOADER>) at the headline shape: every wave
// stamps s_memtime at its phase boundaries (-DVDR_ATTN_STAMPS hooks in csrc/attention.hip; the shipped library has none),
// the host prints the shader clock during the launch (s_memtime per 100 MHz s_memrealtime), when the workgroups start and
// end, and the mean duration of each phase for the computing waves and for the loader wave.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVDR_ATTN_STAMPS -I vit-deep-radiomics_amd/csrc tools/micro/attn_stamps.hip -o tools/micro/attn_stamps
//   tools/micro/attn_stamps [batch = 256] [loader = 1]
#include "attention.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, seq = 197, H = 12;
  const bool loader = argc > 2 ? atoi(argv[2]) != 0 : true;
  const size_t tokens = (size_t)B * seq;
  std::vector<uint16_t> h(tokens * 3 * H * 64);
  uint32_t x = 12345;
  for (auto& v : h) {  // bf16 values in about [-2, 2)
    x = x * 1664525u + 1013904223u;
    const float f = ((int)(x >> 8) % 4096 - 2048) / 1024.0f;
    v = (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
  }
  void *qkv, *out;
  unsigned long long* stamps;
  CK(hipMalloc(&qkv, h.size() * 2));
  CK(hipMalloc(&out, tokens * H * 64 * 2));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int n_items = B * H, grid = std::min(n_items, 256), NWV = 8, NI = 16;
  const size_t nst = (size_t)grid * NWV * NI * 16;
  CK(hipMalloc(&stamps, nst * 8));
  CK(hipMemset(stamps, 0, nst * 8));
  vdr::AttnK k{};
  k.qkv = (const bf16_t*)qkv;
  k.out = (bf16_t*)out;
  k.seq = seq;
  k.heads = H;
  k.ld_qkv = 3 * H * 64;
  k.ld_out = H * 64;
  k.qt_per_block = 7;
  k.n_chunks = 1;
  k.stamps = stamps;
  const size_t lds = 2 * (size_t)(2 * 7 * 32 * 128);
  const void* fn = loader ? (const void*)vdr::attn_persist_kernel<7, true> : (const void*)vdr::attn_persist_kernel<7, false>;
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    if (loader) hipLaunchKernelGGL((vdr::attn_persist_kernel<7, true>), dim3(grid), dim3(512), lds, 0, k, n_items);
    else hipLaunchKernelGGL((vdr::attn_persist_kernel<7, false>), dim3(grid), dim3(448), lds, 0, k, n_items);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch %d: %.1f us\n", rep, ms * 1e3);
  }
  std::vector<unsigned long long> st(nst);
  CK(hipMemcpy(st.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
  const int items = std::min(NI, (n_items + grid - 1) / grid);
  auto at = [&](int wg, int w, int it) { return &st[(((size_t)wg * NWV + w) * NI + it) * 16]; };
  {
    double r = 0;
    unsigned long long r0 = ~0ull;
    std::vector<double> b0(grid), b1(grid);
    for (int wg = 0; wg < grid; ++wg) r0 = std::min(r0, at(wg, 0, 0)[9]);
    for (int wg = 0; wg < grid; ++wg) {
      const unsigned long long *a = at(wg, 0, 0), *z = at(wg, 0, items - 1);
      r += (double)(z[7] - a[0]) / (double)(z[10] - a[9]);
      b0[wg] = (double)(a[9] - r0) / 100.0;
      b1[wg] = (double)(z[10] - r0) / 100.0;
    }
    std::sort(b0.begin(), b0.end());
    std::sort(b1.begin(), b1.end());
    printf("shader clock during the launch: %.0f MHz (s_memtime per s_memrealtime, mean over workgroups)\n", r / grid * 100.0);
    printf("workgroup loop start: min %.1f median %.1f max %.1f us;  loop end: min %.1f median %.1f max %.1f us  (launch %.1f us by events)\n",
           b0[0], b0[grid / 2], b0[grid - 1], b1[0], b1[grid / 2], b1[grid - 1], ms * 1e3);
  }
  const char* cn[7] = {"barrier (top of item)", "QK^T (28 MFMA + K reads)", "mask + row max", "exp / P.V (28 MFMA + V reads)",
                       "vmcnt(0): next Q / pieces", "scale + stores", "lgkmcnt / end of item"};
  double sum[7] = {0}, tot = 0;
  size_t n = 0;
  for (int wg = 0; wg < grid; ++wg)
    for (int w = 0; w < 7; ++w)
      for (int it = 0; it < items; ++it) {
        const unsigned long long* d = at(wg, w, it);
        for (int i = 0; i < 7; ++i) sum[i] += (double)(d[i + 1] - d[i]);
        tot += (double)(d[7] - d[0]);
        ++n;
      }
  printf("computing waves, mean per (wave, item): %.0f ticks (%d items per workgroup)\n", tot / n, items);
  for (int i = 0; i < 7; ++i) printf("  %-32s %8.0f ticks  %5.1f %%\n", cn[i], sum[i] / n, 100.0 * sum[i] / tot);
  if (loader) {
    double s0 = 0, s1 = 0, s2 = 0;
    size_t m = 0;
    for (int wg = 0; wg < grid; ++wg)
      for (int it = 0; it + 1 < items; ++it) {
        const unsigned long long* d = at(wg, 7, it);
        s0 += (double)(d[1] - d[0]);
        s1 += (double)(d[2] - d[1]);
        s2 += (double)(d[7] - d[2]);
        ++m;
      }
    printf("loader wave, mean per item: barrier %.0f, issue of 56 LDS-DMA pieces %.0f, vmcnt(0) %.0f ticks\n", s0 / m, s1 / m, s2 / m);
  }
  printf("workgroup 0, ticks since its first stamp: item x wave -> [top, barrier, QK^T, max, P.V, landed, stored, end]\n");
  const unsigned long long t0 = at(0, 0, 0)[0];
  for (int it = 0; it < std::min(items, 3); ++it)
    for (int w : {0, 4, 6, 7}) {
      if (w == 7 && !loader) continue;
      const unsigned long long* d = at(0, w, it);
      printf("  item %d wave %d:", it, w);
      for (int i = 0; i < 8; ++i) printf(" %7lld", (long long)(d[i] - t0));
      printf("\n");
    }
  return 0;
}
