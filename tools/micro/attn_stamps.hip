// Phase timeline of the two-pass persistent attention kernel (attn_persist2_kernel<7>) at the headline shape:
// every wave stamps s_memtime at its phase boundaries (-DVDR_ATTN_STAMPS hooks in csrc/attention.hip), the host prints
// the mean duration of each phase and the timeline of one workgroup.
//   hipcc -O3 --offload-arch=gfx950 -DVDR_ATTN_STAMPS -I vit-deep-radiomics_amd/csrc tools/micro/attn_stamps.hip -o tools/micro/attn_stamps
#include "attention.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, seq = 197, H = 12;
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
  const int grid = 512, NT = 7, NI = 8;
  const size_t nst = (size_t)grid * NT * NI * 16;
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
  const size_t lds = 3 * 208 * 128 + 2048;
  auto fn = vdr::attn_persist2_kernel<7>;
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(448), lds, 0, k, B * H);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch %d: %.1f us\n", rep, ms * 1e3);
  }
  std::vector<unsigned long long> st(nst);
  CK(hipMemcpy(st.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
  const char* names[8] = {"barrier A (top)", "stage issue", "pass 1 (QK^T, max)", "vmcnt(0): V / next K landed", "barrier B",
                          "pass 2 (QK^T, exp, PV)", "vmcnt(0): next Q", "stores + lgkmcnt"};
  const int items = B * H / grid;
  double sum[8] = {0}, tot = 0;
  unsigned long long t0 = ~0ull, t1 = 0;
  size_t n = 0;
  for (int wg = 0; wg < grid; ++wg)
    for (int w = 0; w < NT; ++w)
      for (int it = 0; it < items && it < NI; ++it) {
        const unsigned long long* d = &st[(((size_t)wg * NT + w) * NI + it) * 16];
        for (int i = 0; i < 8; ++i) sum[i] += (double)(d[i + 1] - d[i]);
        tot += (double)(d[8] - d[0]);
        if (d[0] < t0) t0 = d[0];
        if (d[8] > t1) t1 = d[8];
        ++n;
      }
  {  // shader clock: s_memtime ticks per 100 MHz tick over each workgroup's whole life (wave 0)
    double r = 0;
    for (int wg = 0; wg < grid; ++wg) {
      const unsigned long long* a = &st[(((size_t)wg * NT + 0) * NI + 0) * 16];
      const unsigned long long* z = &st[(((size_t)wg * NT + 0) * NI + (items < NI ? items : NI) - 1) * 16];
      r += (double)(z[8] - a[0]) / (double)(z[10] - a[9]);
    }
    const unsigned long long* a = &st[0];
    const unsigned long long* z = &st[(((size_t)0 * NT + 0) * NI + (items < NI ? items : NI) - 1) * 16];
    printf("shader clock during the launch: %.0f MHz (s_memtime per s_memrealtime, mean over workgroups); workgroup 0 lived %.1f us\n",
           r / grid * 100.0, (double)(z[10] - a[9]) / 100.0);
  }
  {  // when did each workgroup start / end, in real time (s_memrealtime is one clock for the whole device)
    unsigned long long r0 = ~0ull;
    std::vector<double> b0(grid), b1(grid);
    for (int wg = 0; wg < grid; ++wg) r0 = std::min(r0, st[(((size_t)wg * NT + 0) * NI + 0) * 16 + 9]);
    for (int wg = 0; wg < grid; ++wg) {
      b0[wg] = (double)(st[(((size_t)wg * NT + 0) * NI + 0) * 16 + 9] - r0) / 100.0;
      b1[wg] = (double)(st[(((size_t)wg * NT + 0) * NI + (items < NI ? items : NI) - 1) * 16 + 10] - r0) / 100.0;
    }
    printf("workgroup: loop start us -> loop end us (relative to the first workgroup's start)\n");
    for (int wg = 0; wg < grid; wg += 1) if (wg < 24 || wg % 32 == 0 || wg >= grid - 8) printf("  wg %3d: %6.1f -> %6.1f\n", wg, b0[wg], b1[wg]);
    std::sort(b0.begin(), b0.end());
    std::sort(b1.begin(), b1.end());
    printf("start: min %.1f median %.1f p90 %.1f max %.1f us;  end: min %.1f median %.1f max %.1f us\n", b0[0], b0[grid / 2], b0[grid * 9 / 10],
           b0[grid - 1], b1[0], b1[grid / 2], b1[grid - 1]);
  }
  printf("items per workgroup %d; first stamp .. last stamp: %llu ticks (s_memtime) for a %.1f us launch -> %.1f ticks/us\n", items,
         t1 - t0, ms * 1e3, (double)(t1 - t0) / (ms * 1e3));
  printf("mean per (wave, item): %.0f ticks\n", tot / n);
  for (int i = 0; i < 8; ++i) printf("  %-30s %8.0f ticks  %5.1f %%\n", names[i], sum[i] / n, 100.0 * sum[i] / tot);
  for (int wg : {0}) {
    printf("workgroup %d, stamps relative to the launch's first (ticks): item x wave -> [topA, A, issued, p1, V landed, B, p2, Q landed, end]\n", wg);
    for (int it = 0; it < items && it < NI; ++it)
      for (int w : {0, 3, 6}) {
        const unsigned long long* d = &st[(((size_t)wg * NT + w) * NI + it) * 16];
        printf("  item %d wave %d:", it, w);
        for (int i = 0; i < 9; ++i) printf(" %7llu", d[i] - t0);
        printf("\n");
      }
  }
  return 0;
}
