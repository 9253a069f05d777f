// Do a CU's vector stores block its vector loads?  Each workgroup (one per CU, 8 waves):
//   mode bit0: waves 0-3 stream LDS-DMA loads (global_load_lds, L2-resident 32 MB source)
//   mode bit1: waves 4-7 stream 16-byte global stores to a private 1.2 GB region
// We report GB/s of each side alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void pipe_kernel(const char* src, char* dst, int mode, int iters, size_t src_bytes_per_wg, size_t dst_bytes_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (mode == 128) {
    // contiguous 16-byte non-temporal stores by all 8 waves
    char* d = dst + (size_t)blockIdx.x * dst_bytes_per_wg;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4 v = {(unsigned)tid, (unsigned)tid, (unsigned)tid, (unsigned)tid};
    for (int it = 0; it < iters; ++it) {
      const size_t base = ((size_t)it * 8 + wave) * 49152 % (dst_bytes_per_wg - 65536);
#pragma unroll
      for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(d + base + q * 1024 + lane * 16));
    }
    return;
  }
  if (mode == 32 || mode == 64) {
    // store patterns: 32 = contiguous 1 KB per wave-instruction; 64 = "row per lane": lane l writes 16 B
    // at row (l & 31) * 1536 B + (l >> 5) * 16 (32 rows x 32 B per instruction, as a T21 GEMM epilogue)
    char* d = dst + (size_t)blockIdx.x * dst_bytes_per_wg;
    const uint4 v = make_uint4(tid, tid, tid, tid);
    for (int it = 0; it < iters; ++it) {
      const size_t base = ((size_t)it * 8 + wave) * 49152 % (dst_bytes_per_wg - 65536);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = mode == 32 ? base + q * 1024 + lane * 16 : base + (size_t)(lane & 31) * 1536 + q * 32 + (lane >> 5) * 16;
        *reinterpret_cast<uint4*>(d + off) = v;
      }
    }
    return;
  }
  if (mode == 8) {
    const char* s = src + (size_t)blockIdx.x * src_bytes_per_wg;
    for (int it = 0; it < iters; ++it) {
      const size_t base = ((size_t)it * 8 + wave) * 8192 % src_bytes_per_wg;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (base + q * 1024 + lane * 16) % src_bytes_per_wg),
                                         (__attribute__((address_space(3))) void*)(smem + wave * 8192 + q * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (mode == 16 && wave >= 4) return;
  if (mode & (4 | 16)) {
    // register loads: every wave streams 16-byte global loads (16 in flight), result kept alive by xor
    const char* s = src + (size_t)blockIdx.x * src_bytes_per_wg;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
      const size_t base = ((size_t)it * 8 + wave) * 16384 % src_bytes_per_wg;
      uint4 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const uint4*>(s + (base + q * 1024 + lane * 16) % src_bytes_per_wg);
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc.x ^= v[q].x; acc.y ^= v[q].y; acc.z ^= v[q].z; acc.w ^= v[q].w; }
    }
    if (acc.x == 0x12345678u) *reinterpret_cast<uint4*>(dst + (size_t)blockIdx.x * dst_bytes_per_wg + tid * 16) = acc;
    return;
  }
  if (wave < 4) {
    if (!(mode & 1)) return;
    const char* s = src + (size_t)blockIdx.x * src_bytes_per_wg;
    for (int it = 0; it < iters; ++it) {
      // 16 pieces of 1 KB per wave per iteration, wrapping inside the per-WG source window
      const size_t base = ((size_t)it * 4 + wave) * 16384 % src_bytes_per_wg;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (base + q * 1024 + lane * 16) % src_bytes_per_wg),
                                         (__attribute__((address_space(3))) void*)(smem + wave * 16384 + q * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (!(mode & 2)) return;
    char* d = dst + (size_t)blockIdx.x * dst_bytes_per_wg;
    const uint4 v = make_uint4(tid, tid, tid, tid);
    for (int it = 0; it < iters; ++it) {
      const size_t base = ((size_t)it * 4 + (wave - 4)) * 4096 % dst_bytes_per_wg;  // 4 stores of 1 KB per wave per iteration
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(d + (base + q * 1024 + lane * 16) % dst_bytes_per_wg) = v;
    }
  }
}

int main(int argc, char** argv) {
  const int n_wg = 256, iters = 2000;
  const size_t src_per = (argc > 1 ? atoi(argv[1]) : 128) * 1024, dst_per = 4 * 1024 * 1024;
  printf("source window per workgroup: %zu KB (total %.1f MB)\n", src_per / 1024, src_per * n_wg / 1e6);
  char *src, *dst;
  CK(hipMalloc(&src, src_per * n_wg));
  CK(hipMalloc(&dst, dst_per * n_wg));
  CK(hipMemset(src, 1, src_per * n_wg));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipFuncSetAttribute((const void*)pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int modes[] = {1, 8, 4, 16, 32, 64, 128};
  for (int mi = 0; mi < 7; ++mi) {
    const int mode = modes[mi];
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(pipe_kernel, dim3(n_wg), dim3(512), 65536, 0, src, dst, mode, iters, src_per, dst_per);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      const double st2 = (mode == 32 || mode == 64 || mode == 128) ? (double)n_wg * iters * 8 * 4096 : 0;
      const double ld = mode == 8 ? (double)n_wg * iters * 8 * 8192 : mode == 16 ? (double)n_wg * iters * 4 * 16384 : (mode & 4) ? (double)n_wg * iters * 8 * 16384 : (mode & 1) ? (double)n_wg * iters * 4 * 16384 : 0, st = (mode == 32 || mode == 64 || mode == 128) ? st2 : (mode & 2) ? (double)n_wg * iters * 4 * 4096 : 0;
      if (rep) printf("mode %d (%s%s): %.3f ms  loads %.2f TB/s (%.1f GB/s/CU)  stores %.2f TB/s (%.1f GB/s/CU)\n", mode, mode == 128 ? "S8-nt" : mode == 32 ? "S8-contig" : mode == 64 ? "S8-rowperlane" : mode == 8 ? "L8" : mode == 16 ? "R4" : (mode & 4) ? "R8" : (mode & 1) ? "L4" : "-", (mode & 2) ? "S" : "-", ms,
                      ld / ms / 1e9, ld / ms / 1e6 / n_wg, st / ms / 1e9, st / ms / 1e6 / n_wg);
    }
  }
  return 0;
}
