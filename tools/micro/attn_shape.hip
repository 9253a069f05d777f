// Does the MFMA shape change what an attention item costs?  The headline attention kernel (csrc/attention.hip,
// attn_persist_kernel<7, true>) is `sum(matrix pipe + vector issue)` on its fullest SIMD; tools/micro/coissue.hip
// says the 8-pass 32x32x16 MFMA starves the other wave's vector instructions while the 4-pass 16x16x32 one leaves the port free
// half of the time.  This is the per-item compute skeleton of that kernel -- K / V images resident in LDS (swizzled as the
// kernel stages them), Q in registers, S^T = K.Q^T, exact one-pass softmax, P fed back from the accumulators, O^T = V^T.P^T --
// in both shapes, 7 computing waves per workgroup, one workgroup per CU, no global traffic inside the loop:
//   shape 0: 32x32x16 -- one 32-query tile per wave, S^T as 7 tiles of 32 keys (112 registers), 28 + 26 MFMAs of 32 cycles
//   shape 1: 16x16x32 -- the same 32 queries as two 16-query column blocks, S^T as 14 x 2 tiles of 16 keys (112 registers),
//            56 + 52 MFMAs of 16 cycles; a softmax row lives in 4 lanes instead of 2; V^T fragments by the same transposed LDS read
// Same LDS bytes, same exponentials, same matrix-pipe cycles.  Prints cycles per item (shader clock) and the time per item.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o attn_shape attn_shape.hip && ./attn_shape [items] [mode]   (mode: see the kernel)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int SEQ = 197, NT = 7, KEYS = NT * 32;
constexpr int IMG = KEYS * 128;  // one [key][64 d] image

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// fill the K / V images (values in [-1, 1), K chunk ^ ((key >> 1) & 7), V chunk ^ (((key >> 1) & 1) << 2): as staged by the kernel)
__device__ void fill_images(char* smem, int tid, int nthr) {
  for (int i = tid; i < 2 * KEYS * 64; i += nthr) {
    const int img = i / (KEYS * 64), r = (i / 64) % KEYS, d = i % 64;
    const int chunk = d >> 3, e = d & 7;
    const int sw = img == 0 ? (chunk ^ ((r >> 1) & 7)) : (chunk ^ (((r >> 1) & 1) << 2));
    const float v = (float)(((r * 131 + d * 17 + img * 7) % 255) - 127) * (1.0f / 128.0f);
    reinterpret_cast<bf16_t*>(smem + img * IMG + r * 128 + sw * 16)[e] = (bf16_t)(r < SEQ ? v : 0.0f);
  }
}

// ---- shape 0: the kernel's own structure ---------------------------------------------------------------------------
// abl (compile time): 4 = no MFMAs (one vector add per fragment instead), 8 = no exponentials, 16 = no LDS fragment reads (32: K only, 64: V only)
// PF (compile time): 1 = the K fragments of tile t + 1 are requested before the 4 MFMAs of tile t, the V fragments of slice
// it + 1 before the exponentials of slice it + 1 (both one full step ahead instead of just in time)
template <int abl, int PF = 0>
__device__ __forceinline__ float item_32(const char* smem, const bf16x8 (&qf)[4], int lane) {
  const int hh = lane >> 5, l31 = lane & 31, swz = (lane >> 1) & 7;
  const char* sK = smem + l31 * 128;
  const int tq = (lane & 15) >> 2, tp = lane & 3, dg = (lane >> 4) & 1;
  const int vkey = 4 * hh + tq;
  const __attribute__((address_space(3))) char* sV = (const __attribute__((address_space(3))) char*)(smem + IMG) + vkey * 128 + 8 * (tp & 1);
  int vch[2], kch[4];
  for (int nd = 0; nd < 2; ++nd) vch[nd] = ((4 * nd + 2 * dg + (tp >> 1)) ^ (((vkey >> 1) & 1) << 2)) * 16;
  for (int ks = 0; ks < 4; ++ks) kch[ks] = ((2 * ks + hh) ^ swz) * 16;
  const float sc = 0.125f * 1.44269504088896341f;
  f32x16 s[NT];
  __builtin_amdgcn_s_setprio(0);
  bf16x8 kq[3][4];  // ring of PF + 1 tiles of K fragments
  if constexpr (PF != 0) {
#pragma unroll
    for (int t0 = 0; t0 < PF; ++t0)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kq[t0][ks] = *reinterpret_cast<const bf16x8*>(sK + t0 * 32 * 128 + kch[ks]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[t][e] = (t == NT - 1 && (e & 3) + 8 * (e >> 2) + 4 * hh >= SEQ - (NT - 1) * 32) ? -INFINITY : 0.0f;
    if constexpr (PF != 0) {
      if (t + PF < NT) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kq[(t + PF) % (PF + 1)][ks] = *reinterpret_cast<const bf16x8*>(sK + (t + PF) * 32 * 128 + kch[ks]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 kf = qf[ks];
      kf[0] = (bf16_t)(0.01f * (float)(t + 1));  // (ablations: a different operand per tile, or the 7 chains are one common subexpression)
      if constexpr (PF != 0) kf = kq[t % (PF + 1)][ks];
      else if constexpr (!(abl & 16) && !(abl & 32)) kf = *reinterpret_cast<const bf16x8*>(sK + t * 32 * 128 + kch[ks]);
      if constexpr ((abl & 4) != 0) s[t][ks] += (float)kf[0];
      else s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
    }
    if constexpr (PF != 0) __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(2);
  float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) m4[(e >> 1) & 3] = fmaxf(m4[(e >> 1) & 3], s[t][e]);
  float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float nmb = -mx * sc;
  f32x2 lsum2 = {0.0f, 0.0f};
  f32x16 o[2];
  for (int nd = 0; nd < 2; ++nd)
    for (int e = 0; e < 16; ++e) o[nd][e] = 0.0f;
  const f32x2 sc2 = {sc, sc}, nmb2 = {nmb, nmb};
  bf16x4 vq[2][4];  // [slot][nd * 2 + (lo, hi)]
  auto read_v = [&](int it, int slot) {
#pragma unroll
    for (int nd = 0; nd < 2; ++nd) {
      vq[slot][2 * nd] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + it * 16 * 128 + vch[nd]));
      vq[slot][2 * nd + 1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + it * 16 * 128 + 8 * 128 + vch[nd]));
    }
  };
  if constexpr (PF != 0) read_v(0, 0);
#pragma unroll
  for (int it = 0; it < 13; ++it) {  // 13 slices of 16 keys hold the 197 keys
    const int t = it >> 1, s2 = it & 1;
    bf16x8 pf;
    if constexpr (PF != 0) {
      if (it + 1 < 13) read_v(it + 1, (it + 1) & 1);
    }
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f32x2 x = {s[t][8 * s2 + j], s[t][8 * s2 + j + 1]};
      x = __builtin_elementwise_fma(x, sc2, nmb2);
      f32x2 pv = x;
      if constexpr (!(abl & 8)) pv = f32x2{fast_exp2(x[0]), fast_exp2(x[1])};
      lsum2 += pv;
      pf[j] = (bf16_t)pv[0];
      pf[j + 1] = (bf16_t)pv[1];
    }
#pragma unroll
    for (int nd = 0; nd < 2; ++nd) {
      bf16x8 vf = pf;
      if constexpr (PF != 0) {
        for (int j = 0; j < 4; ++j) {
          vf[j] = vq[it & 1][2 * nd][j];
          vf[4 + j] = vq[it & 1][2 * nd + 1][j];
        }
      } else if constexpr (!(abl & 16) && !(abl & 64)) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + it * 16 * 128 + vch[nd]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + it * 16 * 128 + 8 * 128 + vch[nd]));
        for (int j = 0; j < 4; ++j) {
          vf[j] = lo[j];
          vf[4 + j] = hi[j];
        }
      }
      if constexpr ((abl & 4) != 0) o[nd][it] += (float)vf[0] * (float)pf[0];
      else o[nd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[nd], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const float lsum = lsum2[0] + lsum2[1];
  const float l = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / l;
  float acc = 0.0f;
  for (int nd = 0; nd < 2; ++nd)
    for (int e = 0; e < 16; ++e) acc += o[nd][e] * inv;
  return acc;
}

// ---- shape 1: 16x16x32 --------------------------------------------------------------------------------------------
// S^T tile (kb, qb): keys 16 kb + 4 g + e (g = lane >> 4), query 16 qb + (lane & 15).
// P^T operand of key chunk j (32 keys): the lane's 4 values of tile 2j and of tile 2j + 1 = keys 32 j + 4 g + e, 32 j + 16 + 4 g + e;
// the V^T fragment is read in that key order: two transposed reads of 4 keys x 16 d per 16-lane group.
template <int abl>
__device__ __forceinline__ float item_16(const char* smem, const bf16x8 (&qf)[2][2], int lane) {
  const int c = lane & 15, g = lane >> 4;
  const char* sK = smem + c * 128;
  int kch[2];
  for (int kk = 0; kk < 2; ++kk) kch[kk] = ((4 * kk + g) ^ ((c >> 1) & 7)) * 16;  // (rows 16 kb + c: (row >> 1) & 7 = (c >> 1) & 7)
  // transposed V read: lane 4q + p of a 16-lane group addresses key row q (of the group's 4 keys), d columns 4p .. 4p + 3 of a
  // 16-d block and receives d column (lane & 15) of the 4 keys
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int vkey = 4 * g + tq;  // + 32 j (+ 16): multiples of 16 keep (key >> 1) & 1
  const __attribute__((address_space(3))) char* sV = (const __attribute__((address_space(3))) char*)(smem + IMG) + vkey * 128 + 8 * (tp & 1);
  int vch[4];
  for (int db = 0; db < 4; ++db) vch[db] = ((2 * db + (tp >> 1)) ^ (((vkey >> 1) & 1) << 2)) * 16;
  const float sc = 0.125f * 1.44269504088896341f;
  f32x4 s[14][2];
  __builtin_amdgcn_s_setprio(0);
#pragma unroll
  for (int kb = 0; kb < 14; ++kb) {
    bf16x8 kf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(sK + kb * 16 * 128 + kch[kk]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[kb][qb][e] = (kb * 16 + 4 * g + e >= SEQ) ? -INFINITY : 0.0f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if constexpr ((abl & 4) != 0) s[kb][qb][kk] += (float)kf[kk][0];
        else s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kk], qf[qb][kk], s[kb][qb], 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_setprio(2);
  float nmb[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kb = 0; kb < 14; ++kb)
#pragma unroll
      for (int e = 0; e < 4; ++e) m4[e] = fmaxf(m4[e], s[kb][qb][e]);
    float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    nmb[qb] = -mx * sc;
  }
  f32x2 lsum2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
  f32x4 o[4][2];
  for (int db = 0; db < 4; ++db)
    for (int qb = 0; qb < 2; ++qb)
      for (int e = 0; e < 4; ++e) o[db][qb][e] = 0.0f;
  const f32x2 sc2 = {sc, sc};
#pragma unroll
  for (int j = 0; j < 7; ++j) {  // 32-key chunks; the last one holds keys 192 .. 196 (+ masked ones: P = 0)
    bf16x8 pf[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const f32x2 nm2 = {nmb[qb], nmb[qb]};
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          f32x2 x = {s[2 * j + h][qb][e], s[2 * j + h][qb][e + 1]};
          x = __builtin_elementwise_fma(x, sc2, nm2);
          f32x2 pv = {fast_exp2(x[0]), fast_exp2(x[1])};
          lsum2[qb] += pv;
          pf[qb][4 * h + e] = (bf16_t)pv[0];
          pf[qb][4 * h + e + 1] = (bf16_t)pv[1];
        }
    }
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + (32 * j) * 128 + vch[db]));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sV + (32 * j + 16) * 128 + vch[db]));
      bf16x8 vf;
      for (int e = 0; e < 4; ++e) {
        vf[e] = lo[e];
        vf[4 + e] = hi[e];
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if constexpr ((abl & 4) != 0) o[db][qb][j & 3] += (float)vf[0] * (float)pf[qb][0];
        else o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb], o[db][qb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float acc = 0.0f;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = lsum2[qb][0] + lsum2[qb][1];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    for (int db = 0; db < 4; ++db)
      for (int e = 0; e < 4; ++e) acc += o[db][qb][e] * inv;
  }
  return acc;
}

template <int SHAPE, int ABL = 0, int PF = 0>
__global__ __launch_bounds__(512) void attn_shape_kernel(int items, float* sink, unsigned long long* cycles, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fill_images(smem, tid, 512);
  __syncthreads();
  bf16x8 q32[4], q16[2][2];
  for (int ks = 0; ks < 4; ++ks)
    for (int e = 0; e < 8; ++e) q32[ks][e] = (bf16_t)(0.01f * (float)(((lane & 31) * 7 + ks * 16 + (lane >> 5) * 8 + e) % 41 - 20));
  for (int qb = 0; qb < 2; ++qb)
    for (int kk = 0; kk < 2; ++kk)
      for (int e = 0; e < 8; ++e) q16[qb][kk][e] = (bf16_t)(0.01f * (float)(((16 * qb + (lane & 15)) * 7 + 32 * kk + 8 * (lane >> 4) + e) % 41 - 20));
  float acc = 0.0f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  // mode bits 2-4 (shape 0 only): ablations of item_32 (4 no MFMAs, 8 no exponentials, 16 no LDS fragment reads)
  // mode bit 0: no barrier per item (the waves drift apart); bit 1: waves 4-6 -- the second wave of SIMDs 0-2 -- start half an
  // item late (with bit 0: they stay half an item behind their SIMD partner)
  if ((mode & 2) && wave >= 4) {
    const unsigned long long w0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - w0 < 3800) __builtin_amdgcn_s_sleep(4);
  }
  for (int it = 0; it < items; ++it) {
    if (!(mode & 1)) __builtin_amdgcn_s_barrier();  // (the kernel's one barrier per item)
    if (wave < 7) {
      if (SHAPE == 0) acc += item_32<ABL, PF>(smem, q32, lane);
      else acc += item_16<ABL>(smem, q16, lane);
    }
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(q32[ks]));
    for (int qb = 0; qb < 2; ++qb)
      for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(q16[qb][kk]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  sink[(size_t)blockIdx.x * 512 + tid] = acc;
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int items = argc > 1 ? atoi(argv[1]) : 48;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  int dev = 0;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, dev));
  const int grid = pr.multiProcessorCount;
  float* sink;
  unsigned long long* cyc;
  CK(hipMalloc(&sink, (size_t)grid * 512 * 4));
  CK(hipMalloc(&cyc, (size_t)grid * 8));
  const size_t lds = 2 * IMG;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<float> hs((size_t)grid * 512);
  std::vector<unsigned long long> hc(grid);
  typedef void (*kern_t)(int, float*, unsigned long long*, int);
  struct Case { const char* name; kern_t fn; };
  const Case cases[] = {{"32x32x16", attn_shape_kernel<0, 0>}, {"16x16x32", attn_shape_kernel<1, 0>},
                        {"32x32x16, no MFMAs", attn_shape_kernel<0, 4>}, {"16x16x32, no MFMAs", attn_shape_kernel<1, 4>}, {"32x32x16, no exponentials", attn_shape_kernel<0, 8>},
                        {"32x32x16, no LDS fragment reads", attn_shape_kernel<0, 16>}, {"32x32x16, no MFMAs, no exponentials", attn_shape_kernel<0, 12>},
                        {"32x32x16, no MFMAs, no LDS reads", attn_shape_kernel<0, 20>}, {"32x32x16, no exponentials, no LDS reads", attn_shape_kernel<0, 24>},
                        {"32x32x16, none of the three", attn_shape_kernel<0, 28>},
                        {"32x32x16, fragments one step ahead", attn_shape_kernel<0, 0, 1>},
                        {"32x32x16, K fragments two tiles ahead", attn_shape_kernel<0, 0, 2>},
                        {"32x32x16, no K fragment reads", attn_shape_kernel<0, 32>}, {"32x32x16, no V fragment reads", attn_shape_kernel<0, 64>}};
  for (const Case& cs : cases) {
    CK(hipFuncSetAttribute((const void*)cs.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double best = 1e30, sum0 = 0.0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(cs.fn, dim3(grid), dim3(512), lds, 0, items, sink, cyc, mode);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    CK(hipMemcpy(hs.data(), sink, hs.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 448; ++i) sum0 += hs[i];
    printf("mode %d %-40s: %.2f us per item, %llu shader cycles per item (workgroup %d); checksum %.6f\n", mode, cs.name, best * 1e3 / items,
           hc[grid / 2] / (unsigned long long)items, grid / 2, sum0);
  }
  return 0;
}
