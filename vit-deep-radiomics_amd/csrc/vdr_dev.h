// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes, CDNA4 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define VDR_DEV static __device__ __forceinline__

// global -> LDS direct copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
VDR_DEV void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

VDR_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

VDR_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
VDR_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)) (activation="gelu", models_archs.py:133).
// erfc(|z|) by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), used on the complement side for
// negative x so the tail keeps its relative accuracy.
// Written branch-free as  gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2)  (identical for both
// signs), which needs no compare/select and keeps the VALU count of the GEMM epilogue low.
VDR_DEV float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float z = ax * 0.70710678118654752f;
  const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = fast_exp2(z * z * -1.44269504088896341f);
  const float w = (poly * t) * (e * (0.5f * ax));  // 0.5 |x| erfc(|z|)
  return fmaxf(x, 0.0f) - w;
}

VDR_DEV float silu(float x) { return x * fast_rcp(1.0f + fast_exp2(-x * 1.44269504088896341f)); }

// XCD-aware bijective remap of a 1-D workgroup id: workgroups that share an XCD (id % 8) get a
// contiguous range of logical ids, so neighbouring tiles hit the same private L2.
VDR_DEV int xcd_remap(int id, int nwg) {
  const int xcd = id & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (id >> 3);
}
