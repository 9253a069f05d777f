// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes, CDNA4 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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

// The same copy as one opaque instruction: wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset, LDS
// destination = wave-uniform byte address `lds_addr`, handed over in M0 (+ lane*16 by the hardware).  hipcc does not see a memory
// operation here, which is the point: with the builtin it makes every later LDS read that it cannot prove disjoint
// from the DMA's destination (all ds_read_b64_tr_b16 reads, for one) wait `vmcnt(0)` first -- i.e. for the whole
// prefetch that was meant to stay in flight under the arithmetic.  The caller orders the data itself: a counted
// `s_waitcnt vmcnt` + workgroup barrier before the first read of the image (cdna_hip_programming.md §7).
VDR_DEV void glds16_raw(const void* base_uniform, uint32_t lane_off, const void* lds_wave_base) {
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"  // (s_nop: one wait state between the write of M0 and its use)
               :: "v"(lane_off), "s"(base_uniform), "{m0}"(lds_addr) : "memory");
}

// ... and with a per-lane 64-bit source address (no uniform base at hand)
VDR_DEV void glds16_raw(const void* gsrc, const void* lds_wave_base) {
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "{m0}"(lds_addr) : "memory");
}

VDR_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

VDR_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
VDR_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// erf-GELU, 0.5 x (1 + erf(x / sqrt 2)) (activation="gelu", models_archs.py:133; timm / DINOv2 Mlp).
// Branch-free as  gelu(x) = max(x, 0) - w(|x|),  w(a) = a Phi(-a) = a 2^Q(a)  with  Q(a) = log2(0.5 erfc(a / sqrt 2))
// as a degree-6 minimax polynomial on [0, 5.7] (fitted against scipy.special.erfc in float64: |dQ| <= 5.0e-5, i.e.
// w is within 3.5e-5 RELATIVE of the exact value everywhere it exceeds 2e-7, absolute error <= 4.8e-6 at w ~ 0.17;
// beyond a = 5.7 w < 6e-8 and a is clamped).  The output of every epilogue that calls this is rounded to bf16 (half an
// ulp = 2e-3 relative), 55 x coarser than the approximation, so results are those of the exact function to one bf16
// rounding.  Cost: 10 VALU instructions + one v_exp_f32 per value; the earlier Abramowitz-Stegun 7.1.26 form (1.5e-7
// absolute) took 14 + v_rcp + v_exp, and the GELU is VALU-throughput bound: 0.05 ms of a 0.27 ms fc1 launch.
// NaN: v_min / v_max return their non-NaN operand, so gelu_erf(NaN) = -3e-8, not NaN (torch's gelu propagates it).  A
// NaN-preserving form needs a compare + select (2 more instructions per value) -- x + |x| for 2 max(x, 0) breaks -inf,
// 0 * x breaks +inf -- in an epilogue that is bound by vector issue.  Inside a transformer block the residual stream
// carries a NaN row on regardless; the one consumer without a residual, the classifier heads (vdr/model.py _MlpHead),
// hands non-finite feature rows on itself.
VDR_DEV float gelu_erf(float x) {
  // (v_min / v_max written out: fminf / fmaxf make hipcc canonicalise their operands first, one more v_max each)
  float a, relu;
  asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(a) : "v"(x), "v"(5.7f));
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(relu) : "v"(x));
  float q = fmaf(2.480073296e-05f, a, -6.399250922e-04f);
  q = fmaf(q, a, 7.365777341e-03f);
  q = fmaf(q, a, -5.164207073e-02f);
  q = fmaf(q, a, -4.607286841e-01f);
  q = fmaf(q, a, -1.150403490e+00f);
  q = fmaf(q, a, -1.000050145e+00f);
  return fmaf(-a, fast_exp2(q), relu);
}

// The same function on a pair of values, bit for bit (every step is the IEEE operation of the scalar form): the polynomial and
// the last multiply-add as v_pk_fma_f32 -- two values per issue slot at the scalar instruction's rate
// (profiles/r03_valu_rate_micro.txt) -- so a pair costs 2 min + 2 max + 7 packed + 2 exp slots instead of 18 + 2 exp.  The
// GELU epilogues are bound by vector issue.
VDR_DEV f32x2 gelu_erf2(f32x2 x) {
  f32x2 a, relu;
  asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(a[0]) : "v"(x[0]), "v"(5.7f));
  asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(a[1]) : "v"(x[1]), "v"(5.7f));
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(relu[0]) : "v"(x[0]));
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(relu[1]) : "v"(x[1]));
  const auto k = [](float c) { return f32x2{c, c}; };
  f32x2 q = __builtin_elementwise_fma(k(2.480073296e-05f), a, k(-6.399250922e-04f));
  q = __builtin_elementwise_fma(q, a, k(7.365777341e-03f));
  q = __builtin_elementwise_fma(q, a, k(-5.164207073e-02f));
  q = __builtin_elementwise_fma(q, a, k(-4.607286841e-01f));
  q = __builtin_elementwise_fma(q, a, k(-1.150403490e+00f));
  q = __builtin_elementwise_fma(q, a, k(-1.000050145e+00f));
  const f32x2 e = {fast_exp2(q[0]), fast_exp2(q[1])};
  return __builtin_elementwise_fma(-a, e, relu);
}

// One 16-key slice of a softmax row held in the S^T accumulator layout: P = 2^(s * sc + nmb) for the 8 scores of this
// lane, their sum added to `lsum2`, P rounded to bf16 as the B operand of O^T += V^T . P^T.  Written on PAIRS of
// neighbouring accumulator registers so that the scale-and-shift and the row sum are one packed instruction per pair
// (v_pk_fma_f32, v_pk_add_f32: two lanes of fp32 per issue slot): 5 vector instructions per pair instead of 7, and the
// attention kernels are bound by vector issue.  The row sum is kept as (even elements, odd elements) and folded once at
// the end of the row; every attention kernel sums in this order, so their outputs stay bitwise equal to each other.
VDR_DEV void softmax_slice8(const f32x16& st, int s2, float sc, float nmb, f32x2& lsum2, bf16x8& pf) {
  const f32x2 sc2 = {sc, sc}, nmb2 = {nmb, nmb};
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f32x2 x = {st[8 * s2 + j], st[8 * s2 + j + 1]};
    x = __builtin_elementwise_fma(x, sc2, nmb2);
    f32x2 pv;
    pv[0] = fast_exp2(x[0]);
    pv[1] = fast_exp2(x[1]);
    lsum2 += pv;
    pf[j] = (bf16_t)pv[0];
    pf[j + 1] = (bf16_t)pv[1];
  }
}

// row maximum of NT accumulator tiles as four independent chains (a single chain of v_max3 is 8 NT dependent instructions)
template <int NT>
VDR_DEV float row_max_tiles(const f32x16 (&s)[NT]) {
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) m[(e >> 1) & 3] = fmaxf(m[(e >> 1) & 3], s[t][e]);
  return fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
}

VDR_DEV float silu(float x) { return x * fast_rcp(1.0f + fast_exp2(-x * 1.44269504088896341f)); }
// Key mask of one 32-key tile of the attention kernels: accumulator element e of lane (.., hh) is key
// k0 + (e & 3) + 8 (e >> 2) + 4 hh; keys >= len get -inf.  Written as (constant >= per-lane threshold) with the threshold
// made opaque on every call: comparing `key >= len` directly let hipcc hoist all 16 x NT lane masks out of the item loop
// (they depend on the lane and the sequence length only), keep them in SGPR pairs, run out of SGPRs and spill them to
// VGPR lanes -- 229 v_writelane + 229 v_readlane + 164 s_nop per pass of the persistent kernel's loop.  Call it inside a
// wave-uniform `if (k0 + 32 > len)`: the asm statement also keeps that a real branch.
VDR_DEV void mask_keys(f32x16& s, int k0, int hh, int len) {
  int thr = len - k0 - 4 * hh;
  asm volatile("" : "+v"(thr));
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if ((e & 3) + 8 * (e >> 2) >= thr) s[e] = -INFINITY;
}

// One output row of the attention kernels (64 head dims, bf16): in the S^T / O^T accumulator layout lane (row, hh) holds
// dims nd*32 + 8g + 4hh + e (e = 0..3) and its partner lane (xor 32) the other 4 of every 8.  One v_permlane32_swap per
// dword gives the lower lane dims 8g .. 8g+7 and the upper lane 8(g+1) .. 8(g+1)+7 of each group pair: 4 stores of
// 16 B per lane instead of 8 of 8 B (a store instruction costs the same issue slot whatever its width).  Every lane
// must execute this (cross-lane exchange); `ok` only gates the stores.
VDR_DEV void store_row64_bf16(bf16_t* dst, const f32x16 (&o)[2], float inv, int hh, bool ok) {
#pragma unroll
  for (int nd = 0; nd < 2; ++nd)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      uint32_t a[2], b[2];  // a: group 2gp, b: group 2gp+1, each 4 bf16 = 2 dwords
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        bf16x2 va, vb;
        va[0] = (bf16_t)(o[nd][8 * gp + 2 * d] * inv);
        va[1] = (bf16_t)(o[nd][8 * gp + 2 * d + 1] * inv);
        vb[0] = (bf16_t)(o[nd][8 * gp + 4 + 2 * d] * inv);
        vb[1] = (bf16_t)(o[nd][8 * gp + 4 + 2 * d + 1] * inv);
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, va), __builtin_bit_cast(uint32_t, vb), false, false);
        a[d] = r[0];
        b[d] = r[1];
      }
      u32x4 v;
      v[0] = a[0];
      v[1] = a[1];
      v[2] = b[0];
      v[3] = b[1];
      if (ok) *reinterpret_cast<u32x4*>(dst + nd * 32 + 8 * (2 * gp + hh)) = v;
    }
}

// XCD-aware bijective remap of a 1-D workgroup id: workgroups that share an XCD (id % 8) get a
// contiguous range of logical ids, so neighbouring tiles hit the same private L2.
VDR_DEV int xcd_remap(int id, int nwg) {
  const int xcd = id & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (id >> 3);
}
