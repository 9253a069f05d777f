// PARKED EXPERIMENT (round 3; was csrc/patch_fused.hip, vdr_op_patch_embed_fused / vdr_config.patch_fusion of ABI 6): bitwise
// equal to im2col + GEMM and 1.6-2.1x slower (profiles/r03_patch_embed_fused_gbs.json) -- no longer compiled into libvdr.so.
// Known issue if it is ever revived (round-3 advisor): the dynamic-LDS opt-in is remembered as a per-device bool although
// lds = 32 (2 Kp + 32) varies at run time -- a first call with a smaller Kp leaves the attribute too low for a later one
// (loud launch failure); keep the largest size set per device, as launch_cfg in gemm_kernels.h does.
//
// Patchify convolution in ONE launch for patch sides whose pixel runs are not 16-byte chunks (p = 14: DINOv2 /
// ViT-L/14 / ViT-g/14) and for fp32 pixels of any even p (the reference's own input dtype):
//   Conv2d(C, D, kernel = p, stride = p) on NCHW images -> token rows (+ bias, + pos_embed, bf16 or fp32, optional
//   LayerNorm partial sums), i.e. `model.patch_embed(x)` of src/tfds_dense_descriptor.py:128 and the first operator of
//   every ViT forward called at src/tfds_dense_descriptor.py:123.
// It replaces im2col_rows_kernel (rowops.hip) + the EPI_PATCH GEMM for those cases: the im2col rows of a workgroup's
// patches never leave the CU.  (bf16 pixels with p in {8, 16, 32} keep the in-loader gather of the ring4 GEMM.)
//
// A workgroup = up to 32 neighbouring patches of one patch row of one image, 8 waves:
//   1. the C p pixel rows of the run (contiguous, 4 or 8 bytes per lane: whole lines) are converted to bf16 and scattered
//      into an LDS image [patch][Kp] (k = c p^2 + ky p + kx, zero padded to Kp) -- the A operand;
//   2. wave w multiplies the image with 64 output columns (512 per pass, ceil(D / 512) passes): B fragments come
//      straight from W [D][Kp] in global memory (every wave owns its own columns: nothing to share through LDS; W is
//      L2-resident), two 32-deep steps ahead; v_mfma_f32_16x16x32_bf16, issued transposed;
//   3. epilogue in the accumulator layout.  W row slot i of column tile jt is column 32 (jt >> 1) + 8 (i >> 2) + 4 (jt & 1)
//      + (i & 3) (as gemm_stream.hip), so a lane holds 8 consecutive columns per tile pair: 16-byte stores, and the
//      LayerNorm partials (sum, sum of squares of the ROUNDED outputs per row and 64-column group) are summed in the
//      order of the GEMM epilogue they replace (gemm_epi.h: octets in sequence, then (o0+o4 + o2+o6) + (o1+o5 + o3+o7)).
// Same products in the same order (32-deep units ascending), same epilogue arithmetic ((acc + bias) + pos): bitwise
// equal to im2col + GEMM (tests/test_ops_gpu.py::test_patch_embed_fused_equals_two_launches).
// HBM-bound on paper (pixels in + tokens out; the GEMM is 32 GFLOP for 16 slices of 896^2 at D = 384) -- MEASURED SLOWER
// than the two launches it replaces (reference dinov2 mode, fp32 896^2 -> fp32: B = 1 24 vs 21.6 us, B = 16 165-220 vs
// 105 us, profiles/r03_patch_embed_gbs.json): a workgroup is three dependent latency-bound phases (pixel rows from HBM,
// 20 steps of B fragments from L2 at 4-8 MFMAs each, the stores) and every workgroup re-reads all of W (0.5 MB: 1 GB
// through the CUs' address paths at B = 16).  Deeper register prefetch (4 steps: 134 VGPRs, one workgroup per CU) and all
// pixel rows in flight made it slower.  Kept as vdr_config.patch_fusion = 1 / vdr_op_patch_embed_fused, off by default.
#include "gemm_epi.h"

namespace vdr {

struct PatchFusedK {
  const void* images;
  const bf16_t* W;    // [D][Kp], zero padded
  const float* bias;  // [D] or null
  const float* pos;   // [tokens][D] fp32 or null
  void* out;          // bf16 or fp32 rows of D
  float* ln_part;     // [D / 64][part_stride][2] or null
  int64_t part_stride;
  int C, img, p, g, Kp, D;
  int tpb, blocks_per_row;
  int rpg;            // output row of patch m: (m / rpg) * gstride + off + m % rpg ; pos row: off + m % rpg
  int64_t gstride;
  int off;
  int out_f32;
};

template <bool IN_BF16>
__global__ __launch_bounds__(512, 2) void patch_fused_kernel(PatchFusedK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int RS = p.Kp * 2 + 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int xb = bid % p.blocks_per_row;
  bid /= p.blocks_per_row;
  const int py = bid % p.g;
  const int b = bid / p.g;
  const int t0 = xb * p.tpb;
  const int nt = min(p.tpb, p.g - t0);  // patches of this workgroup
  const int P = p.p, pp = P * P, Kreal = p.C * pp;

  // ---- 1. the A image ------------------------------------------------------------------------------------------------
  {
    const int padw = (p.Kp - Kreal) >> 1;  // dwords of K padding per row
    for (int i = tid; i < 32 * padw; i += 512) {
      const int r = i / padw, j = i - r * padw;
      *reinterpret_cast<uint32_t*>(smem + r * RS + (Kreal + 2 * j) * 2) = 0u;
    }
    // rows of patches past nt: zero (their products are never stored, but NaN-free)
    const int kdw = Kreal >> 1;
    for (int i = tid; i < (32 - nt) * kdw; i += 512) {
      const int r = nt + i / kdw, j = i % kdw;
      *reinterpret_cast<uint32_t*>(smem + r * RS + 4 * j) = 0u;
    }
    // a thread owns pixel pair j of the run (its patch and kx fixed once) and one half of the C p image rows
    const int ppr = (nt * P) >> 1;
    const int nrows = p.C * P;
    const int half = tid >= 256 ? 1 : 0, j = tid & 255;
    const int r_lo = half ? (nrows + 1) / 2 : 0, r_hi = half ? nrows : (nrows + 1) / 2;
    if (j < ppr) {
      const int x = 2 * j;
      const int t = x / P, kx = x - t * P;
      char* dst = smem + t * RS + kx * 2;
      const int64_t img_base = ((int64_t)b * p.C * p.img + py * P) * p.img + t0 * P + x;
      auto load = [&](int row) {  // row = c p + ky
        const int c = row / P, ky = row - c * P;
        const int64_t src = img_base + ((int64_t)c * p.img + ky) * p.img;
        bf16x2 v;
        if (IN_BF16) {
          v = *reinterpret_cast<const bf16x2*>((const bf16_t*)p.images + src);
        } else {
          const float2 f = *reinterpret_cast<const float2*>((const float*)p.images + src);
          v[0] = (bf16_t)f.x;
          v[1] = (bf16_t)f.y;
        }
        return v;
      };
      auto put = [&](int row, bf16x2 v) {
        const int c = row / P, ky = row - c * P;
        *reinterpret_cast<bf16x2*>(dst + (c * pp + ky * P) * 2) = v;
      };
      int row = r_lo;
      for (; row + 7 <= r_hi; row += 7) {  // 7 image rows in flight (all 21 of a half at once measured slower: 165 -> 192 us at B = 16)
        bf16x2 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = load(row + u);
#pragma unroll
        for (int u = 0; u < 7; ++u) put(row + u, v[u]);
      }
      for (; row < r_hi; ++row) put(row, load(row));
    }
  }
  __syncthreads();

  // ---- 2. + 3. ---------------------------------------------------------------------------------------------------------
  const int r15 = lane & 15, q4 = lane >> 4;
  const int nsteps = p.Kp >> 5;
  const bool two_tiles = nt > 16;  // (wave-uniform)
  const int rowl = ((r15 >> 2) << 3) + (r15 & 3);  // W row slot -> column permutation (see above)
  for (int n0 = wave * 64; n0 < p.D; n0 += 512) {
    const bf16_t* wsrc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      int n = n0 + 32 * (jt >> 1) + 4 * (jt & 1) + rowl;
      n = n < p.D ? n : p.D - 1;  // (columns past D: a valid row, never stored)
      wsrc[jt] = p.W + (int64_t)n * p.Kp + 8 * q4;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragments run PD steps ahead of their MFMAs in a register ring (they come from L2 with several hundred cycles
    // of latency and feed only 4-8 MFMAs each: one step of distance left the loop waiting for every step's loads)
    constexpr int PD = 2;
    bf16x8 fb[PD][4];
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) fb[d][jt] = *reinterpret_cast<const bf16x8*>(wsrc[jt] + (d < nsteps ? d : nsteps - 1) * 32);
    const char* arow = smem + r15 * RS + q4 * 16;
    auto one_step = [&](int s, bf16x8 (&f)[4]) {
      const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(arow + s * 64);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[0][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[jt], fa0, acc[0][jt], 0, 0, 0);
      if (two_tiles) {
        const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(arow + 16 * RS + s * 64);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[1][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[jt], fa1, acc[1][jt], 0, 0, 0);
      }
      const int sn = s + PD < nsteps ? s + PD : nsteps - 1;  // (past the end: a valid address, the value is not used)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) f[jt] = *reinterpret_cast<const bf16x8*>(wsrc[jt] + sn * 32);
    };
    int s = 0;
    for (; s + PD <= nsteps; s += PD) {
#pragma unroll
      for (int d = 0; d < PD; ++d) one_step(s + d, fb[d]);
    }
#pragma unroll
    for (int d = 0; d < PD; ++d)
      if (s + d < nsteps) one_step(s + d, fb[d]);
    // epilogue: lane (r15, q4) holds, of patch row 16 it + r15, columns n0 + 32 jp + 8 q4 + (0..7) for jp = 0, 1
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it == 1 && !two_tiles) break;
      const int r = it * 16 + r15;
      const bool rok = r < nt;
      const int64_t m = ((int64_t)b * p.g + py) * p.g + t0 + (rok ? r : 0);
      const int64_t gi = m / p.rpg;
      const int ii = (int)(m - gi * p.rpg);
      const int64_t orow = gi * p.gstride + p.off + ii;
      const int prow = p.off + ii;
      float s1[2], s2[2];
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int n = n0 + 32 * jp + 8 * q4;
        const bool ok = rok && n < p.D;
        const int nn = n < p.D ? n : 0;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[it][2 * jp][e];
          v[4 + e] = acc[it][2 * jp + 1][e];
        }
        if (p.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + nn), b1 = *reinterpret_cast<const f32x4*>(p.bias + nn + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
          }
        }
        if (p.pos) {
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.D + nn);
          const f32x4 p1 = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.D + nn + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += p0[e];
            v[4 + e] += p1[e];
          }
        }
        s1[jp] = s2[jp] = 0.0f;
        if (p.out_f32) {
          if (ok) {
            float* dst = reinterpret_cast<float*>(p.out) + orow * p.D + n;
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
        } else {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = (bf16_t)v[e];
            const float rr = (float)o[e];  // statistics of what the consumer will actually read
            s1[jp] += rr;
            s2[jp] = fmaf(rr, rr, s2[jp]);
          }
          if (ok) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.out) + orow * p.D + n) = o;
          if (!ok) s1[jp] = s2[jp] = 0.0f;
        }
      }
      if (p.ln_part) {  // (wave-uniform)
        // octets q4 (jp 0) and 4 + q4 (jp 1) are in this lane; q4 ^ 2 is lane ^ 32, q4 ^ 1 is lane ^ 16
        float a = s1[0] + s1[1], c = s2[0] + s2[1];
        a += __shfl_xor(a, 32, 64);
        c += __shfl_xor(c, 32, 64);
        a += __shfl_xor(a, 16, 64);
        c += __shfl_xor(c, 16, 64);
        if (q4 == 0 && rok && n0 < p.D) {
          float* dst = p.ln_part + ((int64_t)(n0 >> 6) * p.part_stride + orow) * 2;
          dst[0] = a;
          dst[1] = c;
        }
      }
    }
  }
}

// whether the fused kernel takes this patchify (else im2col + GEMM): an even patch side (a pixel pair never straddles two
// patches), 8-byte aligned pixel pairs, the LDS image of 32 patches within half a CU
bool patch_fused_ok(const void* images, int in_bf16, int img, int p, int Kp, int D) {
  if (p <= 0 || (p & 1) || (img & 1) || img % p || (Kp & 31) || (D & 7)) return false;
  if (((uintptr_t)images) & 7) return false;
  return (size_t)32 * (Kp * 2 + 32) <= 80 * 1024;
}

hipError_t launch_patch_fused(const void* images, int in_bf16, const void* W, const float* bias, const float* pos, void* out,
                              int out_f32, float* ln_part, int64_t part_stride, int batch, int C, int img, int p, int Kp, int D,
                              RowMap omap, hipStream_t s) {
  if (!patch_fused_ok(images, in_bf16, img, p, Kp, D) || batch <= 0 || Kp < C * p * p) return hipErrorInvalidValue;
  if (ln_part && (D & 63)) return hipErrorInvalidValue;
  PatchFusedK k{};
  k.images = images;
  k.W = (const bf16_t*)W;
  k.bias = bias;
  k.pos = pos;
  k.out = out;
  k.ln_part = out_f32 ? nullptr : ln_part;
  k.part_stride = part_stride;
  k.C = C;
  k.img = img;
  k.p = p;
  k.g = img / p;
  k.Kp = Kp;
  k.D = D;
  k.tpb = k.g < 32 ? k.g : 32;
  k.blocks_per_row = (k.g + k.tpb - 1) / k.tpb;
  k.rpg = omap.rpg;
  k.gstride = omap.gstride;
  k.off = omap.off;
  k.out_f32 = out_f32;
  const size_t lds = (size_t)32 * (Kp * 2 + 32);
  const int dev = current_device_index();
  if (dev < 0) return hipErrorInvalidDevice;
  const dim3 grid((unsigned)((int64_t)batch * k.g * k.blocks_per_row)), block(512);
  if (in_bf16) {
    static PerDeviceFlag attr;
    if (lds > 65536 && !attr.done[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)patch_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr.done[dev] = true;
    }
    hipLaunchKernelGGL((patch_fused_kernel<true>), grid, block, lds, s, k);
  } else {
    static PerDeviceFlag attr;
    if (lds > 65536 && !attr.done[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)patch_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr.done[dev] = true;
    }
    hipLaunchKernelGGL((patch_fused_kernel<false>), grid, block, lds, s, k);
  }
  return hipGetLastError();
}

}  // namespace vdr
