// HBM-bound row kernels for gfx950: LayerNorm (+ row gather / CLS extraction), im2col for the
// patchify conv, CLS-row and token assembly.
//
// LayerNorm replaces nn.LayerNorm (reference src/models_archs.py:136,145; norm1/norm2/norm of the
// ViT blocks): biased variance, fp32 statistics, eps configurable.  One 64-lane wave per row, the
// whole row held in registers (4 elements per lane per pass), two-pass mean/variance with
// wave-wide shuffle reductions — 2*D*sizeof bytes of HBM traffic per row and nothing else.
// The row maps fold the x[:,0,:] CLS slice (models_archs.py:147) and the x[:,1:,:] dense slice
// (tfds_dense_descriptor.py:130-133) into the final LayerNorm.
#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

struct LnK {
  const void* x;
  void* y;
  const float* gamma;
  const float* beta;
  const float* cls;
  int64_t rows;
  int D;
  float eps;
  int irpg;
  int64_t igs;
  int ioff;
  int orpg;
  int64_t ogs;
  int ooff;
  int cls_period;
  int win_ws, win_g;
};

VDR_DEV int64_t map_row(int64_t r, int rpg, int64_t gs, int off) {
  if (rpg >= (1 << 30)) return r;
  const int64_t g = r / rpg;
  return g * gs + off + (r - g * rpg);
}

template <bool IN_BF16, bool OUT_BF16, int NP>
__global__ __launch_bounds__(256) void layernorm_kernel(LnK p) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  const int64_t ir = map_row(r, p.irpg, p.igs, p.ioff);
  int64_t orow = map_row(r, p.orpg, p.ogs, p.ooff);
  if (p.win_ws > 0) {
    // window partition (segment_anything window_partition): token (b, y, x) -> window-major order
    const int g = p.win_g, ws = p.win_ws, nw = (g + ws - 1) / ws;
    const int64_t b = r / (g * g);
    const int rem = (int)(r - b * g * g);
    const int y = rem / g, x = rem - y * g;
    orow = ((b * nw + y / ws) * nw + x / ws) * (int64_t)(ws * ws) + (y % ws) * ws + (x % ws);
  }
  const bool from_cls = p.cls != nullptr && (r % p.cls_period) == 0;
  float v[NP][4];
  float sum = 0.0f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < p.D) {
      if (from_cls) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.cls + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = t[e];
      } else if (IN_BF16) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>((const bf16_t*)p.x + ir * p.D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = (float)t[e];
      } else {
        const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)p.x + ir * p.D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = t[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += v[k][e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[k][e] = 0.0f;
    }
  }
  const float invD = 1.0f / (float)p.D;
  const float mean = wave_sum(sum) * invD;
  float sq = 0.0f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < p.D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[k][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) * invD + p.eps);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < p.D) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mean) * rstd * g[e] + bt[e];
      if (OUT_BF16) {
        bf16x4 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) ob[e] = (bf16_t)o[e];
        *reinterpret_cast<bf16x4*>((bf16_t*)p.y + orow * p.D + c) = ob;
      } else {
        f32x4 of;
#pragma unroll
        for (int e = 0; e < 4; ++e) of[e] = o[e];
        *reinterpret_cast<f32x4*>((float*)p.y + orow * p.D + c) = of;
      }
    }
  }
}

template <bool IB, bool OB>
static hipError_t ln_dispatch(const LnK& k, hipStream_t s) {
  const int np = (k.D + 255) / 256;
  const dim3 grid((unsigned)((k.rows + 3) / 4)), block(256);
#define VDR_LN(NP)                                                              \
  case NP:                                                                      \
    hipLaunchKernelGGL((layernorm_kernel<IB, OB, NP>), grid, block, 0, s, k);   \
    break;
  switch (np) {
    VDR_LN(1) VDR_LN(2) VDR_LN(3) VDR_LN(4) VDR_LN(5) VDR_LN(6) VDR_LN(7) VDR_LN(8)
    default:
      return hipErrorInvalidValue;
  }
#undef VDR_LN
  return hipGetLastError();
}

hipError_t launch_layernorm(const LnArgs& a, hipStream_t s) {
  if (a.rows <= 0 || a.D <= 0 || (a.D & 3) || a.D > 2048) return hipErrorInvalidValue;
  LnK k;
  k.x = a.x;
  k.y = a.y;
  k.gamma = a.gamma;
  k.beta = a.beta;
  k.cls = a.cls;
  k.rows = a.rows;
  k.D = a.D;
  k.eps = a.eps;
  k.irpg = a.imap.rpg;
  k.igs = a.imap.gstride;
  k.ioff = a.imap.off;
  k.orpg = a.omap.rpg;
  k.ogs = a.omap.gstride;
  k.ooff = a.omap.off;
  k.cls_period = a.cls_period > 0 ? a.cls_period : 1;
  k.win_ws = a.win_ws;
  k.win_g = a.win_g;
  if (a.in_bf16) return a.out_bf16 ? ln_dispatch<true, true>(k, s) : ln_dispatch<true, false>(k, s);
  return a.out_bf16 ? ln_dispatch<false, true>(k, s) : ln_dispatch<false, false>(k, s);
}

// ---------------------------------------------------------------------------------------------
// im2col for Conv2d(C, D, kernel=p, stride=p): col[b*n + py*g + px][c*p*p + ky*p + kx]
// One thread writes 8 consecutive k (one 16-byte store).
// ---------------------------------------------------------------------------------------------
template <bool IN_BF16>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ images, bf16_t* __restrict__ col,
                                                     int64_t total8, int C, int img, int p, int g, int Kp, int fast) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total8) return;
  const int k8 = Kp >> 3;
  const int64_t row = idx / k8;
  const int kk = (int)(idx - row * k8) * 8;
  const int n = g * g;
  const int64_t b = row / n;
  const int pi = (int)(row - b * n);
  const int py = pi / g, px = pi - py * g;
  const int pp = p * p;
  const int Kreal = C * pp;
  bf16x8 o;
  if (fast && kk < Kreal) {
    const int c = kk / pp;
    const int rem = kk - c * pp;
    const int ky = rem / p, kx = rem - ky * p;
    const int64_t src = ((b * C + c) * img + (py * p + ky)) * (int64_t)img + px * p + kx;
    if (IN_BF16) {
      o = *reinterpret_cast<const bf16x8*>((const bf16_t*)images + src);
    } else {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>((const float*)images + src);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>((const float*)images + src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (bf16_t)a0[e];
        o[4 + e] = (bf16_t)a1[e];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kk + e;
      float val = 0.0f;
      if (k < Kreal) {
        const int c = k / pp;
        const int rem = k - c * pp;
        const int ky = rem / p, kx = rem - ky * p;
        const int64_t src = ((b * C + c) * img + (py * p + ky)) * (int64_t)img + px * p + kx;
        val = IN_BF16 ? (float)((const bf16_t*)images)[src] : ((const float*)images)[src];
      }
      o[e] = (bf16_t)val;
    }
  }
  *reinterpret_cast<bf16x8*>(col + row * Kp + kk) = o;
}

// The same matrix for patch sides that are not a multiple of 8 (p = 14: DINOv2, ViT-L/14, ViT-g/14), through LDS: a
// workgroup takes up to 32 neighbouring patches of one patch row -- in the image that is C*p runs of 32*p contiguous
// pixels, read as coalesced pairs (p even: a pair never straddles two patches), converted and scattered into an LDS
// image [patch][Kp] (row stride Kp*2 + 32 bytes: consecutive patches start 8 banks apart) -- and writes the rows out as
// whole 16-byte chunks.  (The direct kernel above needs 8 scalar loads and 8 index divisions per 16-byte store here:
// 2.0 TB/s of pixels + rows in the reference's dinov2 mode; this form: see DESIGN.md §6 f-2.)
template <bool IN_BF16>
__global__ __launch_bounds__(256) void im2col_rows_kernel(const void* __restrict__ images, bf16_t* __restrict__ col, int C,
                                                          int img, int p, int g, int Kp, int tpb, int blocks_per_row) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int RS = Kp * 2 + 32;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int xb = bid % blocks_per_row;
  bid /= blocks_per_row;
  const int py = bid % g;
  const int b = bid / g;
  const int t0 = xb * tpb;
  const int nt = min(tpb, g - t0);  // patches of this workgroup
  const int pp = p * p, Kreal = C * pp;
  // zero the K padding
  const int padw = (Kp - Kreal) >> 1;  // dwords per row (Kreal and Kp are even)
  for (int i = tid; i < nt * padw; i += 256) {
    const int r = i / padw, j = i - r * padw;
    *reinterpret_cast<uint32_t*>(smem + r * RS + (Kreal + 2 * j) * 2) = 0u;
  }
  // pixels: a thread owns pixel pair j of the workgroup's run (its patch and kx fixed once) and walks the C*p image rows
  const int ppr = (nt * p) >> 1;
  const int64_t img_base = ((int64_t)b * C * img + py * p) * img + t0 * p;
  for (int j = tid; j < ppr; j += 256) {
    const int x = 2 * j;
    const int t = x / p, kx = x - t * p;
    char* dst = smem + t * RS + kx * 2;
    auto load = [&](int64_t src) {
      bf16x2 v;
      if (IN_BF16) {
        v = *reinterpret_cast<const bf16x2*>((const bf16_t*)images + src);
      } else {
        const float2 f = *reinterpret_cast<const float2*>((const float*)images + src);
        v[0] = (bf16_t)f.x;
        v[1] = (bf16_t)f.y;
      }
      return v;
    };
    for (int c = 0; c < C; ++c) {
      const int64_t src = img_base + (int64_t)c * img * img + x;
      char* d = dst + c * pp * 2;
      int ky = 0;
      for (; ky + 7 <= p; ky += 7) {  // 7 rows in flight (p = 14: two rounds)
        bf16x2 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = load(src + (int64_t)(ky + u) * img);
#pragma unroll
        for (int u = 0; u < 7; ++u) *reinterpret_cast<bf16x2*>(d + (ky + u) * p * 2) = v[u];
      }
      for (; ky < p; ++ky) *reinterpret_cast<bf16x2*>(d + ky * p * 2) = load(src + (int64_t)ky * img);
    }
  }
  __syncthreads();
  const int k8 = Kp >> 3;
  const int64_t row0 = ((int64_t)b * g + py) * g + t0;
  for (int i = tid; i < nt * k8; i += 256) {
    const int r = i / k8, ch = i - r * k8;
    *reinterpret_cast<bf16x8*>(col + (row0 + r) * Kp + ch * 8) = *reinterpret_cast<const bf16x8*>(smem + r * RS + ch * 16);
  }
}

hipError_t launch_im2col(const void* images, int in_bf16, void* col, int batch, int C, int img, int p,
                         int Kp, hipStream_t s) {
  if (batch <= 0 || p <= 0 || img % p || (Kp & 63) || Kp < C * p * p) return hipErrorInvalidValue;
  if ((p & 7) && !(p & 1) && !(img & 1) && (((uintptr_t)images) & 7) == 0) {
    // even patch side that is not a multiple of 8: the LDS form
    const int g = img / p;
    const int tpb = g < 32 ? g : 32;
    const int bpr = (g + tpb - 1) / tpb;
    const size_t lds = (size_t)tpb * (Kp * 2 + 32);
    if (lds <= 65536) {
      const dim3 grid((unsigned)((int64_t)batch * g * bpr)), block(256);
      if (in_bf16)
        hipLaunchKernelGGL((im2col_rows_kernel<true>), grid, block, lds, s, images, (bf16_t*)col, C, img, p, g, Kp, tpb, bpr);
      else
        hipLaunchKernelGGL((im2col_rows_kernel<false>), grid, block, lds, s, images, (bf16_t*)col, C, img, p, g, Kp, tpb, bpr);
      return hipGetLastError();
    }
  }
  // the fast path reads 8 pixels with vector loads: needs p % 8 == 0 and img % 8 == 0 so that every
  // 8-pixel run starts 16-byte aligned (the image base is assumed 16-byte aligned)
  const int g = img / p;
  const int fast = ((p & 7) == 0 && (img & 7) == 0 && (((uintptr_t)images) & 15) == 0) ? 1 : 0;
  const int64_t total8 = (int64_t)batch * g * g * (Kp / 8);
  const dim3 grid((unsigned)((total8 + 255) / 256)), block(256);
  if (in_bf16)
    hipLaunchKernelGGL((im2col_kernel<true>), grid, block, 0, s, images, (bf16_t*)col, total8, C, img, p, g, Kp, fast);
  else
    hipLaunchKernelGGL((im2col_kernel<false>), grid, block, 0, s, images, (bf16_t*)col, total8, C, img, p, g, Kp, fast);
  return hipGetLastError();
}

// x[b*row_stride][:] = cls + pos[0]
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       bf16_t* __restrict__ x, int batch, int64_t row_stride, int D) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)batch * D) return;
  const int64_t b = idx / D;
  const int d = (int)(idx - b * D);
  const float v = cls[d] + (pos ? pos[d] : 0.0f);
  x[b * row_stride * D + d] = (bf16_t)v;
}

hipError_t launch_cls_rows(const float* cls, const float* pos, void* x, int batch, int64_t row_stride, int D,
                           hipStream_t s) {
  const int64_t total = (int64_t)batch * D;
  hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cls, pos,
                     (bf16_t*)x, batch, row_stride, D);
  return hipGetLastError();
}

// token model: x[b*(S+c) + c + i] = tok[b*S + i] (+ pos[c+i]);  x[b*(S+c)] = cls (+ pos[0])
template <bool IN_BF16>
__global__ __launch_bounds__(256) void assemble_kernel(const void* __restrict__ tok, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, bf16_t* __restrict__ x,
                                                       int batch, int seq, int D, int has_cls) {
  const int N = seq + has_cls;
  const int d4 = D >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)batch * N * d4) return;
  const int64_t row = idx / d4;
  const int d = (int)(idx - row * d4) * 4;
  const int64_t b = row / N;
  const int t = (int)(row - b * N);
  float v[4];
  if (has_cls && t == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = cls[d + e];
  } else {
    const int64_t src = (b * seq + (t - has_cls)) * (int64_t)D + d;
    if (IN_BF16) {
      const bf16x4 a = *reinterpret_cast<const bf16x4*>((const bf16_t*)tok + src);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)a[e];
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)tok + src);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[e];
    }
  }
  if (pos) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += pos[(int64_t)t * D + d + e];
  }
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x4*>(x + row * D + d) = o;
}

hipError_t launch_assemble_tokens(const void* tok, int in_bf16, const float* cls, const float* pos, void* x,
                                  int batch, int seq, int D, int has_cls, hipStream_t s) {
  if (D & 3) return hipErrorInvalidValue;
  const int64_t total = (int64_t)batch * (seq + has_cls) * (D / 4);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (in_bf16)
    hipLaunchKernelGGL((assemble_kernel<true>), grid, block, 0, s, tok, cls, pos, (bf16_t*)x, batch, seq, D, has_cls);
  else
    hipLaunchKernelGGL((assemble_kernel<false>), grid, block, 0, s, tok, cls, pos, (bf16_t*)x, batch, seq, D, has_cls);
  return hipGetLastError();
}

// y[r][:] = x[imap(r)][:]  (bf16 in; bf16 or fp32 out) — the x[:,0,:] / x[:,1:,:] slice for models
// without a final norm (post-LN nn.TransformerEncoder, models_archs.py:147)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ x, void* __restrict__ y,
                                                          int64_t rows, int D, int rpg, int64_t gs, int off) {
  const int d4 = D >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * d4) return;
  const int64_t r = idx / d4;
  const int d = (int)(idx - r * d4) * 4;
  const int64_t ir = map_row(r, rpg, gs, off);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(x + ir * D + d);
  if (OUT_BF16) {
    *reinterpret_cast<bf16x4*>((bf16_t*)y + r * D + d) = a;
  } else {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (float)a[e];
    *reinterpret_cast<f32x4*>((float*)y + r * D + d) = o;
  }
}

hipError_t launch_gather_rows(const void* x, void* y, int out_bf16, int64_t rows, int D, RowMap imap,
                              hipStream_t s) {
  if (D & 3) return hipErrorInvalidValue;
  const int64_t total = rows * (D / 4);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (out_bf16)
    hipLaunchKernelGGL((gather_rows_kernel<true>), grid, block, 0, s, (const bf16_t*)x, y, rows, D, imap.rpg,
                       imap.gstride, imap.off);
  else
    hipLaunchKernelGGL((gather_rows_kernel<false>), grid, block, 0, s, (const bf16_t*)x, y, rows, D, imap.rpg,
                       imap.gstride, imap.off);
  return hipGetLastError();
}

// (sum, sumsq) partials per 64-column group -> (mean, rstd) per row.  The sums come from the producing
// GEMM's epilogue; variance = E[x^2] - mean^2 evaluated in double from the fp32 partials.
__global__ __launch_bounds__(64) void ln_finalize_kernel(const float* __restrict__ part, int groups, int64_t stride,
                                                          float* __restrict__ stats, int64_t rows, float inv_d, float eps) {
  const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= rows) return;
  double s1 = 0.0, s2 = 0.0;
  for (int g0 = 0; g0 < groups; g0 += 16) {
    float2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // 16 independent loads in flight (coalesced across the wave): one round trip for D <= 1024
      const int g = g0 + j < groups ? g0 + j : groups - 1;
      v[j] = *reinterpret_cast<const float2*>(part + ((int64_t)g * stride + r) * 2);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (g0 + j < groups) {
        s1 += (double)v[j].x;
        s2 += (double)v[j].y;
      }
  }
  const double mean = s1 * (double)inv_d;
  double var = s2 * (double)inv_d - mean * mean;
  var = var > 0.0 ? var : 0.0;
  float2 o;
  o.x = (float)mean;
  o.y = (float)(1.0 / sqrt(var + (double)eps));
  *reinterpret_cast<float2*>(stats + r * 2) = o;
}

hipError_t launch_ln_finalize(const float* part, int groups, int64_t stride, float* stats, int64_t rows, int D,
                              float eps, hipStream_t s) {
  if (rows <= 0 || groups <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, part, groups, stride,
                     stats, rows, 1.0f / (float)D, eps);
  return hipGetLastError();
}

// x[b*row_stride][:] = cls + pos[0], one wave per (image, 64-column group), plus that group's partial sums
__global__ __launch_bounds__(64) void cls_rows_stats_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                            bf16_t* __restrict__ x, float* __restrict__ part,
                                                            int64_t part_stride, int groups, int64_t row_stride, int D) {
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int d = g * 64 + threadIdx.x;
  const int64_t row = (int64_t)b * row_stride;
  const bf16_t o = (bf16_t)(cls[d] + (pos ? pos[d] : 0.0f));
  x[row * D + d] = o;
  const float r = (float)o;
  const float s1 = wave_sum(r), s2 = wave_sum(r * r);
  if (threadIdx.x == 0) {
    float* dst = part + ((int64_t)g * part_stride + row) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
}

hipError_t launch_cls_rows_stats(const float* cls, const float* pos, void* x, float* part, int64_t part_stride,
                                 int batch, int64_t row_stride, int D, hipStream_t s) {
  if (D & 63) return hipErrorInvalidValue;
  const int groups = D / 64;
  hipLaunchKernelGGL(cls_rows_stats_kernel, dim3((unsigned)(batch * groups)), dim3(64), 0, s, cls, pos, (bf16_t*)x, part,
                     part_stride, groups, row_stride, D);
  return hipGetLastError();
}

// 3x3 / pad 1 im2col over NHWC tokens (the SAM neck's second conv): one thread moves 8 channels (16 B)
__global__ __launch_bounds__(256) void im2col3_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ col,
                                                      int64_t total, int g, int C) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8 = C >> 3;
  const int64_t row = idx / (9 * c8);
  const int rem = (int)(idx - row * 9 * c8);
  const int j = rem / c8, c = (rem - j * c8) * 8;
  const int ky = j / 3, kx = j - ky * 3;
  const int64_t b = row / (g * g);
  const int pix = (int)(row - b * g * g);
  const int py = pix / g + ky - 1, px = pix % g + kx - 1;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.0f;
  if (py >= 0 && py < g && px >= 0 && px < g) v = *reinterpret_cast<const bf16x8*>(y + ((b * g + py) * g + px) * (int64_t)C + c);
  *reinterpret_cast<bf16x8*>(col + row * (int64_t)(9 * C) + j * C + c) = v;
}

hipError_t launch_im2col3(const void* y, void* col, int batch, int g, int C, hipStream_t s) {
  if (C & 7) return hipErrorInvalidValue;
  const int64_t total = (int64_t)batch * g * g * 9 * (C / 8);
  hipLaunchKernelGGL(im2col3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)col,
                     total, g, C);
  return hipGetLastError();
}

}  // namespace vdr
