// Pre/post-processing either side of the encoder, on the GPU: what the reference does
// per slice on the CPU with numpy / skimage before `.cuda()` and after `.cpu()`.
//
//   prepare_kernel      prepare_image (src/tfds_dense_descriptor.py:30-48): gray -> 3 channels, bilinear resize
//                       with skimage.transform.resize semantics (order 1, mode 'reflect' = mirror without edge
//                       repeat, pixel centres at +0.5, float64 arithmetic), HWC -> CHW, optional flip
//                       (flip_image, :305-324) folded into the gather, fp32 or bf16 out.  Reads strided input,
//                       so the slices of an (H, W, S[, C]) volume are a batch without any transpose.
//   blur_kernel         the anti-aliasing Gaussian skimage applies before DOWN-scaling (sigma = (scale-1)/2,
//                       truncate 4, 'mirror' boundary), one axis per launch, fp32 between the passes
//   window_ct_kernel    apply_window_ct (:287-302)
//   hu_to_rgb_kernel    hu_to_rgb_vectorized (src/visualization_utils.py:128-186), uint8 out, numpy's
//                       float -> int truncation and its float32 / float64 promotion rules reproduced
//   crop_hwc_kernel     crop_image / extract_roi (visualization_utils.py:93-125) of a [B, H, W, C] feature map
//
// All of it is HBM-bound elementwise / gather work: one thread per output pixel, coalesced along x.
#include "vdr_dev.h"
#include "vdr_kernels.h"

// numpy evaluates these expressions without fused multiply-adds; the uint8 truncation of the colour map and the
// 1-ulp agreement of the resize depend on doing the same: the Makefile builds this file with -ffp-contract=off
// (-ffp-contract=fast, used for the rest of the library, ignores `#pragma clang fp contract(off)`)

namespace vdr {

VDR_DEV int mirror_idx(int i, int n) {  // numpy.pad 'reflect' / scipy 'mirror'
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = i < 0 ? -i : i;
  i %= p;
  return i >= n ? p - i : i;
}

struct PrepK {
  const void* src;
  int src_f32;  // 1: fp32, 0: fp64
  int64_t sb, sy, sx, sc;  // element strides of (batch, y, x, channel)
  int batch, h, w, ch;     // ch = 1 (replicated to 3) or 3
  int flip;                // 0 none, 1 horizontal (x reversed), 2 vertical (y reversed)
  int out_h, out_w;
  void* out;  // [batch, 3, out_h, out_w]
  int out_bf16;
};

VDR_DEV double prep_load(const PrepK& p, int b, int y, int x, int c) {
  const int64_t o = (int64_t)b * p.sb + (int64_t)y * p.sy + (int64_t)x * p.sx + (int64_t)c * p.sc;
  return p.src_f32 ? (double)reinterpret_cast<const float*>(p.src)[o] : reinterpret_cast<const double*>(p.src)[o];
}

__global__ __launch_bounds__(256) void prepare_kernel(PrepK p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t npix = (int64_t)p.out_h * p.out_w;
  if (idx >= npix * p.batch) return;
  const int b = (int)(idx / npix);
  const int rem = (int)(idx - (int64_t)b * npix);
  const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
  // the flipped image is resized: output pixel (oy, ox) of resize(flip(img)) reads flip(img) at (r, c)
  const double r = ((double)oy + 0.5) * ((double)p.h / (double)p.out_h) - 0.5;
  const double c = ((double)ox + 0.5) * ((double)p.w / (double)p.out_w) - 0.5;
  const double r0 = floor(r), c0 = floor(c);
  const double dr = r - r0, dc = c - c0;
  int y0 = mirror_idx((int)r0, p.h), y1 = mirror_idx((int)ceil(r), p.h);
  int x0 = mirror_idx((int)c0, p.w), x1 = mirror_idx((int)ceil(c), p.w);
  if (p.flip == 1) {
    x0 = p.w - 1 - x0;
    x1 = p.w - 1 - x1;
  } else if (p.flip == 2) {
    y0 = p.h - 1 - y0;
    y1 = p.h - 1 - y1;
  }
  float res[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    if (ch < p.ch) {
      const double v00 = prep_load(p, b, y0, x0, ch), v01 = prep_load(p, b, y0, x1, ch);
      const double v10 = prep_load(p, b, y1, x0, ch), v11 = prep_load(p, b, y1, x1, ch);
      const double top = (1.0 - dc) * v00 + dc * v01;
      const double bot = (1.0 - dc) * v10 + dc * v11;
      res[ch] = (float)((1.0 - dr) * top + dr * bot);
    } else {
      res[ch] = res[0];  // gray2rgb
    }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const int64_t o = ((int64_t)b * 3 + ch) * npix + rem;
    if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)res[ch];
    else reinterpret_cast<float*>(p.out)[o] = res[ch];
  }
}

// Gaussian along one axis of a (possibly strided) [batch, h, w, ch] image -> contiguous fp32 [batch, h, w, ch]
__global__ __launch_bounds__(256) void blur_kernel(PrepK p, float* __restrict__ dst, int axis, double sigma) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)p.batch * p.h * p.w * p.ch;
  if (idx >= total) return;
  int64_t t = idx;
  const int c = (int)(t % p.ch);
  t /= p.ch;
  const int x = (int)(t % p.w);
  t /= p.w;
  const int y = (int)(t % p.h);
  const int b = (int)(t / p.h);
  if (sigma <= 0.0) {
    dst[idx] = (float)prep_load(p, b, y, x, c);
    return;
  }
  const int rad = (int)(4.0 * sigma + 0.5);
  const double inv2 = -0.5 / (sigma * sigma);
  double acc = 0.0, wsum = 0.0;
  for (int k = -rad; k <= rad; ++k) {
    const double wgt = exp(inv2 * (double)k * (double)k);
    const double v = axis == 0 ? prep_load(p, b, mirror_idx(y + k, p.h), x, c) : prep_load(p, b, y, mirror_idx(x + k, p.w), c);
    acc += wgt * v;
    wsum += wgt;
  }
  dst[idx] = (float)(acc / wsum);
}

size_t prepare_scratch_bytes(int batch, int h, int w, int ch, int out_side) {
  if (h <= out_side && w <= out_side) return 0;
  return (size_t)2 * batch * h * w * ch * sizeof(float);
}

hipError_t launch_prepare(const void* src, int src_f32, int batch, int h, int w, int ch, int64_t sb, int64_t sy, int64_t sx,
                          int64_t sc, int flip, int out_side, void* out, int out_bf16, float* scratch, hipStream_t s) {
  if (batch <= 0 || h <= 0 || w <= 0 || (ch != 1 && ch != 3) || out_side <= 0 || flip < 0 || flip > 2)
    return hipErrorInvalidValue;
  PrepK p{};
  p.src = src;
  p.src_f32 = src_f32;
  p.sb = sb;
  p.sy = sy;
  p.sx = sx;
  p.sc = sc;
  p.batch = batch;
  p.h = h;
  p.w = w;
  p.ch = ch;
  p.flip = flip;
  p.out_h = out_side;
  p.out_w = out_side;
  p.out = out;
  p.out_bf16 = out_bf16;
  if (h > out_side || w > out_side) {
    // anti-aliasing prefilter: rows first, then columns, each pass stored in fp32 (skimage filters the float32
    // image with scipy.ndimage, which keeps the image dtype between the two axes)
    if (!scratch) return hipErrorInvalidValue;
    const int64_t total = (int64_t)batch * h * w * ch;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    float* t0 = scratch;
    float* t1 = scratch + total;
    const double sgy = h > out_side ? ((double)h / out_side - 1.0) / 2.0 : 0.0;
    const double sgx = w > out_side ? ((double)w / out_side - 1.0) / 2.0 : 0.0;
    PrepK q = p;
    q.flip = 0;
    hipLaunchKernelGGL(blur_kernel, grid, block, 0, s, q, t0, 0, sgy);
    q.src = t0;
    q.src_f32 = 1;
    q.sc = 1;
    q.sx = ch;
    q.sy = (int64_t)w * ch;
    q.sb = (int64_t)h * w * ch;
    hipLaunchKernelGGL(blur_kernel, grid, block, 0, s, q, t1, 1, sgx);
    p.src = t1;
    p.src_f32 = 1;
    p.sc = q.sc;
    p.sx = q.sx;
    p.sy = q.sy;
    p.sb = q.sb;
  }
  const int64_t n = (int64_t)batch * out_side * out_side;
  hipLaunchKernelGGL(prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// apply_window_ct: clip((ct - lo) / (hi - lo), 0, 1).  fp32 in: numpy evaluates in float32; int16 in: float64.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void window_ct_kernel(const T* __restrict__ ct, float* __restrict__ out, int64_t n,
                                                        double lo, double range) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v;
  if (sizeof(T) == 4) {
    v = ((float)ct[i] - (float)lo) / (float)range;
  } else {
    v = (float)(((double)ct[i] - lo) / range);
  }
  out[i] = fminf(fmaxf(v, 0.0f), 1.0f);
}

hipError_t launch_window_ct(const void* ct, int in_i16, int64_t n, double width, double level, float* out, hipStream_t s) {
  if (n <= 0 || width == 0.0) return hipErrorInvalidValue;
  const double lo = level - width / 2, hi = level + width / 2;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (in_i16) hipLaunchKernelGGL(window_ct_kernel<int16_t>, grid, block, 0, s, (const int16_t*)ct, out, n, lo, hi - lo);
  else hipLaunchKernelGGL(window_ct_kernel<float>, grid, block, 0, s, (const float*)ct, out, n, lo, hi - lo);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// hu_to_rgb_vectorized: piecewise colour map (air / lung / fat / soft tissue / bone), uint8 [n, 3].
// F32 = true reproduces numpy's float32 intermediates for a float32 HU array: ratios = (hu - min) / (max - min)
// and (1 - ratios) are float32, the colour mix is float64, the assignment into the int array truncates.
// ---------------------------------------------------------------------------------------------------
template <bool F32, typename T>
__global__ __launch_bounds__(256) void hu_to_rgb_kernel(const T* __restrict__ hu, uint8_t* __restrict__ rgb, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = (double)hu[i];
  double a[3], b[3], mn = 0.0, mx = 1.0;
  bool mix = false;
  auto set = [](double (&d)[3], double r, double g, double bl) {
    d[0] = r;
    d[1] = g;
    d[2] = bl;
  };
  set(a, 0, 0, 0);
  set(b, 0, 0, 0);
  if (v <= -1000.0) {
    set(a, 0, 0, 0);
  } else if (v < -600.0) {
    set(a, 0, 0, 0), set(b, 194, 105, 82), mn = -1000, mx = -600, mix = true;
  } else if (v <= -400.0) {
    set(a, 194, 105, 82);
  } else if (v < -100.0) {
    set(a, 194, 105, 82), set(b, 194, 166, 115), mn = -400, mx = -100, mix = true;
  } else if (v <= -60.0) {
    set(a, 194, 166, 115);
  } else if (v < 40.0) {
    set(a, 194, 166, 115), set(b, 102, 0, 0), mn = -60, mx = 40, mix = true;
  } else if (v <= 80.0) {
    set(a, 102, 0, 0), set(b, 153, 0, 0), mn = 80, mx = 400, mix = true;  // the reference's own bounds for this band
  } else if (v < 400.0) {
    set(a, 153, 0, 0), set(b, 255, 255, 255), mn = 80, mx = 400, mix = true;
  } else if (v >= 400.0) {
    set(a, 255, 255, 255);
  }  // NaN: every comparison of the reference's nine masks is False, the pixel keeps the zeros of np.zeros
  double r1 = 0.0, r0 = 1.0;
  if (mix) {
    if (F32) {
      const float rf = ((float)hu[i] - (float)mn) / (float)(mx - mn);
      r1 = (double)rf;
      r0 = (double)(1.0f - rf);
    } else {
      r1 = (v - mn) / (mx - mn);
      r0 = 1.0 - r1;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double o = mix ? a[c] * r0 + b[c] * r1 : a[c];
    rgb[i * 3 + c] = (uint8_t)(int64_t)o;  // float -> int64 truncation, then the uint8 wrap of astype
  }
}

hipError_t launch_hu_to_rgb(const void* hu, int dtype /*0 f32, 1 i16, 2 f64*/, int64_t n, void* rgb, hipStream_t s) {
  if (n <= 0) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == 0) hipLaunchKernelGGL((hu_to_rgb_kernel<true, float>), grid, block, 0, s, (const float*)hu, (uint8_t*)rgb, n);
  else if (dtype == 1) hipLaunchKernelGGL((hu_to_rgb_kernel<false, int16_t>), grid, block, 0, s, (const int16_t*)hu, (uint8_t*)rgb, n);
  else if (dtype == 2) hipLaunchKernelGGL((hu_to_rgb_kernel<false, double>), grid, block, 0, s, (const double*)hu, (uint8_t*)rgb, n);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// crop of a channel-last map: dst[b, y, x, :] = src[b, y0 + y, x0 + x, :], C floats per pixel
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_hwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int batch,
                                                       int H, int W, int C, int y0, int x0, int ch, int cw) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)batch * ch * cw * C;
  if (idx >= total) return;
  int64_t t = idx;
  const int c = (int)(t % C);
  t /= C;
  const int x = (int)(t % cw);
  t /= cw;
  const int y = (int)(t % ch);
  const int b = (int)(t / ch);
  dst[idx] = src[(((int64_t)b * H + (y0 + y)) * W + (x0 + x)) * C + c];
}

hipError_t launch_crop_hwc(const float* src, float* dst, int batch, int H, int W, int C, int y0, int x0, int ch, int cw,
                           hipStream_t s) {
  if (batch <= 0 || C <= 0 || y0 < 0 || x0 < 0 || ch <= 0 || cw <= 0 || y0 + ch > H || x0 + cw > W)
    return hipErrorInvalidValue;
  const int64_t total = (int64_t)batch * ch * cw * C;
  hipLaunchKernelGGL(crop_hwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, batch, H, W, C, y0, x0,
                     ch, cw);
  return hipGetLastError();
}

// Stage-C input builder (reference src/train_models.py:143-182, 'transformer' branch of _get_features, with
// positional_encoding_3d :30-44): out[r, :] = feat[index[r], :] + PE(x_r, y_r, z_r) / 4 for the kept (masked) voxels.
// PE: for i < D / 6, e_i = scale^(6 i / D): columns (2i, 2i + 1) + {0, D/3, 2D/3} = (sin, cos)(coordinate / e_i); every
// other column 0.  float64 like numpy (the sum is rounded once to the output type); the exponents come from the host
// (numpy evaluates scale ** (6 * i / D) with libm's pow).
template <typename TOUT>
__global__ __launch_bounds__(256) void voxel_sequence_kernel(const float* __restrict__ feat, const int64_t* __restrict__ index,
                                                              const double* __restrict__ xyz, const double* __restrict__ expo,
                                                              int64_t n, int D, TOUT* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * D) return;
  const int64_t r = i / D;
  const int c = (int)(i % D);
  const int yoff = D / 3, zoff = (2 * D) / 3, nf = D / 6;  // the reference writes `2 * D // 3` = (2 D) // 3
  const int axis = c >= zoff ? 2 : c >= yoff ? 1 : 0;
  const int k = c - (axis == 2 ? zoff : axis == 1 ? yoff : 0);
  double pe = 0.0;
  if (k < 2 * nf) {  // 2 (D / 6) <= D / 3: the three blocks never overlap, the columns between them stay 0
    const double v = xyz[(int64_t)axis * n + r] / expo[k >> 1];
    pe = (k & 1) ? cos(v) : sin(v);
  }
  const double sum = (double)feat[index[r] * D + c] + pe / 4;
  out[i] = (TOUT)sum;
}

hipError_t launch_voxel_sequence(const float* feat, const int64_t* index, const double* xyz, const double* expo, int64_t n,
                                 int D, void* out, int out_dtype /*0 f32, 1 bf16, 2 f64*/, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned g = (unsigned)((n * D + 255) / 256);
  if (out_dtype == 0) hipLaunchKernelGGL(voxel_sequence_kernel<float>, dim3(g), dim3(256), 0, s, feat, index, xyz, expo, n, D, (float*)out);
  else if (out_dtype == 1) hipLaunchKernelGGL(voxel_sequence_kernel<bf16_t>, dim3(g), dim3(256), 0, s, feat, index, xyz, expo, n, D, (bf16_t*)out);
  else hipLaunchKernelGGL(voxel_sequence_kernel<double>, dim3(g), dim3(256), 0, s, feat, index, xyz, expo, n, D, (double*)out);
  return hipGetLastError();
}

}  // namespace vdr
