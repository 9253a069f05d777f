// In-plane cubic-spline resampling of a slice stack on the GPU: the reference's offline augmentation
// rotate_image (reference src/tfds_dense_descriptor.py:327-350) = scipy.ndimage.rotate(vol, angle, axes=(0, 1),
// reshape=False, mode='nearest') with SciPy's defaults order = 3, prefilter = True.  SciPy hands every (H, W) plane
// to affine_transform; this file is that function for all planes of an [H, W, T] volume (T fastest) at once, in
// float64 and in SciPy's operation order (file compiled with -ffp-contract=off), so image AND thresholded mask come
// out bit-identical to SciPy's (the mask is the truncation (unsigned char) t of a value that rings around 1.0 at the
// 1e-5 level inside the nodule: anything but the same arithmetic flips pixels):
//   rot_pad_kernel      edge-pad by 12 samples per side (_prepad_for_spline_filter, mode 'nearest') -> float64
//   rot_filter_kernel   cubic B-spline prefilter of every line along one axis, in place: gain (1 - z)(1 - 1/z), causal
//                       initialisation of the 'reflect' boundary (what SciPy uses for 'nearest'), causal and
//                       anti-causal recursion.  One lane per line; neighbouring lanes are neighbouring t: coalesced.
//   rot_sample_kernel   x = ((M o) + offset) + 12 per axis, 4 x 4 taps from floor(x) - 1 with clamped tap indices,
//                       get_spline_interpolation_weights' cubic weights, t = sum_i sum_j (c_ij * w0_i) * w1_j from 0.0;
//                       stores double / float (optionally clipped to [0, 1], the np.clip of rotate_image) or the
//                       truncated unsigned char of a boolean mask.
// HBM-bound: one padded float64 copy of the volume written once, three strided passes over it per filter axis, 16
// gathered float64 taps per output sample (neighbouring lanes read neighbouring t of the same taps).
#include <cmath>

#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

namespace {

constexpr int ROT_NPAD = 12;
constexpr double ROT_POLE = -0.267949192431122706472553658494127633;  // sqrt(3) - 2, SciPy's literal

template <typename T>
__global__ __launch_bounds__(256) void rot_pad_kernel(const T* __restrict__ src, double* __restrict__ dst, int H, int W,
                                                       int64_t planes) {
  const int Wp = W + 2 * ROT_NPAD;
  const int64_t total = (int64_t)(H + 2 * ROT_NPAD) * Wp * planes;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t t = i % planes;
  const int64_t pix = i / planes;
  int x = (int)(pix % Wp) - ROT_NPAD;
  int y = (int)(pix / Wp) - ROT_NPAD;
  x = x < 0 ? 0 : x > W - 1 ? W - 1 : x;
  y = y < 0 ? 0 : y > H - 1 ? H - 1 : y;
  dst[i] = (double)src[((int64_t)y * W + x) * planes + t];
}

// line l of `lines`: first element at (l / inner) * outer_stride + (l % inner), elements `stride` apart, n of them.
// The recursions are serial per line, the loads are not: every pass fetches FB elements ahead of the arithmetic
// (a load never depends on a store of the same pass), so a lane keeps FB rows in flight instead of one.
constexpr int FB = 8;

__global__ __launch_bounds__(64) void rot_filter_kernel(double* __restrict__ c, int64_t lines, int64_t inner, int64_t outer_stride,
                                                         int64_t stride, int n, double z_n) {
  const int64_t l = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (l >= lines) return;
  double* p = c + (l / inner) * outer_stride + (l % inner);
  const double z = ROT_POLE;
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  const double c0 = p[0] * gain;
  double acc = c0 + z_n * (p[(int64_t)(n - 1) * stride] * gain);
  double z_i = z;
  double f[FB], r[FB];
  int i = 1;
  for (; i + FB <= n - 1; i += FB) {  // terms 1 .. n-2: c[i] from the front, c[n-1-i] from the back
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      f[k] = p[(int64_t)(i + k) * stride];
      r[k] = p[(int64_t)(n - 1 - i - k) * stride];
    }
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      acc = acc + z_i * (f[k] * gain + z_n * (r[k] * gain));
      z_i *= z;
    }
  }
  for (; i < n; ++i) {
    // SciPy accumulates into c[0] in place and its last term (i = n - 1) reads c[0] itself
    const double other = i == n - 1 ? acc : p[(int64_t)(n - 1 - i) * stride] * gain;
    acc = acc + z_i * (p[(int64_t)i * stride] * gain + z_n * other);
    z_i *= z;
  }
  acc = acc * (z / (1.0 - z_n * z_n));
  double prev = acc + c0;
  p[0] = prev;
  i = 1;
  for (; i + FB <= n; i += FB) {
#pragma unroll
    for (int k = 0; k < FB; ++k) f[k] = p[(int64_t)(i + k) * stride];
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      prev = f[k] * gain + z * prev;
      p[(int64_t)(i + k) * stride] = prev;
    }
  }
  for (; i < n; ++i) {
    prev = p[(int64_t)i * stride] * gain + z * prev;
    p[(int64_t)i * stride] = prev;
  }
  prev = prev * (z / (z - 1.0));
  p[(int64_t)(n - 1) * stride] = prev;
  i = n - 2;
  for (; i - FB + 1 >= 0; i -= FB) {
#pragma unroll
    for (int k = 0; k < FB; ++k) f[k] = p[(int64_t)(i - k) * stride];
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      prev = z * (prev - f[k]);
      p[(int64_t)(i - k) * stride] = prev;
    }
  }
  for (; i >= 0; --i) {
    prev = z * (prev - p[(int64_t)i * stride]);
    p[(int64_t)i * stride] = prev;
  }
}

struct RotTaps {
  int idx[4];
  double w[4];
};

__device__ inline RotTaps rot_taps(double x, int n) {
  RotTaps r;
  const double fl = floor(x);
  // coordinates far outside the array: every tap clamps to the edge sample anyway
  const int start = (fl < -8.0 ? -8 : fl > (double)(n + 8) ? n + 8 : (int)fl) - 1;
  const double y = x - fl;
  const double zz = 1.0 - y;
  r.w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
  r.w[2] = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0;
  r.w[0] = zz * zz * zz / 6.0;
  r.w[3] = 1.0 - r.w[0] - r.w[1] - r.w[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = start + k;
    r.idx[k] = i < 0 ? 0 : i > n - 1 ? n - 1 : i;
  }
  return r;
}

struct RotK {
  const double* coef;
  void* out;
  int H, W;
  int64_t planes;
  double m00, m01, m10, m11, off0, off1;
  int clip01;
};

// OUT: 0 double, 1 float, 2 unsigned char (boolean mask)
template <int OUT>
__global__ __launch_bounds__(256) void rot_sample_kernel(RotK p) {
  const int64_t total = (int64_t)p.H * p.W * p.planes;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t t = i % p.planes;
  const int64_t pix = i / p.planes;
  const double o1 = (double)(pix % p.W);
  const double o0 = (double)(pix / p.W);
  const int Hp = p.H + 2 * ROT_NPAD, Wp = p.W + 2 * ROT_NPAD;
  const double x0 = ((o0 * p.m00 + o1 * p.m01) + p.off0) + (double)ROT_NPAD;
  const double x1 = ((o0 * p.m10 + o1 * p.m11) + p.off1) + (double)ROT_NPAD;
  const RotTaps a = rot_taps(x0, Hp);
  const RotTaps b = rot_taps(x1, Wp);
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double cf = p.coef[((int64_t)a.idx[u] * Wp + b.idx[v]) * p.planes + t];
      acc = acc + (cf * a.w[u]) * b.w[v];
    }
  if (OUT == 0) {
    if (p.clip01) acc = acc < 0.0 ? 0.0 : acc > 1.0 ? 1.0 : acc;
    ((double*)p.out)[i] = acc;
  } else if (OUT == 1) {
    float f = (float)acc;
    if (p.clip01) f = f < 0.0f ? 0.0f : f > 1.0f ? 1.0f : f;
    ((float*)p.out)[i] = f;
  } else {
    ((unsigned char*)p.out)[i] = (unsigned char)(int)acc;  // (npy_bool) t: truncation toward zero
  }
}

}  // namespace

size_t affine_cubic_scratch_bytes(int H, int W, int64_t planes) {
  return (size_t)(H + 2 * ROT_NPAD) * (size_t)(W + 2 * ROT_NPAD) * (size_t)planes * sizeof(double);
}

hipError_t launch_affine_cubic(const void* src, int dtype /*0 f64, 1 f32, 2 u8*/, int H, int W, int64_t planes,
                               const double* matrix, const double* offset, void* out, int clip01, double* scratch,
                               hipStream_t s) {
  const int Hp = H + 2 * ROT_NPAD, Wp = W + 2 * ROT_NPAD;
  const int64_t padded = (int64_t)Hp * Wp * planes;
  const unsigned gpad = (unsigned)((padded + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(rot_pad_kernel<double>, dim3(gpad), dim3(256), 0, s, (const double*)src, scratch, H, W, planes);
  else if (dtype == 1) hipLaunchKernelGGL(rot_pad_kernel<float>, dim3(gpad), dim3(256), 0, s, (const float*)src, scratch, H, W, planes);
  else hipLaunchKernelGGL(rot_pad_kernel<unsigned char>, dim3(gpad), dim3(256), 0, s, (const unsigned char*)src, scratch, H, W, planes);
  // axis 0: one line per (x, t), elements Wp * planes apart; then axis 1: one line per (y, t), elements planes apart.
  // pow() on the host: the same libm call SciPy's C code makes
  {
    const int64_t lines = (int64_t)Wp * planes;
    hipLaunchKernelGGL(rot_filter_kernel, dim3((unsigned)((lines + 63) / 64)), dim3(64), 0, s, scratch, lines, lines, (int64_t)0,
                       (int64_t)Wp * planes, Hp, std::pow(ROT_POLE, (double)Hp));
  }
  {
    const int64_t lines = (int64_t)Hp * planes;
    hipLaunchKernelGGL(rot_filter_kernel, dim3((unsigned)((lines + 63) / 64)), dim3(64), 0, s, scratch, lines, planes,
                       (int64_t)Wp * planes, planes, Wp, std::pow(ROT_POLE, (double)Wp));
  }
  RotK k{scratch, out, H, W, planes, matrix[0], matrix[1], matrix[2], matrix[3], offset[0], offset[1], clip01};
  const int64_t total = (int64_t)H * W * planes;
  const unsigned g = (unsigned)((total + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(rot_sample_kernel<0>, dim3(g), dim3(256), 0, s, k);
  else if (dtype == 1) hipLaunchKernelGGL(rot_sample_kernel<1>, dim3(g), dim3(256), 0, s, k);
  else hipLaunchKernelGGL(rot_sample_kernel<2>, dim3(g), dim3(256), 0, s, k);
  return hipGetLastError();
}

}  // namespace vdr
