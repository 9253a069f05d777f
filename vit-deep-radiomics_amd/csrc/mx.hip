// MX-fp8 (OCP e4m3 payload + e8m0 scale per 32 consecutive K elements) operands for the gfx950 block-scaled
// MFMA  v_mfma_scale_f32_32x32x64_f8f6f4  (BASELINE config 5: DINOv2 ViT-g/14 with fp8 weights).
//
// Layout of an MX tensor X[rows][K]:
//   payload  q[rows][K]              one byte per element, row-major
//   scales   s[K/32][rows_pad]       e8m0 (value 2^(s-127)); rows_pad = rows rounded up to 256; inside every
//                                    64-row group the rows are stored as (r, r+32) pairs:
//                                    perm(row) = (row & ~63) + 2 (row & 31) + ((row >> 5) & 1)
// so the scales a GEMM wave needs for one 64-element K unit (2 blocks x its 64 rows, both MFMA row tiles of a
// lane adjacent) are two contiguous 64-byte runs: one 4-byte global_load_lds per lane stages them, and a lane
// reads the pair of its two row tiles with one ds_read_u16 (the MFMA's opsel picks the byte).
//
// Quantiser (restated on the CPU by the test suite): per block, e = ceil(log2(amax / 448)) clamped to
// [-126, 126], scale byte = e + 127, payload = RNE_e4m3(x * 2^-e)  (|x| 2^-e <= 448: never saturates).
#include "vdr_dev.h"
#include "vdr_kernels.h"

namespace vdr {

VDR_DEV int mx_perm(int row64) { return 2 * (row64 & 31) + ((row64 >> 5) & 1); }

// scale byte of a block whose absolute maximum is amax
VDR_DEV int mx_scale_byte(float amax) {
  const float t = amax * (1.0f / 448.0f);
  const uint32_t b = __float_as_uint(t);
  int e = (int)((b >> 23) & 255) - 127 + ((b & 0x7fffff) ? 1 : 0);  // ceil(log2 t)
  e = e < -126 ? -126 : (e > 126 ? 126 : e);
  return e + 127;
}
VDR_DEV float mx_inv_scale(int byte) { return __uint_as_float((uint32_t)(254 - byte) << 23); }  // 2^-(byte-127)

// 4 floats -> 4 e4m3 bytes (RNE), packed little-endian
VDR_DEV uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// ---------------------------------------------------------------------------------------------------
// plain quantiser: bf16 [rows][K] -> MX.  One lane = 8 consecutive elements, 4 lanes = one block.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mx_quant_kernel(const bf16_t* __restrict__ x, int64_t rows, int K, int64_t ldx,
                                                       uint8_t* __restrict__ q, uint8_t* __restrict__ s,
                                                       int64_t rows_pad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, 8-element group)
  const int k8 = K >> 3;
  const int64_t row = idx / k8;
  const int g = (int)(idx - row * k8);
  const bool ok = row < rows;
  float v[8];
  if (ok) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + row * ldx + g * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  }
  float amax = 0.0f;
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  const int sb = mx_scale_byte(amax);
  const float inv = mx_inv_scale(sb);
  if (!ok) return;
  u32x2 o;
  o[0] = pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
  o[1] = pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
  *reinterpret_cast<u32x2*>(q + row * K + g * 8) = o;
  if ((g & 3) == 0) s[(int64_t)(g >> 2) * rows_pad + (row & ~(int64_t)63) + mx_perm((int)(row & 63))] = (uint8_t)sb;
}

hipError_t launch_mx_quant(const void* x, int64_t rows, int K, int64_t ldx, void* q, void* scales, hipStream_t st) {
  if (rows <= 0 || K <= 0 || (K & 31) || (ldx & 7)) return hipErrorInvalidValue;
  const int64_t total = rows * (K >> 3);
  hipLaunchKernelGGL(mx_quant_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x, rows, K,
                     ldx, (uint8_t*)q, (uint8_t*)scales, mx_rows_pad(rows));
  return hipGetLastError();
}

// MX -> fp32 (tests / inspection): e4m3 decode * 2^(scale - 127)
__global__ __launch_bounds__(256) void mx_dequant_kernel(const uint8_t* __restrict__ q, const uint8_t* __restrict__ s,
                                                         int64_t rows, int K, int64_t rows_pad, float* __restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * K) return;
  const int64_t row = idx / K;
  const int k = (int)(idx - row * K);
  const uint32_t b = q[idx];
  const uint32_t e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? (float)m * 0.001953125f : __uint_as_float(((e + 120) << 23) | (m << 20));
  if (e == 15 && m == 7) v = __uint_as_float(0x7fc00000);
  if (b & 128) v = -v;
  const int sb = s[(int64_t)(k >> 5) * rows_pad + (row & ~(int64_t)63) + mx_perm((int)(row & 63))];
  y[idx] = sb == 255 ? __uint_as_float(0x7fc00000) : v * (sb == 0 ? 5.877471754111438e-39f : __uint_as_float((uint32_t)sb << 23));
}

hipError_t launch_mx_dequant(const void* q, const void* scales, int64_t rows, int K, float* y, hipStream_t st) {
  if (rows <= 0 || K <= 0 || (K & 31)) return hipErrorInvalidValue;
  const int64_t total = rows * K;
  hipLaunchKernelGGL(mx_dequant_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint8_t*)q,
                     (const uint8_t*)scales, rows, K, mx_rows_pad(rows), y);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm (bf16 rows in) -> MX out: the qkv / fc1 GEMM operand of the fp8 path.  Wave per row as
// layernorm_kernel (rowops.hip); a lane owns 8 consecutive columns per 512-column pass (one 16-byte load, one 8-byte
// store), 4 lanes = one 32-column block.  At ViT-g/14 (8224 rows of 1536: one row per wave fills the chip exactly
// once) a launch takes 16 us for 38 MB with 4 or with 8 columns per lane: one latency chain of load, two wave
// reductions, block maxima, store -- not bandwidth.
// ---------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void ln_mx_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, int64_t rows, int D,
                                                    uint8_t* __restrict__ q, uint8_t* __restrict__ s,
                                                    int64_t rows_pad, int win_ws, int win_g) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  int64_t orow = r;
  if (win_ws > 0) {
    // SAM window partition of the output rows (as layernorm_kernel): token (b, y, x) -> window-major order
    const int g = win_g, ws = win_ws, nw = (g + ws - 1) / ws;
    const int64_t b = r / (g * g);
    const int rem = (int)(r - b * g * g);
    const int y = rem / g, x = rem - y * g;
    orow = ((b * nw + y / ws) * nw + x / ws) * (int64_t)(ws * ws) + (y % ws) * ws + (x % ws);
  }
  float v[NP][8];
  float sum = 0.0f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int c = k * 512 + lane * 8;
    if (c < D) {  // D is a multiple of 32: a lane's 8 columns are entirely inside or outside
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + r * D + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[k][e] = (float)t[e];
        sum += v[k][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = 0.0f;
    }
  }
  const float invD = 1.0f / (float)D;
  const float mean = wave_sum(sum) * invD;
  float sq = 0.0f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int c = k * 512 + lane * 8;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[k][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) * invD + eps);
  const int64_t srow = (orow & ~(int64_t)63) + mx_perm((int)(orow & 63));
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int c = k * 512 + lane * 8;
    const bool ok = c < D;
    float o[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (ok) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[k][e] - mean) * rstd * g0[e] + b0[e];
        o[4 + e] = (v[k][4 + e] - mean) * rstd * g1[e] + b1[e];
      }
    }
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const int sb = mx_scale_byte(amax);
    const float inv = mx_inv_scale(sb);
    if (ok) {
      u32x2 w;
      w[0] = pack_fp8x4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      w[1] = pack_fp8x4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
      *reinterpret_cast<u32x2*>(q + orow * D + c) = w;
      if ((lane & 3) == 0) s[(int64_t)(c >> 5) * rows_pad + srow] = (uint8_t)sb;
    }
  }
}

hipError_t launch_ln_mx(const void* x, const float* gamma, const float* beta, float eps, int64_t rows, int D, void* q,
                        void* scales, hipStream_t st, int win_ws, int win_g, int64_t out_rows) {
  if (rows <= 0 || D <= 0 || (D & 31) || D > 2048) return hipErrorInvalidValue;
  const int np = (D + 511) / 512;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int64_t rp = mx_rows_pad(out_rows > 0 ? out_rows : rows);  // rows of the MX tensor the scales belong to
#define VDR_LNMX(NP)                                                                                              \
  case NP:                                                                                                        \
    hipLaunchKernelGGL((ln_mx_kernel<NP>), grid, block, 0, st, (const bf16_t*)x, gamma, beta, eps, rows, D,      \
                       (uint8_t*)q, (uint8_t*)scales, rp, win_ws, win_g);                                         \
    break;
  switch (np) {
    VDR_LNMX(1) VDR_LNMX(2) VDR_LNMX(3) VDR_LNMX(4)
    default:
      return hipErrorInvalidValue;
  }
#undef VDR_LNMX
  return hipGetLastError();
}

}  // namespace vdr
