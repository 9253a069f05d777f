// Host-side launch interface of the gfx950 kernels (internal; the public ABI is include/vdr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// tuning knobs (-DVDR_TUNING builds read VDR_* environment variables): re-read on every call there, so that one
// process can alternate settings between forwards (interleaved A/B, tools/ab_forward.py); constants in the shipped build
#ifdef VDR_TUNING
#define VDR_KNOB const
#else
#define VDR_KNOB static const
#endif

namespace vdr {

// One-time launch state of a kernel instantiation is PER DEVICE: hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in),
// occupancy and CU counts belong to the device that is current when they are set / read.  A handle runs on its own
// device whatever device the caller has current (include/vdr.h), so a second handle on another GPU of the same process
// must find its own flags, not the first device's.
constexpr int VDR_MAX_DEVICES = 64;
static inline int current_device_index() {
  int d = -1;
  return hipGetDevice(&d) == hipSuccess && d >= 0 && d < VDR_MAX_DEVICES ? d : -1;
}
struct PerDeviceFlag {
  bool done[VDR_MAX_DEVICES] = {};
};
static inline int device_cu_count(int dev) {  // 0 on failure
  static int n[VDR_MAX_DEVICES] = {};
  if (dev < 0 || dev >= VDR_MAX_DEVICES) return 0;
  if (!n[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
  }
  return n[dev];
}

// row r of a compact [R, *] view  <->  row (r / rpg) * gstride + off + (r % rpg) of a token buffer
struct RowMap {
  int rpg;
  int64_t gstride;
  int off;
};
static inline RowMap identity_map() { return RowMap{1 << 30, 0, 0}; }

enum Epilogue { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_SWIGLU = 3, EPI_PATCH = 4,
                // internal to gemm_mx.hip: GELU / SwiGLU with the output re-quantised to MX-fp8
                EPI_BIAS_GELU_MX = 5, EPI_SWIGLU_MX = 6,
                // internal to launch_cfg: EPI_BIAS_RESID with the residual stream kept in fp32 (GemmArgs::resid32 / C32)
                EPI_BIAS_RESID32 = 7 };

struct GemmArgs {
  const void* A;      // [M, K] bf16, row stride lda
  const void* W;      // [N, K] bf16 (PyTorch Linear layout), row stride ldw; or the packed form (w_interleaved)
  int w_interleaved = 0;  // W is [N/2][K/32][2][32] (launch_w_interleave): whole-line operand loads, gemm_kernels.h
  const float* bias;  // [N] or null
  const void* resid;  // [*, ldr] bf16 (EPI_BIAS_RESID), indexed by the OUTPUT row
  // vdr_config.resid_fp32 (ring3 / ring4 kernels, EPI_BIAS_RESID): the residual is READ from resid32 [*, ldr] fp32 instead of
  // `resid`, the fp32 sum is written to C32 [*, ldc] fp32 AND, rounded once, to C (the bf16 copy the next GEMM multiplies)
  const float* resid32 = nullptr;
  float* C32 = nullptr;
  // rows of A (and of ln_stats) that are READABLE memory, >= M (0 = M).  Tile variant 31 loads whole 256-row tiles: it takes
  // a ragged M only when the buffers are readable up to M rounded up to 256 (the forward's workspace is); rows past M are
  // never stored (the store descriptor ends after row M - 1)
  int64_t a_rows = 0;
  const float* gamma; // [N] LayerScale or null
  const float* pos;   // [tokens, N] fp32 (EPI_PATCH), indexed by off + r % rpg
  void* C;            // bf16, row stride ldc
  int64_t M;
  int N, K;
  int64_t lda, ldw, ldc, ldr;
  RowMap omap;        // output row map (EPI_PATCH); identity otherwise
  // LayerNorm folded into the GEMM (pre-LN models):
  //   consumer side: y = r_m * (x.W'^T - mu_m * colsum) + bias', with (mu_m, r_m) = ln_stats[m]
  const float* ln_stats = nullptr;  // [M][2] fp32 (mean, rstd) or null
  const float* colsum = nullptr;    // [N] fp32: sum_k W'[n][k]
  //   consumer side, statistics finalised inside the GEMM (ring3 variants 22-24, up to 16 groups): the producers'
  //   partials instead of ln_stats; (mean, rstd) = what ln_finalize_kernel computes from them, bit for bit
  const float* ln_cpart = nullptr;  // [ln_groups][ln_cstride][2] fp32 or null
  int ln_groups = 0;
  int64_t ln_cstride = 0;
  float ln_inv_d = 0.0f, ln_eps = 0.0f;
  // producer side, ring4 tile variants with the residual epilogue: the workgroup that stores the LAST partial sums of a
  // block of tile rows turns them into (mean, rstd) at fin_stats [M][2] -- ln_finalize_kernel's arithmetic, no launch.
  // fin_cnt: one zeroed counter per tile row (left zeroed); the partials span fin_groups = N / 64 groups
  float* fin_stats = nullptr;
  uint32_t* fin_cnt = nullptr;
  float fin_inv_d = 0.0f, fin_eps = 0.0f;
  //   producer side: per output row and 64-column group, (sum, sum of squares) of the bf16 outputs
  float* ln_part = nullptr;         // [N/64][part_stride][2] fp32 or null
  int64_t part_stride = 0;
  // SAM window un-partition on the OUTPUT rows: GEMM row m is a token of a ws x ws window (windows
  // row-major over the zero-padded grid); it lands at token (y, x) of the g x g grid, padding is dropped
  int win_ws = 0, win_g = 0;
  // A rows gathered with two strides: row m starts at element (m / a_rpg) * a_gs + (m % a_rpg) * a_is
  // (a_rpg = 0: plain m * lda).  Used to read the per-head q slices of a packed qkv activation as rows.
  int a_rpg = 0;
  int64_t a_gs = 0, a_is = 0;
  int out_f32 = 0;  // C is fp32 (EPI_BIAS / EPI_PATCH)
  // im2col-free patchify (EPI_PATCH on the ring4 variants): A = bf16 NCHW images [B, C, g*p, g*p], M = B*g*g tokens,
  // K = C*p*p with p in {8, 16, 32}; the operand loader gathers 16-byte runs of pixels straight from the images
  int patch_p = 0, patch_g = 0, patch_C = 0;
  // MX-fp8 GEMM (launch_gemm_mx): A and W are e4m3 payloads, *_scale their e8m0 scale arrays (mx.hip layout)
  const void* a_scale = nullptr;
  const void* w_scale = nullptr;
  void* c_scale = nullptr;  // non-null: C is written as MX-fp8 (payload [M][ldc] bytes + scales), not bf16
};

hipError_t launch_gemm(const GemmArgs& a, int epilogue, int variant, hipStream_t s);
// whether tile variant 31 (gemm_8p.hip: the 256 x 256 x 64 8-phase kernel, one workgroup per CU) takes this launch: EPI_BIAS /
// EPI_BIAS_GELU with a plain bf16 output, plain weight layout, N % 256 == 0, K % 128 == 0, M % 256 == 0 or a_rows covering the
// last 256-row tile, at least 2 tiles per CU with the last round of workgroups at least 85 % full
bool gemm_8p_eligible(const GemmArgs& a, int epilogue);
bool gemm_8p_shape_ok(int64_t M, int N);  // its tile-count rule alone
// [N][K] bf16 (row stride ld elements) -> the pair-interleaved weight layout (N even, K % 32 == 0)
hipError_t launch_w_interleave(const void* src, void* dst, int N, int K, int64_t ld, hipStream_t s);

// y[r] = LN(x[imap(r)]) ; optional cls source: rows with r % cls_period == 0 read cls (fp32 [D])
struct LnArgs {
  const void* x;
  int in_bf16;
  void* y;
  int out_bf16;
  const float* gamma;
  const float* beta;
  int64_t rows;  // output rows
  int D;
  float eps;
  RowMap imap;
  RowMap omap;
  const float* cls;  // or null
  int cls_period;
  // SAM window partition on the OUTPUT rows (input rows are tokens of a g x g grid, row-major)
  int win_ws = 0, win_g = 0;
};
hipError_t launch_layernorm(const LnArgs& a, hipStream_t s);

//   out_scale != NULL: `out` is written as MX-fp8 (payload [batch*seq][heads*64] bytes + e8m0 scales), the
//   operand of the fp8 out-projection
hipError_t launch_attention(const void* qkv, void* out, int batch, int seq, int heads, int variant,
                            hipStream_t s, void* out_scale = nullptr, const int* lens = nullptr, int len_add = 0);
//   lens != NULL: entry b attends over its first lens[b] + len_add rows only (sequences padded to seq)

// SAM / MedSAM decomposed relative position bias (attention_relpos.hip)
//   qkv rows are S*S-token windows (or whole grids) back to back
//   table [Npad][64] bf16: rows [0, 2S-1) = rel_pos_h, rows [Npad/2, Npad/2 + 2S-1) = rel_pos_w, rest zero;
//   Npad = relpos_npad(S).  T = q . table^T is one GEMM over (token, head) rows (launch_gemm with a_rpg).
// ---- MX-fp8 (mx.hip, gemm_mx.hip) ---------------------------------------------------------------------
static inline int64_t mx_rows_pad(int64_t rows) { return (rows + 255) / 256 * 256; }
static inline size_t mx_scale_bytes(int64_t rows, int K) { return (size_t)mx_rows_pad(rows) * (size_t)(K / 32); }
//   bf16 [rows][K] (row stride ldx elements) -> e4m3 payload [rows][K] + e8m0 scales
hipError_t launch_mx_quant(const void* x, int64_t rows, int K, int64_t ldx, void* q, void* scales, hipStream_t s);
hipError_t launch_mx_dequant(const void* q, const void* scales, int64_t rows, int K, float* y, hipStream_t s);
//   LayerNorm over bf16 rows, MX out
//   win_ws > 0: output rows in SAM window-partition order (out_rows = rows of that padded layout)
hipError_t launch_ln_mx(const void* x, const float* gamma, const float* beta, float eps, int64_t rows, int D, void* q,
                        void* scales, hipStream_t s, int win_ws = 0, int win_g = 0, int64_t out_rows = 0);
//   C = epi(A . W^T) with MX operands; variant 0: 128x256 tile / 8 waves, 1: 256x256 / 16 waves, 2: 128x128 / 4 waves
hipError_t launch_gemm_mx(const GemmArgs& a, int epilogue, int variant, hipStream_t s);

// ---- pre/post-processing (prep.hip) --------------------------------------------------------------------
size_t prepare_scratch_bytes(int batch, int h, int w, int ch, int out_side);
hipError_t launch_prepare(const void* src, int src_f32, int batch, int h, int w, int ch, int64_t sb, int64_t sy, int64_t sx,
                          int64_t sc, int flip, int out_side, void* out, int out_bf16, float* scratch, hipStream_t s);
hipError_t launch_window_ct(const void* ct, int in_i16, int64_t n, double width, double level, float* out, hipStream_t s);
hipError_t launch_hu_to_rgb(const void* hu, int dtype, int64_t n, void* rgb, hipStream_t s);
hipError_t launch_crop_hwc(const float* src, float* dst, int batch, int H, int W, int C, int y0, int x0, int ch, int cw,
                           hipStream_t s);
// scipy.ndimage.affine_transform(order=3, mode='nearest', prefilter=True) on every (H, W) plane of an [H, W, planes]
// volume (rotate.hip); dtype 0 f64, 1 f32, 2 u8 (boolean mask, truncating store); scratch = affine_cubic_scratch_bytes
hipError_t launch_voxel_sequence(const float* feat, const int64_t* index, const double* xyz, const double* expo, int64_t n,
                                 int D, void* out, int out_dtype, hipStream_t s);
size_t affine_cubic_scratch_bytes(int H, int W, int64_t planes);
hipError_t launch_affine_cubic(const void* src, int dtype, int H, int W, int64_t planes, const double* matrix,
                               const double* offset, void* out, int clip01, double* scratch, hipStream_t s);

static inline int relpos_npad(int S) { return 2 * ((2 * S - 1 + 31) / 32 * 32); }
hipError_t launch_relpos_pack(const float* rel_h, const float* rel_w, void* table, int S, hipStream_t s);
//   softmax(q k^T / 8 + T[qh - kh + S-1] + T[Npad/2 + qw - kw + S-1]) v per (window, head);
//   T fp32 [batch*S*S*heads][Npad]; S in {4, 7, 10, 14} single pass, 64 chunked
hipError_t launch_attention_relpos(const void* qkv, const float* T, void* out, int batch, int S, int heads,
                                   hipStream_t s);

// 3x3 / pad 1 im2col over NHWC tokens of a g x g grid: col[r][j*C + c] = y[(y+ky-1, x+kx-1)][c], j = ky*3 + kx
hipError_t launch_im2col3(const void* y, void* col, int batch, int g, int C, hipStream_t s);

// images NCHW -> col [batch*n, Kp] bf16 with k = c*p*p + ky*p + kx, zero padded to Kp
hipError_t launch_im2col(const void* images, int in_bf16, void* col, int batch, int C, int img, int p,
                         int Kp, hipStream_t s);

// x[b*row_stride + 0][:] = cls + pos[0]  (bf16 out)
hipError_t launch_cls_rows(const float* cls, const float* pos, void* x, int batch, int64_t row_stride,
                           int D, hipStream_t s);

// token assembly for the token model: x[b*(S+1)+1+i] = tok[b*S+i] (+pos), x[b*(S+1)] = cls (+pos[0]); bf16 out
hipError_t launch_assemble_tokens(const void* tok, int in_bf16, const float* cls, const float* pos,
                                  void* x, int batch, int seq, int D, int has_cls, hipStream_t s);

// LayerNorm statistics kept apart from the normalisation (the normalisation itself is folded into
// the consuming GEMM): part [groups][stride][2] (sum, sumsq per 64-column group) -> stats [rows][2]
// (mean, rstd)
hipError_t launch_ln_finalize(const float* part, int groups, int64_t stride, float* stats, int64_t rows, int D,
                              float eps, hipStream_t s);
// x[b*row_stride][:] = cls + pos[0] as launch_cls_rows, plus that row's partial sums
hipError_t launch_cls_rows_stats(const float* cls, const float* pos, void* x, float* part, int64_t part_stride,
                                 int batch, int64_t row_stride, int D, hipStream_t s);

// y[r] = x[imap(r)], bf16 -> bf16 / fp32
hipError_t launch_gather_rows(const void* x, void* y, int out_bf16, int64_t rows, int D, RowMap imap,
                              hipStream_t s);

}  // namespace vdr
