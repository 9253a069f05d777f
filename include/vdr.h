/*
 * vdr.h — C ABI of libvdr.so, the MI355X-native (gfx950) ViT dense-descriptor /
 * CLS-feature forward path for the larosi/vit-deep-radiomics pipeline.
 *
 * The reference has no FFI of its own: its boundary for this path is the
 * torch.nn.Module call protocol at three sites:
 *
 *   R1  src/tfds_dense_descriptor.py:51-67    model = load_model(name, path); model.model_name
 *   R2  src/tfds_dense_descriptor.py:122-129  model.image_encoder(x) / model.patch_embed(x)
 *   R3  src/models_archs.py:141-147           model(x[B,S,D]) -> (logits[B,C], cls[B,D])
 *
 * Every entry point below says which of those call sites (or which torch op
 * invoked underneath them) it replaces.  The Python shim in
 * vit-deep-radiomics_amd/vdr/ binds these symbols with ctypes and re-creates
 * R1-R3 on top of them (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary;
 *   - every function returns 0 on success or a negative vdr_status; nothing
 *     throws; vdr_last_error() gives the text of the last failure on a handle
 *     (or of the last failure of a handle-less call when passed NULL);
 *   - device pointers are gfx950 HBM addresses of the device the handle was
 *     created on; the caller owns inputs, outputs and the workspace and keeps
 *     them alive until the stream has drained; the library owns only its packed
 *     weights;
 *   - all work is enqueued on the hipStream_t passed as `void* stream`
 *     (NULL = the null stream); the hot-path calls (vdr_forward*, vdr_op_*) never
 *     synchronise the device, allocate or copy from the host: they can be captured
 *     into a HIP graph from the first call on.  The load-time calls
 *     (vdr_set_weight, vdr_finalize) are synchronous;
 *   - a handle's calls run on the handle's device whatever device is current, and
 *     leave the caller's current device unchanged;
 *   - a handle is bound to one device and is not thread-safe (one per rank); several handles on DIFFERENT devices of one
 *     process are supported (the kernels' one-time launch state -- > 64 KB dynamic-LDS opt-in, occupancy, CU count -- is
 *     kept per device), each used from one thread at a time;
 *   - there is NO CPU path in this library: with no HIP device every compute
 *     call fails with VDR_ERR_NO_DEVICE.
 */
#ifndef VDR_H_
#define VDR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDR_ABI_VERSION 8

typedef enum {
  VDR_OK = 0,
  VDR_ERR_INVALID = -1,      /* bad argument / shape / config               */
  VDR_ERR_NO_DEVICE = -2,    /* no gfx950 HIP device visible                */
  VDR_ERR_HIP = -3,          /* a HIP runtime call failed                   */
  VDR_ERR_UNKNOWN_NAME = -4, /* vdr_set_weight: name not part of the config */
  VDR_ERR_WORKSPACE = -5,    /* workspace too small                         */
  VDR_ERR_INCOMPLETE = -6,   /* a weight is missing / vdr_finalize has not run */
  VDR_ERR_UNSUPPORTED = -7   /* config outside what the kernels cover       */
} vdr_status;

typedef enum { VDR_F32 = 0, VDR_BF16 = 1, VDR_F64 = 2, VDR_I16 = 3, VDR_U8 = 4 } vdr_dtype; /* F64 / I16 / U8: pre-processing only */

typedef enum { VDR_ACT_GELU = 0, /* exact erf GELU: models_archs.py:133, timm/DINOv2 Mlp */
               VDR_ACT_SWIGLU = 1 /* DINOv2 ViT-g SwiGLUFFN (w12 / w3)                   */
} vdr_act;

/* What vdr_forward writes. */
typedef enum {
  VDR_OUT_CLS = 0,         /* [B, D]      final-LN(x)[:,0,:]     (models_archs.py:147 contract)     */
  VDR_OUT_DENSE = 1,       /* [B, n, D]   final-LN(x)[:,1:,:]    (tfds_dense_descriptor.py:130-133) */
  VDR_OUT_PATCH_EMBED = 2, /* [B, n, D]   conv patchify only     (tfds_dense_descriptor.py:128)     */
  VDR_OUT_TOKENS = 3,      /* [B, N, D]   every token after the last block + final LN (if any)      */
  VDR_OUT_ENCODER = 4      /* [B, g, g, C] SAM neck output, channel-LAST (tfds_dense_descriptor.py:123-126   */
                           /*             transposes the reference's [B, C, g, g] to (h, w, C) anyway)     */
} vdr_out_mode;

/* Geometry of one frozen ViT.  Mirrors the constructor arguments the reference
 * passes to its third-party ViTs (tfds_dense_descriptor.py:87,104) and to
 * nn.TransformerEncoderLayer (models_archs.py:130-135). */
typedef struct {
  int32_t img;        /* square input side in pixels (224, 336, 896 ...); 0 for a token model */
  int32_t patch;      /* patch side p (14, 16); 0 for a token model                           */
  int32_t in_chans;   /* 3                                                                    */
  int32_t dim;        /* D                                                                    */
  int32_t heads;      /* H ; D / H must be 64                                                 */
  int32_t layers;     /* L                                                                    */
  int32_t mlp_hidden; /* F (GELU: fc1 out; SwiGLU: hidden of w3's input)                      */
  int32_t act;        /* vdr_act                                                              */
  int32_t pre_ln;     /* 1: x += f(LN(x)) + final norm (timm/DINOv2/SAM);                     */
                      /* 0: x = LN(x + f(x)), no final norm (nn.TransformerEncoderLayer)      */
  int32_t layerscale; /* 1: DINOv2 ls1/ls2 gamma                                              */
  int32_t has_cls;    /* 1: a learned cls_token row is prepended                              */
  int32_t has_pos;    /* 1: learned pos_embed [1,N,D] is added                                */
  int32_t input_ln;   /* 1: LayerNorm applied to the assembled tokens before block 0          */
                      /*    (models_archs.py:145)                                             */
  float ln_eps;       /* 1e-6 timm/DINOv2/SAM, 1e-5 torch default (models_archs.py:136)       */
  int32_t micro_batch;/* images per internal pass (0 = library default: the whole batch, split   */
                      /* evenly over `streams`); results do not depend on it                    */
  int32_t streams;    /* internal HIP streams the micro-batches are spread over (0/1 = the caller's   */
                      /* stream only); >1 lets kernels of independent micro-batches overlap, e.g. one */
                      /* GEMM's store-bound epilogue under another's MFMA main loop                   */
  /* SAM / MedSAM image encoder (segment_anything ImageEncoderViT, tfds_dense_descriptor.py:104,123):   */
  int32_t window;     /* > 0: windowed attention of this side (14) with decomposed relative position   */
                      /* bias in every block; needs has_cls = 0, has_pos = 1, pre_ln = 1               */
  int32_t global_mask;/* bit i set: block i attends over the whole grid (SAM ViT-B: 2,5,8,11 = 0x924)  */
  int32_t neck_chans; /* output channels of the conv neck (256); 1x1 conv, LN2d, 3x3 conv, LN2d        */
  /* BASELINE config 5 ("fp8 weights (CDNA4 fp8 MFMA)"): */
  int32_t fp8;        /* 1: the qkv / fc1 (w12) / fc2 (w3) weights are kept as MX-fp8 (OCP e4m3 + e8m0 scale per 32 K    */
                      /* elements) and run on v_mfma_scale_f32_32x32x64_f8f6f4 with MX-fp8 activations; the            */
                      /* out-projection, attention and the residual stream stay bf16; pre_ln only; 0 or 1             */
  int32_t no_ln_fold; /* 0 (default): pre-LN image models fold norm1 / norm2 into the qkv / fc1 GEMMs (the producers of    */
                      /* the residual stream leave row statistics, gamma goes into the weights: no LayerNorm pass);     */
                      /* 1: keep the explicit LayerNorm kernel (numerics A/B, tests)                                    */
  int32_t full_last_block; /* 0 (default): with out_mode VDR_OUT_CLS a pre-LN model runs the out-projection, norm2 and */
                      /* MLP of its LAST block on the CLS rows only -- after the last attention every operation is      */
                      /* row-wise and x[:, 0] is all `model(x) -> (logits, cls)` (models_archs.py:24-29) returns; the   */
                      /* features are bitwise those of the full block.  1: every row (A/B, tests, bench.py              */
                      /* --full-last-block).  Other out_modes and post-LN models always run every row.                  */
  int32_t fp8_cls_bf16; /* fp8 = 1, models with a CLS token: 1 = the MLP (norm2, fc1 / w12, activation, fc2 / w3, residual) of the     */
                      /* CLS ROWS -- one row in ntok per image, the rows the [B, D] CLS feature of models_archs.py:147 is made  */
                      /* of -- runs on the bf16 weights (kept beside the MX-fp8 copies), on a side stream under the fp8 GEMMs of  */
                      /* the other rows; every other row and the qkv projection stay MX-fp8.  Why: a CLS row is carried by     */
                      /* its OWN MLP chain; when the other tokens hold massive-activation channels (DINOv2-g checkpoints) the   */
                      /* attention adds little to it and nothing averages the MX-fp8 rounding of that chain out: row cosine to  */
                      /* fp32 0.989 at 4 blocks / 0.919 at 40 in the stress case of tests/test_model_gpu.py, 0.9986 / 0.9886    */
                      /* with this switch (tools/fp8_outlier_analysis.py, profiles/r04_fp8_outlier_analysis.txt).  0: every row */
                      /* MX-fp8 (stated gate of the fp8 path: row cosine >= 0.99 against fp32 on ordinary weights; under        */
                      /* injected massive-activation channels the CLS rows are gated at 0.985 -- a stated deviation).           */
  int32_t resid_fp32; /* 1 (bf16 path of pre-LN image models; ignored with fp8 = 1, SAM windows and token models): the residual    */
                      /* stream has an fp32 master copy -- the out-projection / fc2 epilogues read it, add in fp32, write it back */
                      /* and write the bf16 copy the next GEMM multiplies; LayerNorm and the final norm read the fp32 copy.  The  */
                      /* reference computes in fp32 throughout (tfds_dense_descriptor.py:123); with the stream stored as bf16      */
                      /* (0, default) its rounding accumulates over the blocks: rel-L2 of the features to fp32 arithmetic 9e-3 /     */
                      /* 1.2e-2 / 1.7e-2 at 12 / 24 / 40 blocks, against 5.8e-3 / 6.6e-3 / 1.0e-2 with this switch                 */
                      /* (tools/resid_precision.py; SURVEY 8d states 1e-2).  Costs 8 more bytes per element of HBM traffic in     */
                      /* the out-projection and fc2 launches.                                                                     */
  int32_t ln_fin_fused; /* LayerNorm fold: where the (sum, sumsq) partials a residual GEMM leaves become (mean, rstd) for a   */
                      /* consumer that reads finalised statistics (large launches).  0 (default): a small ln_finalize launch  */
                      /* between the out-projection / fc2 and the qkv / fc1; 1: inside the residual GEMM -- the workgroup     */
                      /* that adds the last partial to a block of rows finalises the block (ring4 tile variants), no launch.  */
                      /* Same arithmetic: the features are bitwise equal.  Measured equal in time too (ViT-B batch 256: 8.73  */
                      /* vs 8.71-8.74 ms; ViT-L/14@336 batch 64: 25.96 vs 25.99 ms): the counter's round trip at the end of   */
                      /* every tile costs the residual GEMMs what the 23 launches cost, so the simpler form is the default.   */
                      /* (The counters live in the handle, like its internal streams: one forward at a time per handle.)      */
} vdr_config;

typedef struct vdr_model* vdr_handle;

/* ---- lifecycle ------------------------------------------------------------------------ */

/* ABI version of the loaded library (== VDR_ABI_VERSION it was built with). */
int vdr_abi_version(void);

/* 1 when the library was built with -DVDR_TUNING (tools/: diagnostic `variant` encodings >= 100 and VDR_* environment
 * knobs exist), 0 for the shipped build (no environment dependence, unknown variants are an error). */
int vdr_tuning_build(void);

/* Number of visible gfx950 devices (0 when there is none; never fails). */
int vdr_device_count(void);

/* Replaces: model construction in load_dinov2 / load_medsam
 * (tfds_dense_descriptor.py:70-107) and TransformerNoduleClassifier.__init__
 * (models_archs.py:128-139).  Binds the handle to HIP device `device`. */
int vdr_create(const vdr_config* cfg, int device, vdr_handle* out);
void vdr_destroy(vdr_handle h);
const char* vdr_last_error(vdr_handle h);

/* Replaces: load_state_dict (models_archs.py:32-35) / sam_model_registry(path)
 * (tfds_dense_descriptor.py:104).  `name` uses the timm/DINOv2 state_dict keys
 * listed in SURVEY.md §8a ("patch_embed.proj.weight", "cls_token", "pos_embed",
 * "blocks.{i}.norm1.weight", "blocks.{i}.attn.qkv.weight", "blocks.{i}.attn.proj.bias",
 * "blocks.{i}.ls1.gamma", "blocks.{i}.mlp.fc1.weight", "blocks.{i}.mlp.w12.weight",
 * "norm.weight", "input_norm.weight" ...).  `host` points to `numel` contiguous fp32
 * values in HOST memory in the PyTorch layout of that key; the library converts,
 * repacks and uploads (synchronously; this is load time, not the hot path). */
int vdr_set_weight(vdr_handle h, const char* name, const float* host, const int64_t* shape, int ndim);

/* Replaces: the end of load_state_dict / model.eval() (models_archs.py:32-35, tfds_dense_descriptor.py:89,105).
 * Call once after the last vdr_set_weight (and again after any later vdr_set_weight): checks that every weight is
 * set, folds LayerNorm into the consuming linears, builds the packed GEMM layouts / MX-fp8 copies / rel-pos tables.
 * Synchronous (hipMalloc, hipMemcpy, hipDeviceSynchronize): this is load time.  vdr_forward* return
 * VDR_ERR_INCOMPLETE until it has run. */
int vdr_finalize(vdr_handle h);

/* Number of weight tensors the config expects, and the i-th expected name. */
int vdr_num_weights(vdr_handle h);
const char* vdr_weight_name(vdr_handle h, int i);

/* ---- the hot path ----------------------------------------------------------------------- */

/* Bytes of device workspace vdr_forward / vdr_forward_tokens need for `batch`
 * images (token models: `batch` sequences of `seq` tokens, seq ignored otherwise). */
int vdr_workspace_bytes(vdr_handle h, int batch, int seq, size_t* out);

/* Replaces: model.image_encoder(x) / model.patch_embed(x)
 * (tfds_dense_descriptor.py:123,128) followed by the CLS / patch-token slice
 * (models_archs.py:147; tfds_dense_descriptor.py:130-133), batched.
 *   images : device, NCHW [batch, in_chans, img, img], dtype in_dtype, values as the
 *            reference feeds them (raw [0,1]; no mean/std normalisation is applied)
 *   out    : device, shape by out_mode, dtype out_dtype, C-contiguous, row b = image b */
int vdr_forward(vdr_handle h, const void* images, int in_dtype, int batch, void* out, int out_mode,
                int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: TransformerNoduleClassifier.forward up to x[:,0,:]
 * (models_archs.py:141-147): tokens [batch, seq, D] fp32/bf16 on device ->
 * [cls ; tokens] -> (input LN) -> L blocks -> out by out_mode
 * (VDR_OUT_CLS -> [batch, D]; VDR_OUT_TOKENS -> [batch, seq+has_cls, D]). */
int vdr_forward_tokens(vdr_handle h, const void* tokens, int in_dtype, int batch, int seq, void* out,
                       int out_mode, int out_dtype, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Variable-length token sequences: the reference feeds one patient's masked-voxel sequence at a
 * time (batch_size 1, train_models.py:143-182 / conf parameters_models.yaml); here `batch` sequences padded to
 * max_seq go through one call.  seq_lens: device int32 [batch], 1 <= seq_lens[b] <= max_seq; rows past a
 * sequence's length may hold anything finite — attention masks them as keys, and the rows of a sequence never
 * mix with another's.  Outputs as vdr_forward_tokens (VDR_OUT_CLS is what models_archs.py:147 returns); rows of
 * VDR_OUT_TOKENS / DENSE past a sequence's length are undefined. */
int vdr_forward_tokens_varlen(vdr_handle h, const void* tokens, int in_dtype, int batch, int max_seq,
                              const int32_t* seq_lens, void* out, int out_mode, int out_dtype, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- single operators (the torch ops the reference invokes underneath R2/R3) ------------ */
/* Exposed so that each HIP kernel is parity-tested against its torch op through
 * this ABI (tests/test_ops_gpu.py).  All pointers are device pointers. */

/* F.layer_norm(x, (D,), gamma, beta, eps)  — nn.LayerNorm at models_archs.py:136,145 and
 * norm1/norm2/norm of the ViT blocks.  x [rows, D] in_dtype -> y [rows, D] out_dtype;
 * gamma/beta fp32 [D]. */
int vdr_op_layernorm(const void* x, int in_dtype, void* y, int out_dtype, const float* gamma,
                     const float* beta, int64_t rows, int D, float eps, void* stream);

/* epilogues of vdr_op_linear */
typedef enum {
  VDR_EPI_BIAS = 0,      /* y = xW^T + b                      F.linear                          */
  VDR_EPI_BIAS_GELU = 1, /* y = gelu_erf(xW^T + b)            linear1 + activation (a9).  Deviation from torch: a NaN   */
                         /*   input gives -3e-8, not NaN (the device form is max(x,0) - a 2^Q(a) on v_min / v_max, which   */
                         /*   return their non-NaN operand; csrc/vdr_dev.h).  Inside a block the residual stream carries   */
                         /*   the NaN row on regardless; the classifier heads hand non-finite rows on themselves.          */
  VDR_EPI_BIAS_RESID = 2,/* y = resid + gamma*(xW^T + b)      out_proj / linear2 + residual     */
  VDR_EPI_SWIGLU = 3     /* y[:, :N/2] = silu(a)*b, (a,b) = split(xW^T + b)  DINOv2 SwiGLUFFN   */
} vdr_epilogue;

/* F.linear(x, W, b) (+ fused epilogue) — nn.Linear inside nn.MultiheadAttention /
 * TransformerEncoderLayer (models_archs.py:130-135), attn.qkv / attn.proj / mlp.fc1 / mlp.fc2.
 *   x [M, K] bf16, W [N, K] bf16 (PyTorch layout), bias fp32 [N] or NULL,
 *   resid [M, N] bf16 (EPI_BIAS_RESID; may alias y), gamma fp32 [N] or NULL (LayerScale),
 *   y [M, N] bf16 (EPI_SWIGLU: [M, N/2]).  K % 64 == 0, N % 8 == 0.
 *   `variant` selects the tile configuration (0 = library default). */
int vdr_op_linear(const void* x, const void* W, const float* bias, const void* resid,
                  const float* gamma, void* y, int64_t M, int N, int K, int epilogue, int variant,
                  void* stream);
/* The same with W in the library's packed weight layout — what vdr_finalize builds for every nn.Linear weight of a
 * model: [N/2][K/32][2][32] bf16, so that a 128-byte line holds one 32-deep K block of two neighbouring rows and the
 * operand loader of the GEMM touches whole lines.  vdr_op_pack_linear_weight converts W [N, K] bf16 (PyTorch layout)
 * into `packed` (N*K bf16, device); N even, K % 32 == 0.  Results are bitwise those of vdr_op_linear. */
int vdr_op_pack_linear_weight(const void* W, int N, int K, void* packed, void* stream);
int vdr_op_linear_packed(const void* x, const void* Wp, const float* bias, const void* resid,
                         const float* gamma, void* y, int64_t M, int N, int K, int epilogue, int variant,
                         void* stream);

/* ---- MX-fp8 operators (BASELINE config 5: "DINOv2 ViT-g/14 fp8 weights (CDNA4 fp8 MFMA)") ----------
 * An MX tensor X[rows, K] is an OCP e4m3 payload q[rows, K] (one byte per element) plus e8m0 scales, one per
 * 32 consecutive K elements, in the device layout  s[K/32][rows_pad]  (rows_pad = rows rounded up to 256; inside
 * each 64-row group rows are stored as (r, r+32) pairs).  vdr_mx_scale_bytes gives the size of that array.
 * The reference has no fp8 path (fp32 eager everywhere); these replace the same nn.Linear / nn.LayerNorm
 * call sites as vdr_op_linear / vdr_op_layernorm (models_archs.py:130-136, ViT blocks). */
size_t vdr_mx_scale_bytes(int64_t rows, int K);
/* bf16 x [rows, K] -> MX (q, scales).  K % 32 == 0. */
int vdr_op_mx_quantize(const void* x, int64_t rows, int K, void* q, void* scales, void* stream);
/* MX (q, scales) -> fp32 y [rows, K]  (test / inspection helper: value = e4m3(q) * 2^(scale-127)) */
int vdr_op_mx_dequantize(const void* q, const void* scales, int64_t rows, int K, float* y, void* stream);
/* F.layer_norm over bf16 rows with MX output (the operand of the following fp8 linear).  D % 32 == 0. */
int vdr_op_layernorm_mx(const void* x, const float* gamma, const float* beta, float eps, int64_t rows, int D,
                        void* q, void* scales, void* stream);
/* F.linear with MX operands on the block-scaled fp8 MFMA, fp32 accumulate, same epilogues as vdr_op_linear.
 *   xq/xs: MX x [M, K];  wq/ws: MX W [N, K];  y bf16 [M, N] ([M, N/2] for EPI_SWIGLU), or — when yscales is
 *   non-NULL (EPI_BIAS_GELU / EPI_SWIGLU only) — an MX tensor (y = payload, yscales) ready for the next
 *   fp8 linear.  K % 64 == 0, N % 64 == 0.  variant: 0 = 128x256 tile, 1 = 256x256, 2 = 128x128. */
int vdr_op_linear_mx(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias,
                     const void* resid, const float* gamma, void* y, void* yscales, int64_t M, int N, int K,
                     int epilogue, int variant, void* stream);

/* ---- pre/post-processing either side of the encoder --------------------------
 * What the reference does per slice on the CPU with numpy / skimage; all pointers are device pointers. */

/* prepare_image (tfds_dense_descriptor.py:30-48) for a batch of slices: gray2rgb, skimage.transform.resize
 * (order 1, mode 'reflect', anti-aliasing Gaussian when down-scaling, float64 arithmetic), HWC -> CHW; the
 * flips of flip_image (:305-324) are folded in.
 *   src        fp32 (src_dtype VDR_F32) or fp64 (VDR_F64) image(s) with ELEMENT strides (stride_b, stride_y,
 *              stride_x, stride_c): the slices of an (H, W, S[, C]) volume are a batch as they lie
 *   channels   1 (replicated to 3, target side 1024 in the reference) or 3 (target side 896)
 *   flip       0 none, 1 'horizontal' (x reversed), 2 'vertical' (y reversed)
 *   out        [batch, 3, out_side, out_side] fp32 or bf16 (out_dtype)
 *   scratch    device scratch of vdr_prepare_scratch_bytes(...) bytes (0 when nothing is down-scaled) */
size_t vdr_prepare_scratch_bytes(int batch, int h, int w, int channels, int out_side);
int vdr_op_prepare_image(const void* src, int src_dtype, int batch, int h, int w, int channels, int64_t stride_b,
                         int64_t stride_y, int64_t stride_x, int64_t stride_c, int flip, int out_side, void* out,
                         int out_dtype, void* scratch, void* stream);
/* apply_window_ct (tfds_dense_descriptor.py:287-302, windowing_ct :204-237): clip((ct - (L - W/2)) / W, 0, 1).
 *   ct fp32 (VDR_F32) or int16 (VDR_I16), n elements -> out fp32 */
int vdr_op_window_ct(const void* ct, int in_dtype, int64_t n, double width, double level, float* out, void* stream);
/* hu_to_rgb_vectorized (visualization_utils.py:128-186): HU -> uint8 RGB [n, 3].  hu fp32 / int16 / fp64. */
int vdr_op_hu_to_rgb(const void* hu, int in_dtype, int64_t n, void* rgb, void* stream);
/* crop_image / extract_roi (visualization_utils.py:93-125) of channel-last maps:
 *   src fp32 [batch, H, W, C] -> dst fp32 [batch, crop_h, crop_w, C], window origin (y0, x0), fully inside */
int vdr_op_crop_hwc(const float* src, float* dst, int batch, int H, int W, int C, int y0, int x0, int crop_h,
                    int crop_w, void* stream);
/* rotate_image (tfds_dense_descriptor.py:327-350): scipy.ndimage.rotate(vol, angle, axes=(0, 1), reshape=False,
 * mode='nearest') = scipy.ndimage.affine_transform(plane, matrix, offset, order=3, mode='nearest', prefilter=True)
 * on every (H, W) plane of the volume; float64 arithmetic in SciPy's operation order (bit-identical results).
 *   src      [h, w, planes] (planes = slices [x channels], fastest) fp64 (VDR_F64), fp32 (VDR_F32) or a boolean mask
 *            as bytes 0 / 1 (VDR_U8); out has the same shape and dtype
 *   matrix   2 x 2 row-major, offset 2: input coordinate = matrix . output index + offset (SciPy's convention;
 *            rotate() passes [[cos, sin], [-sin, cos]] and in_center - matrix . out_center)
 *   clip01   1: clip the float result to [0, 1] (the np.clip of rotate_image); masks: out = (unsigned char) t,
 *            SciPy's store into a boolean output, so `out > 0` is rotate_image's mask
 *   scratch  vdr_affine_cubic_scratch_bytes(h, w, planes) bytes: the edge-padded float64 spline coefficients */
size_t vdr_affine_cubic_scratch_bytes(int h, int w, int64_t planes);
int vdr_op_affine_cubic(const void* src, int dtype, int h, int w, int64_t planes, const double* matrix,
                        const double* offset, void* out, int clip01, void* scratch, void* stream);
/* Stage-C input sequence (train_models.py:143-182 'transformer' branch + positional_encoding_3d :30-44):
 * out[r, :] = feat[index[r], :] + PE(x_r, y_r, z_r) / 4, the masked voxels of an (h, w, S) feature volume with their
 * 3-D sinusoidal position code.
 *   feat   fp32 [positions, D], positions in (h, w, S) order;  index int64 [n] positions kept (ascending)
 *   xyz    fp64 [3][n] coordinates of the kept voxels;  expo fp64 [D / 6] = scale^(6 i / D)
 *   out    [n, D] fp32 (VDR_F32), bf16 (VDR_BF16) or fp64 (VDR_F64, what numpy produces) */
int vdr_op_voxel_sequence(const float* feat, const int64_t* index, const double* xyz, const double* expo, int64_t n,
                          int D, void* out, int out_dtype, void* stream);

/* F.scaled_dot_product_attention over a packed qkv activation — the core of
 * nn.MultiheadAttention (models_archs.py:130) / Attention.forward of the ViTs.
 *   qkv [batch*seq, 3*H*64] bf16, row = token, columns [q | k | v] each [H, 64]
 *   out [batch*seq, H*64] bf16;  softmax(q k^T / 8) v per (batch, head), no mask.
 *   variant: 0 = library choice; 1 = chunked (online softmax over 128-key chunks), 2 = persistent kernel without,
 *   4 = with its loader wave (sequences of 129..224 tokens), 3 = one workgroup per (batch, head).  All variants
 *   produce the same bits for sequences that fit one chunk. */
int vdr_op_attention(const void* qkv, void* out, int batch, int seq, int heads, int variant,
                     void* stream);

/* SAM / MedSAM Attention.forward with use_rel_pos (third-party segment_anything ImageEncoderViT, called at
 * tfds_dense_descriptor.py:123): per (window, head) softmax(q k^T dh^-0.5 + q.Rh[qh-kh] + q.Rw[qw-kw]) v.
 *   qkv   [batch*S*S, 3*H*64] bf16, `batch` windows (or whole grids) of S x S tokens, row-major (h, w)
 *   rel_pos_h / rel_pos_w  fp32 [2S-1, 64] (the block's parameters)
 *   rel   device scratch, fp32 [batch*S*S*H*Np + Np*32], Np = 2*roundup(2S-1, 32): the products of q with
 *         every relative-offset row of both tables (one MFMA GEMM) and, behind them, the packed bf16 tables
 *   out   [batch*S*S, H*64] bf16.   S in {4, 7, 10, 14} (one pass) or 64 (online softmax). */
int vdr_op_attention_relpos(const void* qkv, const float* rel_pos_h, const float* rel_pos_w, float* rel, void* out,
                            int batch, int S, int heads, void* stream);

/* nn.Conv2d(in_chans, D, kernel=p, stride=p) + flatten(2).transpose(1,2) — DINOv2 PatchEmbed,
 * the op called at tfds_dense_descriptor.py:128.
 *   images NCHW [batch, C, img, img] in_dtype; W bf16 [D, Kp] (C*p*p columns zero-padded to
 *   Kp = roundup(C*p*p, 64)); bias fp32 [D]; pos fp32 [n(+1), D] or NULL;
 *   col: device scratch of batch*n*Kp bf16 (left untouched when the images are bf16, 16-byte aligned and p is 8, 16
 *   or 32: the GEMM then gathers its operand from the images, no im2col pass);
 *   y bf16: row (b*row_stride + row_offset + i) for patch i of image b, plus pos[row_offset+i]. */
int vdr_op_patch_embed(const void* images, int in_dtype, const void* W, const float* bias,
                       const float* pos, void* col, void* y, int batch, int C, int img, int p, int D,
                       int row_stride, int row_offset, void* stream);
/* ---- measurement ------------------------------------------------------------------------ */

/* Kernel classes timed by the built-in HIP-event profiler. */
typedef enum {
  VDR_K_IM2COL = 0,
  VDR_K_GEMM_PATCH = 1,
  VDR_K_LAYERNORM = 2,
  VDR_K_GEMM_QKV = 3,
  VDR_K_ATTENTION = 4,
  VDR_K_GEMM_PROJ = 5,
  VDR_K_GEMM_FC1 = 6,
  VDR_K_GEMM_FC2 = 7,
  VDR_K_FINAL_LN = 8,
  VDR_K_ASSEMBLE = 9,
  VDR_K_CLS_TAIL = 10, /* out-projection, norm2 and MLP of the last block on the CLS rows only (vdr_config.full_last_block) */
  VDR_K_COUNT = 11
} vdr_kernel_class;

/* When enabled, vdr_forward* brackets every launch with hipEventRecord on the
 * caller's stream.  vdr_profile_read synchronises on those events and returns,
 * per class, the summed milliseconds, the launch count and the algorithmic
 * FLOPs and bytes of those launches since the last read / reset. */
int vdr_profile_enable(vdr_handle h, int on);
/* Restrict the event bracketing to the kernel classes whose bit (1 << class) is set (default: all).
 * Timing one class keeps the profiler's cost out of a throughput measurement. */
int vdr_profile_mask(vdr_handle h, uint32_t class_mask);
int vdr_profile_read(vdr_handle h, double* ms, int64_t* launches, double* flops, double* bytes,
                     int n /* = VDR_K_COUNT */);
const char* vdr_kernel_class_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* VDR_H_ */
