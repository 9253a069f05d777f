// C ABI of libvdr.so (include/vdr.h): handle, weight packing, forward orchestration, profiler.
// Host code only; every kernel lives in gemm.hip / attention.hip / rowops.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vdr.h"
#include "vdr_kernels.h"

using namespace vdr;

namespace {

thread_local std::string g_err;

uint16_t f32_to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // keep NaN a NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                  // round to nearest even
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// entry points run on the handle's device whatever the caller's current device is, and leave that one current
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

enum WKind { W_VEC_F32, W_MAT_BF16, W_PATCH_BF16, W_W12_BF16, W_W12_BIAS, W_CONV3_BF16 };

struct WSlot {
  std::string name;
  WKind kind;
  int64_t numel;      // expected fp32 elements from the caller
  int64_t rows, cols; // logical matrix shape for MAT kinds
  void* dev = nullptr;
  bool set = false;
  std::vector<float> host;  // fp32 copy kept until resolve() (LayerNorm folding needs it)
};

struct LayerW {
  const float *n1w, *n1b, *n2w, *n2b, *bqkv, *bproj, *b1, *b2, *ls1, *ls2;
  const void *wqkv, *wproj, *w1, *w2;
  // LayerNorm folded into the consuming GEMM (pre-LN image models): W' = W.diag(gamma) in bf16,
  // colsum[n] = sum_k W'[n][k], tbias[n] = sum_k beta[k] W[n][k] + b[n]
  const float *relh = nullptr, *relw = nullptr;  // SAM decomposed relative position tables
  void* reltab = nullptr;  // both tables packed as one [relpos_npad(S)][64] bf16 GEMM operand
  void *wqkv_f = nullptr, *w1_f = nullptr;
  // fp8 path: MX-fp8 copies (payload, scales) of the qkv / fc1 (w12) / fc2 (w3) weights
  void *qkv_q = nullptr, *qkv_s = nullptr, *w1_q = nullptr, *w1_s = nullptr, *w2_q = nullptr, *w2_s = nullptr;
  float *sqkv = nullptr, *tqkv = nullptr, *s1 = nullptr, *t1 = nullptr;
};

constexpr int FIN_ROWS = 1 << 16;  // finalisation counters per stream: blocks of >= 64 tile rows, i.e. launches of up to 4 M rows

struct ProfEvent {
  int cls;
  hipEvent_t a, b;
};

}  // namespace

struct vdr_model {
  vdr_config cfg;
  int device = 0;
  int n_patches = 0, n_tokens = 0, Kp = 0;
  std::vector<WSlot> slots;
  std::map<std::string, int> index;
  std::vector<LayerW> layers;
  const void* w_patch = nullptr;
  const void *w_neck0 = nullptr, *w_neck2 = nullptr;
  const float *neck1w = nullptr, *neck1b = nullptr, *neck3w = nullptr, *neck3b = nullptr;
  const float *b_patch = nullptr, *cls = nullptr, *pos = nullptr, *normw = nullptr, *normb = nullptr,
              *inw = nullptr, *inb = nullptr;
  bool resolved = false;
  bool ln_fuse = false;
  // GEMM weights in the pair-interleaved layout the operand loader wants (gemm_kernels.h), keyed by the row-major
  // device copy they were packed from; built by resolve()
  std::map<const void*, void*> w_il;
  std::string err;
  // internal streams (cfg.streams > 1)
  std::vector<hipStream_t> streams;
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
  // fp8_cls_bf16: one side stream (+ fork / join events) per stream a forward can run on, created by vdr_finalize; the CLS
  // rows' bf16 MLP of a block runs there under the MX-fp8 GEMMs of the other rows
  std::vector<hipStream_t> aux;
  std::vector<hipEvent_t> aux_fork, aux_join;
  int cur_aux = 0;  // index of the stream run_blocks is being called for (set by the forward's micro-batch loop)
  // producer-side LayerNorm finalisation (GemmArgs::fin_stats): zeroed counters, one per block of tile rows and per stream
  // a forward can run on (FIN_ROWS each; the kernels leave them zeroed); stats_fresh: the (mean, rstd) buffer of the
  // workspace in use already holds the statistics of the stream's current contents (set by gemm(), taken by ln_consumer())
  uint32_t* fin_cnt = nullptr;
  bool stats_fresh = false;
  // profiler
  bool prof = false;
  uint32_t prof_mask = 0xffffffffu;
  std::vector<ProfEvent> ev_used, ev_free;
  int prof_as = -1;  // >= 0: profiler class every launch is booked under (instead of its own)
  double p_flops[VDR_K_COUNT] = {0}, p_bytes[VDR_K_COUNT] = {0};
  int64_t p_launch[VDR_K_COUNT] = {0};
};

namespace {

#define VDR_TRY(expr, what)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return hip_fail(m, _e, what); \
  } while (0)

int hip_fail(vdr_handle h, hipError_t e, const char* what);

int fail(vdr_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}

int hip_fail(vdr_handle h, hipError_t e, const char* what) {
  return fail(h, VDR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

void add_slot(vdr_model* m, const std::string& name, WKind kind, int64_t rows, int64_t cols) {
  WSlot s;
  s.name = name;
  s.kind = kind;
  s.rows = rows;
  s.cols = cols;
  s.numel = rows * cols;
  m->index[name] = (int)m->slots.size();
  m->slots.push_back(s);
}

void build_slots(vdr_model* m) {
  const vdr_config& c = m->cfg;
  const int D = c.dim, F = c.mlp_hidden;
  if (c.patch) {
    add_slot(m, "patch_embed.proj.weight", W_PATCH_BF16, D, (int64_t)c.in_chans * c.patch * c.patch);
    add_slot(m, "patch_embed.proj.bias", W_VEC_F32, 1, D);
  }
  if (c.has_cls) add_slot(m, "cls_token", W_VEC_F32, 1, D);
  if (c.has_pos) add_slot(m, "pos_embed", W_VEC_F32, m->n_tokens, D);
  if (c.input_ln) {
    add_slot(m, "input_norm.weight", W_VEC_F32, 1, D);
    add_slot(m, "input_norm.bias", W_VEC_F32, 1, D);
  }
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    add_slot(m, p + "norm1.weight", W_VEC_F32, 1, D);
    add_slot(m, p + "norm1.bias", W_VEC_F32, 1, D);
    add_slot(m, p + "attn.qkv.weight", W_MAT_BF16, 3 * D, D);
    add_slot(m, p + "attn.qkv.bias", W_VEC_F32, 1, 3 * D);
    if (c.window > 0) {
      const int size = ((c.global_mask >> i) & 1) ? c.img / c.patch : c.window;
      add_slot(m, p + "attn.rel_pos_h", W_VEC_F32, 2 * size - 1, 64);
      add_slot(m, p + "attn.rel_pos_w", W_VEC_F32, 2 * size - 1, 64);
    }
    add_slot(m, p + "attn.proj.weight", W_MAT_BF16, D, D);
    add_slot(m, p + "attn.proj.bias", W_VEC_F32, 1, D);
    if (c.layerscale) add_slot(m, p + "ls1.gamma", W_VEC_F32, 1, D);
    add_slot(m, p + "norm2.weight", W_VEC_F32, 1, D);
    add_slot(m, p + "norm2.bias", W_VEC_F32, 1, D);
    if (c.act == VDR_ACT_SWIGLU) {
      add_slot(m, p + "mlp.w12.weight", W_W12_BF16, 2 * F, D);
      add_slot(m, p + "mlp.w12.bias", W_W12_BIAS, 1, 2 * F);
      add_slot(m, p + "mlp.w3.weight", W_MAT_BF16, D, F);
      add_slot(m, p + "mlp.w3.bias", W_VEC_F32, 1, D);
    } else {
      add_slot(m, p + "mlp.fc1.weight", W_MAT_BF16, F, D);
      add_slot(m, p + "mlp.fc1.bias", W_VEC_F32, 1, F);
      add_slot(m, p + "mlp.fc2.weight", W_MAT_BF16, D, F);
      add_slot(m, p + "mlp.fc2.bias", W_VEC_F32, 1, D);
    }
    if (c.layerscale) add_slot(m, p + "ls2.gamma", W_VEC_F32, 1, D);
  }
  if (c.pre_ln && c.window == 0) {
    add_slot(m, "norm.weight", W_VEC_F32, 1, D);
    add_slot(m, "norm.bias", W_VEC_F32, 1, D);
  }
  if (c.window > 0) {
    const int C = c.neck_chans;
    add_slot(m, "neck.0.weight", W_MAT_BF16, C, D);
    add_slot(m, "neck.1.weight", W_VEC_F32, 1, C);
    add_slot(m, "neck.1.bias", W_VEC_F32, 1, C);
    add_slot(m, "neck.2.weight", W_CONV3_BF16, C, (int64_t)C * 9);
    add_slot(m, "neck.3.weight", W_VEC_F32, 1, C);
    add_slot(m, "neck.3.bias", W_VEC_F32, 1, C);
  }
}

const void* dev_of(vdr_model* m, const std::string& name) {
  auto it = m->index.find(name);
  return it == m->index.end() ? nullptr : m->slots[it->second].dev;
}

float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

const std::vector<float>* host_of(vdr_model* m, const std::string& name) {
  auto it = m->index.find(name);
  return it == m->index.end() ? nullptr : &m->slots[it->second].host;
}

// tuning knobs are read from the environment in tuning builds only (-DVDR_TUNING, `make tuning`); the shipped library
// has no environment dependence
int env_int(const char* name, int dflt) {
#ifdef VDR_TUNING
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

bool ln_fusion_wanted(const vdr_model* m) {
  const vdr_config& c = m->cfg;
  if (c.no_ln_fold || env_int("VDR_LN_FUSE", 1) == 0) return false;
  return c.patch && c.pre_ln && !c.input_ln && !c.fp8 && (c.dim % 64) == 0;
}

// W' = W.diag(gamma) (bf16), colsum, tbias for one linear layer; `perm` (optional) maps packed row -> source row
int fold_ln(vdr_model* m, const std::vector<float>& W, const std::vector<float>& b, const std::vector<float>& gam,
            const std::vector<float>& bet, int64_t N, int64_t K, const std::vector<int64_t>* perm, void** wf_dev,
            float** colsum_dev, float** tbias_dev) {
  std::vector<uint16_t> wf((size_t)N * K);
  std::vector<float> cs(N), tb(N);
  for (int64_t pr = 0; pr < N; ++pr) {
    const int64_t n = perm ? (*perm)[pr] : pr;
    const float* w = &W[(size_t)n * K];
    double s = 0.0, t = 0.0;
    for (int64_t k = 0; k < K; ++k) {
      const uint16_t h = f32_to_bf16(gam[k] * w[k]);
      wf[(size_t)pr * K + k] = h;
      s += (double)bf16_to_f32(h);
      t += (double)bet[k] * (double)w[k];
    }
    cs[pr] = (float)s;
    tb[pr] = (float)(t + (double)b[n]);
  }
  if (!*wf_dev) VDR_TRY(hipMalloc(wf_dev, wf.size() * 2 + 256), "hipMalloc(folded weight)");
  if (!*colsum_dev) VDR_TRY(hipMalloc((void**)colsum_dev, (size_t)N * 4 + 256), "hipMalloc(colsum)");
  if (!*tbias_dev) VDR_TRY(hipMalloc((void**)tbias_dev, (size_t)N * 4 + 256), "hipMalloc(tbias)");
  VDR_TRY(hipMemcpy(*wf_dev, wf.data(), wf.size() * 2, hipMemcpyHostToDevice), "hipMemcpy(folded weight)");
  VDR_TRY(hipMemcpy(*colsum_dev, cs.data(), (size_t)N * 4, hipMemcpyHostToDevice), "hipMemcpy(colsum)");
  VDR_TRY(hipMemcpy(*tbias_dev, tb.data(), (size_t)N * 4, hipMemcpyHostToDevice), "hipMemcpy(tbias)");
  return VDR_OK;
}

int resolve(vdr_model* m) {
  for (auto& s : m->slots)
    if (!s.set) return fail(m, VDR_ERR_INCOMPLETE, "weight not set: " + s.name);
  if (ln_fusion_wanted(m))
    for (auto& s : m->slots)
      if (s.host.empty())
        return fail(m, VDR_ERR_INCOMPLETE, "weights changed after the first forward: set every weight again (" + s.name + ")");
  const vdr_config& c = m->cfg;
  m->w_patch = dev_of(m, "patch_embed.proj.weight");
  m->b_patch = (const float*)dev_of(m, "patch_embed.proj.bias");
  m->cls = (const float*)dev_of(m, "cls_token");
  m->pos = (const float*)dev_of(m, "pos_embed");
  m->inw = (const float*)dev_of(m, "input_norm.weight");
  m->inb = (const float*)dev_of(m, "input_norm.bias");
  m->normw = (const float*)dev_of(m, "norm.weight");
  m->normb = (const float*)dev_of(m, "norm.bias");
  m->w_neck0 = dev_of(m, "neck.0.weight");
  m->w_neck2 = dev_of(m, "neck.2.weight");
  m->neck1w = (const float*)dev_of(m, "neck.1.weight");
  m->neck1b = (const float*)dev_of(m, "neck.1.bias");
  m->neck3w = (const float*)dev_of(m, "neck.3.weight");
  m->neck3b = (const float*)dev_of(m, "neck.3.bias");
  m->layers.resize(c.layers);
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    LayerW& L = m->layers[i];
    L.n1w = (const float*)dev_of(m, p + "norm1.weight");
    L.n1b = (const float*)dev_of(m, p + "norm1.bias");
    L.n2w = (const float*)dev_of(m, p + "norm2.weight");
    L.n2b = (const float*)dev_of(m, p + "norm2.bias");
    L.wqkv = dev_of(m, p + "attn.qkv.weight");
    L.bqkv = (const float*)dev_of(m, p + "attn.qkv.bias");
    L.wproj = dev_of(m, p + "attn.proj.weight");
    L.bproj = (const float*)dev_of(m, p + "attn.proj.bias");
    L.relh = (const float*)dev_of(m, p + "attn.rel_pos_h");
    L.relw = (const float*)dev_of(m, p + "attn.rel_pos_w");
    L.ls1 = (const float*)dev_of(m, p + "ls1.gamma");
    L.ls2 = (const float*)dev_of(m, p + "ls2.gamma");
    if (c.act == VDR_ACT_SWIGLU) {
      L.w1 = dev_of(m, p + "mlp.w12.weight");
      L.b1 = (const float*)dev_of(m, p + "mlp.w12.bias");
      L.w2 = dev_of(m, p + "mlp.w3.weight");
      L.b2 = (const float*)dev_of(m, p + "mlp.w3.bias");
    } else {
      L.w1 = dev_of(m, p + "mlp.fc1.weight");
      L.b1 = (const float*)dev_of(m, p + "mlp.fc1.bias");
      L.w2 = dev_of(m, p + "mlp.fc2.weight");
      L.b2 = (const float*)dev_of(m, p + "mlp.fc2.bias");
    }
  }
  m->ln_fuse = ln_fusion_wanted(m);
  if (m->ln_fuse) {
    VDR_TRY(hipSetDevice(m->device), "hipSetDevice");
    const int64_t D = c.dim, F = c.mlp_hidden;
    for (int i = 0; i < c.layers; ++i) {
      const std::string p = "blocks." + std::to_string(i) + ".";
      LayerW& L = m->layers[i];
      int rc = fold_ln(m, *host_of(m, p + "attn.qkv.weight"), *host_of(m, p + "attn.qkv.bias"), *host_of(m, p + "norm1.weight"),
                       *host_of(m, p + "norm1.bias"), 3 * D, D, nullptr, &L.wqkv_f, &L.sqkv, &L.tqkv);
      if (rc) return rc;
      if (c.act == VDR_ACT_SWIGLU) {
        std::vector<int64_t> perm(2 * F);
        for (int64_t pr = 0; pr < 2 * F; ++pr) {
          const int64_t blk = pr / 64, t = pr % 64;
          perm[pr] = t < 32 ? blk * 32 + t : F + blk * 32 + (t - 32);
        }
        rc = fold_ln(m, *host_of(m, p + "mlp.w12.weight"), *host_of(m, p + "mlp.w12.bias"), *host_of(m, p + "norm2.weight"),
                     *host_of(m, p + "norm2.bias"), 2 * F, D, &perm, &L.w1_f, &L.s1, &L.t1);
      } else {
        rc = fold_ln(m, *host_of(m, p + "mlp.fc1.weight"), *host_of(m, p + "mlp.fc1.bias"), *host_of(m, p + "norm2.weight"),
                     *host_of(m, p + "norm2.bias"), F, D, nullptr, &L.w1_f, &L.s1, &L.t1);
      }
      if (rc) return rc;
    }
  }
  if (c.fp8) {
    VDR_TRY(hipSetDevice(m->device), "hipSetDevice");
    const int D = c.dim, F = c.mlp_hidden;
    const int N1 = c.act == VDR_ACT_SWIGLU ? 2 * F : F;
    auto quant = [&](const void* wdev, int N, int K, void** q, void** sc) -> int {
      if (!*q) VDR_TRY(hipMalloc(q, (size_t)N * K + 256), "hipMalloc(fp8 weight)");
      if (!*sc) VDR_TRY(hipMalloc(sc, mx_scale_bytes(N, K) + 256), "hipMalloc(fp8 weight scales)");
      VDR_TRY(hipMemset(*sc, 0, mx_scale_bytes(N, K)), "hipMemset(fp8 weight scales)");
      VDR_TRY(launch_mx_quant(wdev, N, K, K, *q, *sc, nullptr), "mx_quant(weight)");
      return VDR_OK;
    };
    // the e4m3 payloads go to the packed (pair-interleaved) layout too: as bytes, [N][K] fp8 is [N][K/2] bf16, and a
    // 32-element bf16 block is one 64-element MX unit
    auto pack8 = [&](void* q, int N, int K) -> int {
      void*& dst = m->w_il[q];
      if (!dst) VDR_TRY(hipMalloc(&dst, (size_t)N * K + 256), "hipMalloc(interleaved fp8 weight)");
      VDR_TRY(launch_w_interleave(q, dst, N, K / 2, K / 2, nullptr), "w_interleave(fp8)");
      return VDR_OK;
    };
    for (int i = 0; i < c.layers; ++i) {
      LayerW& L = m->layers[i];
      int rc;
      if ((rc = quant(L.wqkv, 3 * D, D, &L.qkv_q, &L.qkv_s)) || (rc = pack8(L.qkv_q, 3 * D, D))) return rc;
      if ((rc = quant(L.w1, N1, D, &L.w1_q, &L.w1_s)) || (rc = pack8(L.w1_q, N1, D))) return rc;
      if ((rc = quant(L.w2, D, F, &L.w2_q, &L.w2_s)) || (rc = pack8(L.w2_q, D, F))) return rc;
    }
    VDR_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  if (c.window > 0) {
    VDR_TRY(hipSetDevice(m->device), "hipSetDevice");
    const int g = c.img / c.patch;
    for (int i = 0; i < c.layers; ++i) {
      LayerW& L = m->layers[i];
      const int S = (c.global_mask >> i) & 1 ? g : c.window;
      if (!L.reltab) VDR_TRY(hipMalloc(&L.reltab, (size_t)relpos_npad(S) * 64 * 2 + 256), "hipMalloc(rel-pos table)");
      VDR_TRY(launch_relpos_pack(L.relh, L.relw, L.reltab, S, nullptr), "relpos_pack");
    }
    VDR_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  {
    // pair-interleaved copies of every GEMM weight (whole-line operand loads, gemm_kernels.h)
    VDR_TRY(hipSetDevice(m->device), "hipSetDevice");
    const int D = c.dim, F = c.mlp_hidden;
    const int N1 = c.act == VDR_ACT_SWIGLU ? 2 * F : F;
    auto pack = [&](const void* w, int N, int K) -> int {
      if (!w || (N & 1) || (K & 31)) return VDR_OK;
      void*& dst = m->w_il[w];
      if (!dst) VDR_TRY(hipMalloc(&dst, (size_t)N * K * 2 + 256), "hipMalloc(interleaved weight)");
      VDR_TRY(launch_w_interleave(w, dst, N, K, K, nullptr), "w_interleave");
      return VDR_OK;
    };
    int rc = VDR_OK;
    if (c.patch && (rc = pack(m->w_patch, D, m->Kp))) return rc;
    for (int i = 0; i < c.layers && !c.fp8; ++i) {
      const LayerW& L = m->layers[i];
      if ((rc = pack(L.wqkv, 3 * D, D)) || (rc = pack(L.wqkv_f, 3 * D, D))) return rc;  // (SAM blocks use the unfolded qkv)
      if ((rc = pack(L.wproj, D, D))) return rc;
      if ((rc = pack(L.w1, N1, D)) || (rc = pack(L.w1_f, N1, D))) return rc;
      if ((rc = pack(L.w2, D, F))) return rc;
    }
    if (c.fp8)  // the out-projection stays bf16 on the fp8 path
      for (int i = 0; i < c.layers; ++i)
        if ((rc = pack(m->layers[i].wproj, D, D))) return rc;
    if (c.fp8 && c.fp8_cls_bf16 && c.has_cls) {  // the CLS rows' MLP runs on the bf16 weights
      for (int i = 0; i < c.layers; ++i)
        if ((rc = pack(m->layers[i].w1, N1, D)) || (rc = pack(m->layers[i].w2, D, F))) return rc;
      const size_t ns = (size_t)(c.streams > 1 ? (c.streams > 8 ? 8 : c.streams) : 1);
      while (m->aux.size() < ns) {
        hipStream_t st;
        hipEvent_t ea, eb;
        VDR_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate(CLS side stream)");
        VDR_TRY(hipEventCreateWithFlags(&ea, hipEventDisableTiming), "hipEventCreate");
        VDR_TRY(hipEventCreateWithFlags(&eb, hipEventDisableTiming), "hipEventCreate");
        m->aux.push_back(st);
        m->aux_fork.push_back(ea);
        m->aux_join.push_back(eb);
      }
    }
    if (c.window > 0) {
      if ((rc = pack(m->w_neck0, c.neck_chans, D))) return rc;
      if ((rc = pack(m->w_neck2, c.neck_chans, 9 * c.neck_chans))) return rc;
    }
    VDR_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  if (!m->fin_cnt) {
    VDR_TRY(hipMalloc(&m->fin_cnt, (size_t)8 * FIN_ROWS * 4), "hipMalloc(LayerNorm finalisation counters)");
    VDR_TRY(hipMemset(m->fin_cnt, 0, (size_t)8 * FIN_ROWS * 4), "hipMemset");
    VDR_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  for (auto& sl : m->slots) std::vector<float>().swap(sl.host);  // host copies are no longer needed
  m->resolved = true;
  return VDR_OK;
}

// every GEMM of the forward goes through here: the weight is swapped for its interleaved copy
hipError_t launch_gemm_w(vdr_model* m, GemmArgs& g, int epi, int variant, hipStream_t s) {
  auto it = m->w_il.find(g.W);
  if (it != m->w_il.end()) {
    g.W = it->second;
    g.w_interleaved = 1;
  }
  return launch_gemm(g, epi, variant, s);
}

// ---- workspace carving ------------------------------------------------------------------------
struct Carve {
  char *x, *h, *qkv, *o, *u;
  char* hg = nullptr;    // SAM: LN1 output of the global blocks (h holds the windowed, zero-padded order)
  float* rel = nullptr;  // SAM: rel-pos products T[tokens][heads][relpos_npad(S)] (q . every table row)
  char *hs = nullptr, *us = nullptr, *os = nullptr;  // fp8 path: e8m0 scales of the MX activations kept in h, u, o
  char *cls_h = nullptr, *cls_u = nullptr, *cls_x = nullptr;  // fp8_cls_bf16: norm2 / activation / new residual rows of the CLS rows
  float *x32 = nullptr, *xc32 = nullptr;  // resid_fp32: fp32 master copy of the residual stream [Mp, D] / of the compact CLS rows
  float *part, *stats;  // LayerNorm partial sums [D/64][Mp][2] and (mean, rstd) [Mp][2]
  int64_t Mp;
  size_t total;
};

// vdr_config.resid_fp32 applies to the bf16 path of pre-LN image models (plain ViTs: no SAM windows, no fp8)
bool resid_fp32_on(const vdr_config& c) { return c.resid_fp32 && c.pre_ln && c.patch && !c.fp8 && c.window == 0; }

Carve carve(const vdr_model* m, char* base, int mb, int ntok) {
  const vdr_config& c = m->cfg;
  size_t rows = (size_t)mb * ntok;
  size_t rel_floats = 0;
  if (c.window > 0) {
    const size_t g = c.img / c.patch, ws = c.window, nw = (g + ws - 1) / ws;
    const size_t wtok = nw * nw * ws * ws;
    if (wtok > (size_t)ntok) rows = (size_t)mb * wtok;
    const size_t rw = (size_t)mb * wtok * c.heads * relpos_npad((int)ws), rg = (size_t)mb * ntok * c.heads * relpos_npad((int)g);
    rel_floats = rw > rg ? rw : rg;
  }
  const size_t Mp = (size_t)round_up((int)rows, 256) + 256;
  const size_t D = c.dim, F = c.mlp_hidden;
  size_t off = 0;
  Carve w;
  auto take = [&](size_t bytes) {
    char* p = base + off;
    off += align256(bytes);
    return p;
  };
  w.x = take(Mp * D * 2);
  w.h = take(Mp * D * 2);
  w.qkv = take(Mp * 3 * D * 2);
  w.o = take(Mp * D * 2);
  size_t ub = Mp * F * 2;
  if (c.patch) {
    const size_t colb = (size_t)mb * m->n_patches * m->Kp * 2 + 4096;
    if (colb > ub) ub = colb;
  }
  if (c.window > 0) {
    const size_t col3 = (size_t)mb * ntok * 9 * c.neck_chans * 2 + 4096;
    if (col3 > ub) ub = col3;
  }
  w.u = take(ub);
  if (c.window > 0) {
    w.hg = take(Mp * D * 2);
    w.rel = (float*)take(rel_floats * 4 + 256);
  }
  if (c.fp8) {
    w.hs = take(mx_scale_bytes((int64_t)Mp, (int)D));
    w.us = take(mx_scale_bytes((int64_t)Mp, (int)F));
    w.os = take(mx_scale_bytes((int64_t)Mp, (int)D));
    if (c.fp8_cls_bf16 && c.has_cls) {
      const size_t rb = (size_t)round_up(mb, 256);
      w.cls_h = take(rb * D * 2);
      w.cls_u = take(rb * F * 2);
      w.cls_x = take(rb * D * 2);
    }
  }
  if (resid_fp32_on(c)) {
    w.x32 = (float*)take(Mp * D * 4);
    w.xc32 = (float*)take((size_t)round_up(mb, 256) * D * 4);
  }
  w.part = (float*)take((size_t)(D / 64 + 1) * Mp * 8);
  w.stats = (float*)take(Mp * 8);
  w.Mp = (int64_t)Mp;
  w.total = off;
  return w;
}

int num_streams(const vdr_model* m) { return m->cfg.streams > 1 ? (m->cfg.streams > 8 ? 8 : m->cfg.streams) : 1; }

int default_micro_batch(const vdr_model* m, int batch) {
  if (m->cfg.micro_batch > 0) return m->cfg.micro_batch < batch ? m->cfg.micro_batch : batch;
  const int ns = num_streams(m);
  return (batch + ns - 1) / ns;
}

// fork: the internal streams wait for everything already enqueued on the caller's stream
int fork_streams(vdr_model* m, hipStream_t caller) {
  const int ns = num_streams(m);
  if (ns == 1) return VDR_OK;
  if (m->streams.empty()) {
    m->streams.resize(ns);
    m->ev_join.resize(ns);
    for (int i = 0; i < ns; ++i) {
      if (hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess) return VDR_ERR_HIP;
      if (hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming) != hipSuccess) return VDR_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess) return VDR_ERR_HIP;
  }
  if (hipEventRecord(m->ev_fork, caller) != hipSuccess) return VDR_ERR_HIP;
  for (int i = 0; i < ns; ++i)
    if (hipStreamWaitEvent(m->streams[i], m->ev_fork, 0) != hipSuccess) return VDR_ERR_HIP;
  return VDR_OK;
}

// join: the caller's stream waits for every internal stream
int join_streams(vdr_model* m, hipStream_t caller) {
  const int ns = num_streams(m);
  if (ns == 1) return VDR_OK;
  for (int i = 0; i < ns; ++i) {
    if (hipEventRecord(m->ev_join[i], m->streams[i]) != hipSuccess) return VDR_ERR_HIP;
    if (hipStreamWaitEvent(caller, m->ev_join[i], 0) != hipSuccess) return VDR_ERR_HIP;
  }
  return VDR_OK;
}

// ---- profiler -----------------------------------------------------------------------------------
struct Scope {
  vdr_model* m;
  hipStream_t s;
  ProfEvent e;
  bool on;
  Scope(vdr_model* m_, hipStream_t s_, int cls, double flops, double bytes) : m(m_), s(s_) {
    if (m->prof_as >= 0) cls = m->prof_as;  // (block_tail_cls: its small launches are one class of their own)
    on = m->prof && ((m->prof_mask >> cls) & 1u);
    if (!on) return;
    if (!m->ev_free.empty()) {
      e = m->ev_free.back();
      m->ev_free.pop_back();
    } else {
      hipEventCreate(&e.a);
      hipEventCreate(&e.b);
    }
    e.cls = cls;
    m->p_flops[cls] += flops;
    m->p_bytes[cls] += bytes;
    m->p_launch[cls] += 1;
    hipEventRecord(e.a, s);
  }
  ~Scope() {
    if (!on) return;
    hipEventRecord(e.b, s);
    m->ev_used.push_back(e);
  }
};

// tile configuration per GEMM class; VDR_GEMM_VARIANT overrides all of them (tuning aid)
int gemm_variant_for(int cls, int64_t M = 1 << 30, int N = 1 << 30) {
  VDR_KNOB int forced = env_int("VDR_GEMM_VARIANT", -1);
  if (forced >= 0) return forced;
  // small problems (a single 1024^2 SAM slice is M = 4096): fewer 128x256 tiles than CUs -> 128x128 tiles
  // (measured, MedSAM batch 1: fc2 0.72 -> 0.57 ms, proj 0.35 -> 0.31 ms per forward)
  if (cls != VDR_K_GEMM_QKV && ((M + 127) / 128) * ((N + 255) / 256) < 256) {
    VDR_KNOB int small = env_int("VDR_GEMM_VARIANT_SMALL", -1);
    if (small >= 0) return small;
    // ring4 128x128 tiles (variant 28).  Measured at M = 4096 / 4900 (one MedSAM slice), interleaved rounds: against
    // ring3 128x128 (24) proj 18.8 -> 17.7 us, fc2 43.9 -> 41.0 us.  Variant 25 (the K loop split across two wave groups
    // of the workgroup) is faster still (16.5 / 37.1 us) but sums K in a different order than the big-batch kernels,
    // so a row would no longer be bitwise independent of the batch it travels in: tuning builds only
    // (VDR_GEMM_VARIANT_SMALL=25), the default keeps rows batch-invariant.
    return 28;
  }
  // measured per shape at M = 50432 (tools/kbench.py): 16 waves per CU with 64-register accumulators
  // (wave tile 64x64) beat 8 waves with 128-register accumulators on every shape
  // per-class override for tuning: VDR_GEMM_VARIANT_QKV / _PROJ / _FC1 / _FC2
  VDR_KNOB int o_qkv = env_int("VDR_GEMM_VARIANT_QKV", -1), o_proj = env_int("VDR_GEMM_VARIANT_PROJ", -1),
                   o_fc1 = env_int("VDR_GEMM_VARIANT_FC1", -1), o_fc2 = env_int("VDR_GEMM_VARIANT_FC2", -1);
  const int o = cls == VDR_K_GEMM_QKV ? o_qkv : cls == VDR_K_GEMM_PROJ ? o_proj : cls == VDR_K_GEMM_FC1 ? o_fc1
                : cls == VDR_K_GEMM_FC2 ? o_fc2 : -1;
  if (o >= 0) return o;
  // ring4 (whole-line operand staging: packed weights + 64-deep activation pieces), 128x256 tile, 8 waves, two
  // workgroups per CU.  Measured at M = 50432, interleaved rounds in one process, weights packed in both arms
  // (tools/kbench.py, alternating order): against ring3 128x256 qkv 0.181 -> 0.176 ms, proj 0.086 -> 0.078, fc1
  // 0.267 -> 0.257, fc2 0.246 -> 0.230; the 256x256 forms (ring3 23, ring4 27) lose on every shape.
  return 26;
}

// im2col-free patchify (GemmArgs::patch_p): bf16 NCHW images, 16-byte aligned, patch side 8 / 16 / 32, a ring4 tile variant
bool patch_gather_ok(int in_dtype, int patch, int variant, const void* images) {
  return in_dtype == VDR_BF16 && (patch == 8 || patch == 16 || patch == 32) && variant >= 26 && variant <= 28 &&
         ((uintptr_t)images & 15) == 0;
}

#ifndef VDR_GEMM_8P_DEFAULT
#define VDR_GEMM_8P_DEFAULT 3  // qkv + fc1, measured in the forward per launch: qkv 172 -> 147 us; fc1 (erf-GELU) 257 -> 247 once the two wave sets finish a half tile in the same slot (gemm_8p.hip), ViT-L/14@336 fc1 7.62 -> 6.89 ms per step
#endif

struct LnFold {
  const float* stats = nullptr;   // consumer: (mean, rstd) per row
  const float* colsum = nullptr;  // consumer: column sums of the folded weight
  float* part = nullptr;          // producer: partial sums out
  int64_t part_stride = 0;
  const float* cpart = nullptr;   // consumer: the producers' partials, finalised inside the GEMM (instead of stats)
  int groups = 0;
  int64_t cstride = 0;
  float inv_d = 0.0f, eps = 0.0f;
  float* fin_stats = nullptr;     // producer: finalise the statistics of the rows it completes here (see finalize_rows_if_last)
};

// which GEMM classes take tile variant 31 when the launch is eligible (gemm_8p_eligible): measured per class in the
// forward; tuning builds: VDR_GEMM_8P = bit mask (1 qkv, 2 fc1), -1 = the default
bool use_8p(const vdr_model* m, int cls) {
  (void)m;
  VDR_KNOB int mask_env = env_int("VDR_GEMM_8P", -1);
  const int mask = mask_env >= 0 ? mask_env : VDR_GEMM_8P_DEFAULT;
  return (cls == VDR_K_GEMM_QKV && (mask & 1)) || (cls == VDR_K_GEMM_FC1 && (mask & 2));
}

// Where the (sum, sumsq) partials of the residual stream become (mean, rstd): inside the consuming GEMM when it runs
// a ring3 variant (22-24; every workgroup finalises its own rows in LDS while its ring fills: no launch), otherwise by
// ln_finalize_kernel ahead of it.  Same arithmetic either way: results are bitwise equal.  Measured (same box,
// alternating runs): the fetch of the partials adds about a microsecond to every tile's prologue, so it pays where a
// launch is a handful of tile rounds (MedSAM batch 1: 2.694 -> 2.669 ms, batch 4: 6.606 -> 6.588) and costs where it
// is many (ViT-B batch 256: qkv +3 %, fc1 +4 % against 0.15 ms of ln_finalize launches: 12.15 -> 12.19 ms): taken
// for launches of at most 2048 tiles.  VDR_LN_IN_GEMM=0 / 1 forces the separate kernel / the in-GEMM form.
bool ln_stats_in_gemm(int cls, int64_t M, int N, int groups) {
  VDR_KNOB int mode = env_int("VDR_LN_IN_GEMM", -1);
  const int v = gemm_variant_for(cls, M, N);
  if (mode == 0 || groups > 16 || v < 22 || v == 25 || v > 28) return false;
  if (mode == 1) return true;
  // (a launch the 8-phase kernel takes reads finalised statistics: it has no in-GEMM finalisation)
  if (use_8p(nullptr, cls) && gemm_8p_shape_ok(M, N)) return false;
  return ((M + 127) / 128) * ((N + 255) / 256) <= 2048;
}

// consumer side of the fold for one GEMM: hands it the partials, or finalises the statistics ahead of it
int ln_consumer(vdr_model* m, hipStream_t s, int cls, int64_t M, int N, int D, const Carve& w, LnFold* cons) {
  const int groups = D / 64;
  if (ln_stats_in_gemm(cls, M, N, groups)) {
    cons->stats = nullptr;
    cons->cpart = w.part;
    cons->groups = groups;
    cons->cstride = w.Mp;
    cons->inv_d = 1.0f / (float)D;
    cons->eps = m->cfg.ln_eps;
    m->stats_fresh = false;
    return VDR_OK;
  }
  if (!m->stats_fresh) {  // (otherwise the residual GEMM that wrote the stream finalised its rows' statistics itself)
    Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)M * (groups + 1) * 8);
    VDR_TRY(launch_ln_finalize(w.part, groups, w.Mp, w.stats, M, D, m->cfg.ln_eps, s), "ln_finalize");
  }
  m->stats_fresh = false;
  cons->cpart = nullptr;
  cons->stats = w.stats;
  return VDR_OK;
}

int gemm(vdr_model* m, hipStream_t s, int cls, const void* A, const void* W, const float* bias, const void* resid,
         const float* gamma, void* C, int64_t M, int N, int K, int ldc, int epi, const LnFold& ln = LnFold(),
         int64_t lda = 0, int64_t ldr = 0,  // lda / ldr: row strides of A / resid when they are not K / ldc
         const float* resid32 = nullptr, float* C32 = nullptr,  // resid_fp32: the fp32 residual stream in / out (same strides)
         int64_t a_rows = 0) {  // rows of A / of the fold's row statistics that are readable (workspace buffers: Mp); 0 = unknown
  GemmArgs g{};
  g.ln_stats = ln.stats;
  g.colsum = ln.colsum;
  g.ln_part = ln.part;
  g.part_stride = ln.part_stride;
  g.ln_cpart = ln.cpart;
  g.ln_groups = ln.groups;
  g.ln_cstride = ln.cstride;
  g.ln_inv_d = ln.inv_d;
  g.ln_eps = ln.eps;
  g.A = A;
  g.W = W;
  g.bias = bias;
  g.resid = resid;
  g.gamma = gamma;
  g.C = C;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda ? lda : K;
  g.ldw = K;
  g.ldc = ldc;
  g.ldr = ldr ? ldr : ldc;
  g.omap = identity_map();
  g.resid32 = resid32;
  g.C32 = C32;
  const double outw = epi == EPI_SWIGLU ? N / 2 : N;
  Scope sc(m, s, cls, 2.0 * M * N * K,
           2.0 * ((double)M * K + (double)N * K + (double)M * outw * (resid32 ? 5 : resid ? 2 : 1)));  // (fp32 in + fp32 out + bf16 out)
  // Tile variant 31 (gemm_8p.hip) for the write-once linears of large launches (plain weight layout).  The workspace
  // buffers A points into hold Mp >= M + 256 rows, so a ragged last 256-row tile reads rows that exist; they are never stored.
  if (use_8p(m, cls) && lda == 0 && a_rows > 0) {
    g.a_rows = a_rows;
    if (gemm_8p_eligible(g, epi)) {
      VDR_TRY(launch_gemm(g, epi, 31, s), "gemm_8p");
      return VDR_OK;
    }
  }
  const int variant = gemm_variant_for(cls, g.M, N);
  // a producer of LayerNorm partials on a ring4 tile variant also finalises them (finalize_rows_if_last): the consumer's
  // ln_finalize launch is not needed then (ln_consumer takes m->stats_fresh)
  bool fin = false;
  if (ln.fin_stats && ln.part && m->fin_cnt && m->cfg.ln_fin_fused && variant >= 26 && variant <= 29 && epi == EPI_BIAS_RESID &&
      (g.M + 63) / 64 <= FIN_ROWS) {
    g.fin_stats = ln.fin_stats;
    g.fin_cnt = m->fin_cnt + (size_t)m->cur_aux * FIN_ROWS;
    g.fin_inv_d = 1.0f / (float)N;
    g.fin_eps = m->cfg.ln_eps;
    fin = true;
  }
  VDR_TRY(launch_gemm_w(m, g, epi, variant, s), "gemm");
  if (ln.part) m->stats_fresh = fin;
  return VDR_OK;
}

int layernorm(vdr_model* m, hipStream_t s, int cls, const void* x, int in_bf16, void* y, int out_bf16,
              const float* gw, const float* gb, int64_t rows, RowMap imap, const float* clsrc = nullptr,
              int cls_period = 0, int width = 0) {
  LnArgs a{};
  a.x = x;
  a.in_bf16 = in_bf16;
  a.y = y;
  a.out_bf16 = out_bf16;
  a.gamma = gw;
  a.beta = gb;
  a.rows = rows;
  a.D = width ? width : m->cfg.dim;
  a.eps = m->cfg.ln_eps;
  a.imap = imap;
  a.omap = identity_map();
  a.cls = clsrc;
  a.cls_period = cls_period;
  Scope sc(m, s, cls, 0.0, (double)rows * a.D * ((in_bf16 ? 2 : 4) + (out_bf16 ? 2 : 4)));
  VDR_TRY(launch_layernorm(a, s), "layernorm");
  return VDR_OK;
}

// L transformer blocks over x [M = mb*ntok rows]; leaves the result in w.x
// tile configuration of the MX-fp8 GEMM per class and shape (VDR_MX_VARIANT overrides)
int mx_variant_for(int cls, int64_t M, int N) {
  VDR_KNOB int forced = env_int("VDR_MX_VARIANT", -1);
  if (forced >= 0) return forced;
  if (((M + 127) / 128) * ((N + 255) / 256) < 256) return 2;  // small problem: 128x128 tiles
  if (cls == VDR_K_GEMM_QKV) return 0;
  return M >= 16384 ? 0 : 1;  // measured (tools/mx_bench.py): 256x256 tiles win at ViT-g's M = 8224
}

// one linear on the block-scaled fp8 MFMA: MX operands (aq, as) x (wq, wsc); cs != NULL -> MX output
int gemm_mx(vdr_model* m, hipStream_t s, int cls, const void* aq, const void* as, const void* wq, const void* wsc,
            const float* bias, const void* resid, const float* gamma, void* C, void* cs, int64_t M, int N, int K, int ldc,
            int epi) {
  GemmArgs g{};
  g.A = aq;
  g.a_scale = as;
  g.W = wq;
  g.w_scale = wsc;
  g.bias = bias;
  g.resid = resid;
  g.gamma = gamma;
  g.C = C;
  g.c_scale = cs;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldw = K;
  g.ldc = ldc;
  g.ldr = ldc;
  g.omap = identity_map();
  const double outb = cs ? 1.0 : 2.0;
  Scope sc(m, s, cls, 2.0 * M * N * K, (double)M * K + (double)N * K + (double)M * ldc * (resid ? 2 * outb : outb));
  {
    VDR_KNOB int packed = env_int("VDR_MX_PACKED", 1);  // (tuning builds: 0 = the row-major payload, for A/B)
    auto it = m->w_il.find(g.W);
    if (packed && it != m->w_il.end()) {
      g.W = it->second;
      g.w_interleaved = 1;
    }
  }
  VDR_TRY(launch_gemm_mx(g, epi, mx_variant_for(cls, M, N), s), "gemm_mx");
  return VDR_OK;
}

// CLS-only tail of the LAST block (VDR_OUT_CLS: `model(x) -> (logits, cls)`, models_archs.py:24-29 -- the reference
// computes every token of the last block and then keeps x[:, 0]).  After the last attention nothing mixes rows any
// more: out-projection, norm2, MLP and the final norm are row-wise, so the [mb] CLS rows are all that reaches the output.
// They are gathered by the out-projection itself (A and the residual are read with a row stride of ntok * D, the
// result goes to a compact [mb, D] buffer -- the head of w.h, which no later kernel of this forward reads) and the MLP
// runs on mb rows instead of mb * ntok.  Same kernels, same per-row arithmetic: the CLS features are bitwise those of
// the full block (test_cls_rows_only_last_block_bitwise).  vdr_config.full_last_block = 1 keeps every row.
struct BookAs {  // the profiler books these launches as VDR_K_CLS_TAIL, so that gemm_proj / fc1 / fc2 stay classes of
  vdr_model* m;  // identical full-size launches (their averages are what the rocprofv3 summaries are compared with)
  explicit BookAs(vdr_model* m_) : m(m_) { m->prof_as = VDR_K_CLS_TAIL; }
  ~BookAs() { m->prof_as = -1; }
};

// vdr_config.fp8_cls_bf16: norm2 -> fc1 / w12 -> activation -> fc2 / w3 + residual of the CLS rows (row b * ntok of the
// residual stream) on the bf16 weights, enqueued on `ax`; leaves the rows' new residual values in w.cls_x [mb, D].  Reads
// the residual stream as it is after the out-projection; the caller puts w.cls_x back once the MX-fp8 MLP of every row has
// written w.x.  Same kernels and per-row arithmetic as the CLS tail of the last block (block_tail_cls, explicit-LayerNorm
// branch): a row's bits do not depend on which of the two computed it.
int cls_mlp_bf16(vdr_model* m, hipStream_t ax, const Carve& w, const LayerW& L, int mb, int ntok) {
  BookAs book(m);
  const vdr_config& c = m->cfg;
  const int D = c.dim, F = c.mlp_hidden;
  const bool sw = c.act == VDR_ACT_SWIGLU;
  const int64_t stride = (int64_t)ntok * D;
  int rc;
  if ((rc = layernorm(m, ax, VDR_K_LAYERNORM, w.x, 1, w.cls_h, 1, L.n2w, L.n2b, mb, RowMap{1, ntok, 0}))) return rc;
  if ((rc = gemm(m, ax, VDR_K_GEMM_FC1, w.cls_h, L.w1, L.b1, nullptr, nullptr, w.cls_u, mb, sw ? 2 * F : F, D, F,
                 sw ? EPI_SWIGLU : EPI_BIAS_GELU)))
    return rc;
  return gemm(m, ax, VDR_K_GEMM_FC2, w.cls_u, L.w2, L.b2, w.x, L.ls2, w.cls_x, mb, D, F, D, EPI_BIAS_RESID, LnFold(), 0, stride);
}

int block_tail_cls(vdr_model* m, hipStream_t s, const Carve& w, const LayerW& L, int mb, int ntok) {
  BookAs book(m);
  const vdr_config& c = m->cfg;
  const int D = c.dim, F = c.mlp_hidden;
  const bool sw = c.act == VDR_ACT_SWIGLU;
  const int64_t stride = (int64_t)ntok * D;
  char* xc = w.h;  // [mb, D] bf16
  int rc;
  if (c.fp8 && !(c.fp8_cls_bf16 && c.has_cls)) {  // (fp8_cls_bf16: the CLS rows' MLP on the bf16 weights -- the explicit-LayerNorm branch below)
    // MX-fp8 linears: norm2 of the compact rows goes out as MX-fp8 behind them in w.h (payload) / w.hs (scales: the
    // layouts depend only on the row count each launch is given), fc1 / fc2 on the block-scaled MFMA at M = mb
    char* hq = w.h + (size_t)round_up(mb, 256) * D * 2;
    const int N1 = sw ? 2 * F : F;
    if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, xc, mb, D, D, D, EPI_BIAS_RESID, LnFold(), stride, stride)))
      return rc;
    {
      Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)mb * D * 3);
      VDR_TRY(launch_ln_mx(xc, L.n2w, L.n2b, c.ln_eps, mb, D, hq, w.hs, s), "layernorm_mx");
    }
    if ((rc = gemm_mx(m, s, VDR_K_GEMM_FC1, hq, w.hs, L.w1_q, L.w1_s, L.b1, nullptr, nullptr, w.u, w.us, mb, N1, D, F,
                      sw ? EPI_SWIGLU : EPI_BIAS_GELU)))
      return rc;
    return gemm_mx(m, s, VDR_K_GEMM_FC2, w.u, w.us, L.w2_q, L.w2_s, L.b2, xc, L.ls2, xc, nullptr, mb, D, F, D, EPI_BIAS_RESID);
  }
  const float* r32 = w.x32;  // resid_fp32: the CLS rows' fp32 residual comes from x32 (strided) and stays in xc32 (compact)
  float* c32 = w.x32 ? w.xc32 : nullptr;
  if (m->ln_fuse) {
    LnFold prod, cons;
    prod.part = w.part;
    prod.part_stride = w.Mp;
    prod.fin_stats = w.stats;
    if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, xc, mb, D, D, D, EPI_BIAS_RESID, prod, stride, stride, r32, c32)))
      return rc;
    if ((rc = ln_consumer(m, s, VDR_K_GEMM_FC1, mb, sw ? 2 * F : F, D, w, &cons))) return rc;
    cons.colsum = L.s1;
    if ((rc = gemm(m, s, VDR_K_GEMM_FC1, xc, L.w1_f, L.t1, nullptr, nullptr, w.u, mb, sw ? 2 * F : F, D, F,
                   sw ? EPI_SWIGLU : EPI_BIAS_GELU, cons)))
      return rc;
  } else {
    char* hc = w.h + (size_t)round_up(mb, 256) * D * 2;  // norm2 of the compact rows (w.h holds Mp >= mb * ntok + 256 rows)
    if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, xc, mb, D, D, D, EPI_BIAS_RESID, LnFold(), stride, stride, r32, c32)))
      return rc;
    if ((rc = layernorm(m, s, VDR_K_LAYERNORM, c32 ? (const void*)c32 : (const void*)xc, c32 ? 0 : 1, hc, 1, L.n2w, L.n2b, mb, identity_map())))
      return rc;
    if ((rc = gemm(m, s, VDR_K_GEMM_FC1, hc, L.w1, L.b1, nullptr, nullptr, w.u, mb, sw ? 2 * F : F, D, F,
                   sw ? EPI_SWIGLU : EPI_BIAS_GELU)))
      return rc;
  }
  return gemm(m, s, VDR_K_GEMM_FC2, w.u, L.w2, L.b2, xc, L.ls2, xc, mb, D, F, D, EPI_BIAS_RESID, LnFold(), 0, 0, c32, c32);
}

// cls_tail: the caller only wants the CLS rows (see block_tail_cls); *compact is set when they were left in w.h [mb, D]
int run_blocks(vdr_model* m, hipStream_t s, const Carve& w, int mb, int ntok, const int* lens = nullptr, int len_add = 0,
               bool cls_tail = false, bool* compact = nullptr) {
  const vdr_config& c = m->cfg;
  const int D = c.dim, F = c.mlp_hidden, H = c.heads;
  const int64_t M = (int64_t)mb * ntok;
  const bool sw = c.act == VDR_ACT_SWIGLU;
  int rc;
  if (compact) *compact = false;
  // (post-LN blocks keep every row: their last operation is a LayerNorm over the block's own output, also row-wise, but
  // the classifier that uses them is not a throughput path)
  const int tail_at = (cls_tail && compact && c.pre_ln && !c.full_last_block && ntok > 1) ? c.layers - 1 : -1;
  if (c.fp8) {
    // BASELINE config 5: qkv / fc1 / fc2 on the block-scaled fp8 MFMA.  LayerNorm writes its output as MX-fp8
    // (the qkv / fc1 operand), the attention kernel and the fc1 epilogue write theirs as MX-fp8 (the proj / fc2
    // operands); the residual stream and the attention arithmetic stay bf16 / fp32.
    const int N1 = sw ? 2 * F : F;
    // vdr_config.fp8_cls_bf16 (image models with a CLS token; not the variable-length token path)
    const int ai = m->cur_aux;
    const bool cls_bf16 = c.fp8_cls_bf16 && c.has_cls && c.patch && ntok > 1 && !lens && ai < (int)m->aux.size() && w.cls_x;
    for (int i = 0; i < c.layers; ++i) {
      const LayerW& L = m->layers[i];
      {
        Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)M * D * 3);
        VDR_TRY(launch_ln_mx(w.x, L.n1w, L.n1b, c.ln_eps, M, D, w.h, w.hs, s), "layernorm_mx");
      }
      if ((rc = gemm_mx(m, s, VDR_K_GEMM_QKV, w.h, w.hs, L.qkv_q, L.qkv_s, L.bqkv, nullptr, nullptr, w.qkv, nullptr, M, 3 * D,
                        D, 3 * D, EPI_BIAS)))
        return rc;
      {
        Scope sc(m, s, VDR_K_ATTENTION, 4.0 * (double)ntok * ntok * 64.0 * H * mb, 2.0 * (double)M * 4 * D);
        VDR_KNOB int attn_variant = env_int("VDR_ATTN_VARIANT", 0);  // (tuning builds)
        VDR_TRY(launch_attention(w.qkv, w.o, mb, ntok, H, attn_variant, s, nullptr, lens, len_add), "attention");
      }
      if (i == tail_at) {
        *compact = true;
        return block_tail_cls(m, s, w, L, mb, ntok);
      }
      // the out-projection stays bf16: quantising it too measured 0.987 row cosine at 40 blocks (gate 0.99)
      if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, w.x, M, D, D, D, EPI_BIAS_RESID))) return rc;
      if (cls_bf16) {
        // fork: the CLS rows' bf16 MLP runs on the side stream under norm2 / fc1 of every row (nothing writes w.x there)
        VDR_TRY(hipEventRecord(m->aux_fork[ai], s), "hipEventRecord");
        VDR_TRY(hipStreamWaitEvent(m->aux[ai], m->aux_fork[ai], 0), "hipStreamWaitEvent");
        if ((rc = cls_mlp_bf16(m, m->aux[ai], w, L, mb, ntok))) return rc;
        VDR_TRY(hipEventRecord(m->aux_join[ai], m->aux[ai]), "hipEventRecord");
      }
      {
        Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)M * D * 3);
        VDR_TRY(launch_ln_mx(w.x, L.n2w, L.n2b, c.ln_eps, M, D, w.h, w.hs, s), "layernorm_mx");
      }
      if ((rc = gemm_mx(m, s, VDR_K_GEMM_FC1, w.h, w.hs, L.w1_q, L.w1_s, L.b1, nullptr, nullptr, w.u, w.us, M, N1, D, F,
                        sw ? EPI_SWIGLU : EPI_BIAS_GELU)))
        return rc;
      // join: fc2 rewrites every row of w.x, the CLS rows' residual reads must be over
      if (cls_bf16) VDR_TRY(hipStreamWaitEvent(s, m->aux_join[ai], 0), "hipStreamWaitEvent");
      if ((rc = gemm_mx(m, s, VDR_K_GEMM_FC2, w.u, w.us, L.w2_q, L.w2_s, L.b2, w.x, L.ls2, w.x, nullptr, M, D, F, D, EPI_BIAS_RESID)))
        return rc;
      if (cls_bf16)  // ... and the bf16 result replaces the MX-fp8 one in the CLS rows
        VDR_TRY(hipMemcpy2DAsync(w.x, (size_t)ntok * D * 2, w.cls_x, (size_t)D * 2, (size_t)D * 2, (size_t)mb, hipMemcpyDeviceToDevice, s),
                "hipMemcpy2DAsync(CLS rows)");
    }
    return VDR_OK;
  }
  if (c.pre_ln && m->ln_fuse) {
    // LayerNorm never materialised: producers leave (sum, sumsq) partials, a tiny kernel turns them into
    // (mean, rstd), the consuming GEMM applies them in its epilogue (weights pre-multiplied by gamma).
    LnFold prod;
    prod.part = w.part;
    prod.part_stride = w.Mp;
    // (the producers finalise the statistics where a consumer reads finalised ones; launches small enough for the ring3 /
    // ring4 consumers to finalise their own rows from the partials need nothing)
    if (!ln_stats_in_gemm(VDR_K_GEMM_QKV, M, 3 * D, D / 64) || !ln_stats_in_gemm(VDR_K_GEMM_FC1, M, sw ? 2 * F : F, D / 64))
      prod.fin_stats = w.stats;
    for (int i = 0; i < c.layers; ++i) {
      const LayerW& L = m->layers[i];
      LnFold cons;
      if ((rc = ln_consumer(m, s, VDR_K_GEMM_QKV, M, 3 * D, D, w, &cons))) return rc;
      cons.colsum = L.sqkv;
      if ((rc = gemm(m, s, VDR_K_GEMM_QKV, w.x, L.wqkv_f, L.tqkv, nullptr, nullptr, w.qkv, M, 3 * D, D, 3 * D, EPI_BIAS, cons, 0, 0, nullptr,
                     nullptr, w.Mp)))
        return rc;
      {
        Scope sc(m, s, VDR_K_ATTENTION, 4.0 * (double)ntok * ntok * 64.0 * H * mb, 2.0 * (double)M * 4 * D);
        VDR_KNOB int attn_variant = env_int("VDR_ATTN_VARIANT", 0);  // (tuning builds)
        VDR_TRY(launch_attention(w.qkv, w.o, mb, ntok, H, attn_variant, s, nullptr, lens, len_add), "attention");
      }
      if (i == tail_at) {
        *compact = true;
        return block_tail_cls(m, s, w, L, mb, ntok);
      }
      if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, w.x, M, D, D, D, EPI_BIAS_RESID, prod, 0, 0, w.x32, w.x32)))
        return rc;
      if ((rc = ln_consumer(m, s, VDR_K_GEMM_FC1, M, sw ? 2 * F : F, D, w, &cons))) return rc;
      cons.colsum = L.s1;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC1, w.x, L.w1_f, L.t1, nullptr, nullptr, w.u, M, sw ? 2 * F : F, D, F,
                     sw ? EPI_SWIGLU : EPI_BIAS_GELU, cons, 0, 0, nullptr, nullptr, w.Mp)))
        return rc;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC2, w.u, L.w2, L.b2, w.x, L.ls2, w.x, M, D, F, D, EPI_BIAS_RESID, prod, 0, 0, w.x32, w.x32)))
        return rc;
    }
    return VDR_OK;
  }
  for (int i = 0; i < c.layers; ++i) {
    const LayerW& L = m->layers[i];
    const void* attn_in = w.x;
    // (resid_fp32: the explicit LayerNorm reads the fp32 master copy of the stream)
    const void* xin = w.x32 ? (const void*)w.x32 : (const void*)w.x;
    const int xin_bf16 = w.x32 ? 0 : 1;
    if (c.pre_ln) {
      if ((rc = layernorm(m, s, VDR_K_LAYERNORM, xin, xin_bf16, w.h, 1, L.n1w, L.n1b, M, identity_map()))) return rc;
      attn_in = w.h;
    }
    if ((rc = gemm(m, s, VDR_K_GEMM_QKV, attn_in, L.wqkv, L.bqkv, nullptr, nullptr, w.qkv, M, 3 * D, D, 3 * D, EPI_BIAS, LnFold(), 0, 0, nullptr,
                   nullptr, w.Mp)))
      return rc;
    {
      Scope sc(m, s, VDR_K_ATTENTION, 4.0 * (double)ntok * ntok * 64.0 * H * mb, 2.0 * (double)M * 4 * D);
      VDR_KNOB int attn_variant = env_int("VDR_ATTN_VARIANT", 0);
      VDR_TRY(launch_attention(w.qkv, w.o, mb, ntok, H, attn_variant, s, nullptr, lens, len_add), "attention");
    }
    if (i == tail_at) {
      *compact = true;
      return block_tail_cls(m, s, w, L, mb, ntok);
    }
    if (c.pre_ln) {
      if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, w.x, M, D, D, D, EPI_BIAS_RESID, LnFold(), 0, 0, w.x32, w.x32)))
        return rc;
      if ((rc = layernorm(m, s, VDR_K_LAYERNORM, xin, xin_bf16, w.h, 1, L.n2w, L.n2b, M, identity_map()))) return rc;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC1, w.h, L.w1, L.b1, nullptr, nullptr, w.u, M, sw ? 2 * F : F, D, F,
                     sw ? EPI_SWIGLU : EPI_BIAS_GELU, LnFold(), 0, 0, nullptr, nullptr, w.Mp)))
        return rc;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC2, w.u, L.w2, L.b2, w.x, L.ls2, w.x, M, D, F, D, EPI_BIAS_RESID, LnFold(), 0, 0, w.x32, w.x32)))
        return rc;
    } else {
      // nn.TransformerEncoderLayer, norm_first=False: x = LN1(x + SA(x)); x = LN2(x + FF(x))
      if ((rc = gemm(m, s, VDR_K_GEMM_PROJ, w.o, L.wproj, L.bproj, w.x, L.ls1, w.h, M, D, D, D, EPI_BIAS_RESID))) return rc;
      if ((rc = layernorm(m, s, VDR_K_LAYERNORM, w.h, 1, w.x, 1, L.n1w, L.n1b, M, identity_map()))) return rc;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC1, w.x, L.w1, L.b1, nullptr, nullptr, w.u, M, sw ? 2 * F : F, D, F,
                     sw ? EPI_SWIGLU : EPI_BIAS_GELU, LnFold(), 0, 0, nullptr, nullptr, w.Mp)))
        return rc;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC2, w.u, L.w2, L.b2, w.x, L.ls2, w.h, M, D, F, D, EPI_BIAS_RESID))) return rc;
      if ((rc = layernorm(m, s, VDR_K_LAYERNORM, w.h, 1, w.x, 1, L.n2w, L.n2b, M, identity_map()))) return rc;
    }
  }
  return VDR_OK;
}

// SAM / MedSAM ImageEncoderViT blocks + neck over x [mb * g*g rows] (tokens NHWC, pos_embed already added).
// Window blocks: LN1 writes the window-partitioned, zero-padded order (padding rows of w.h stay zero),
// qkv / rel-pos / attention run on windows, the proj epilogue un-partitions while adding the residual.
// T[(token, head)][j] = q . table[j] for every relative offset j of both axes: one GEMM whose A rows are the
// per-head q slices of the packed qkv activation (M = tokens * heads, N = relpos_npad(S), K = 64), fp32 out.
hipError_t relpos_products(const void* qkv, const void* table, float* T, int64_t tokens, int S, int heads, hipStream_t s) {
  GemmArgs ga{};
  ga.A = qkv;
  ga.W = table;
  ga.C = T;
  ga.M = tokens * heads;
  ga.N = relpos_npad(S);
  ga.K = 64;
  ga.lda = 64;
  ga.ldw = 64;
  ga.ldc = ga.N;
  ga.ldr = ga.N;
  ga.omap = identity_map();
  ga.a_rpg = heads;
  ga.a_gs = (int64_t)3 * heads * 64;
  ga.a_is = 64;
  ga.out_f32 = 1;
  return launch_gemm(ga, EPI_BIAS, ga.N <= 128 ? 24 : 22, s);  // ring3: 128x128 tiles for the 64-column window table, 128x256 for the global one
}

int run_sam(vdr_model* m, hipStream_t s, const Carve& w, int mb, int out_dtype, char* out, bool tokens_only) {
  const vdr_config& c = m->cfg;
  const int D = c.dim, F = c.mlp_hidden, H = c.heads, C = c.neck_chans;
  const int g = c.img / c.patch, n = g * g, ws = c.window, nw = (g + ws - 1) / ws, wtok = nw * nw * ws * ws;
  const int64_t M = (int64_t)mb * n;
  int rc;
  VDR_TRY(hipMemsetAsync(w.h, 0, (size_t)mb * wtok * D * 2, s), "memset(window padding)");
  // fp8 (qkv / fc1 / fc2 on the block-scaled MFMA; out-projection, rel-pos GEMM, attention and neck stay bf16): the
  // padding rows of the windowed MX activation are zero payload (memset above) under zeroed, i.e. finite, scales
  const bool fp8 = c.fp8 != 0;
  if (fp8) VDR_TRY(hipMemsetAsync(w.hs, 0, mx_scale_bytes(w.Mp, D), s), "memset(window padding scales)");
  for (int i = 0; i < c.layers; ++i) {
    const LayerW& L = m->layers[i];
    const bool glob = (c.global_mask >> i) & 1;
    const int S = glob ? g : ws;
    const int64_t T = glob ? M : (int64_t)mb * wtok;
    const int nb = glob ? mb : mb * nw * nw;
    char* hbuf = glob ? w.hg : w.h;
    if (fp8) {
      {
        Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)M * D * 3);
        VDR_TRY(launch_ln_mx(w.x, L.n1w, L.n1b, c.ln_eps, M, D, hbuf, w.hs, s, glob ? 0 : ws, g, T), "layernorm_mx(window)");
      }
      if ((rc = gemm_mx(m, s, VDR_K_GEMM_QKV, hbuf, w.hs, L.qkv_q, L.qkv_s, L.bqkv, nullptr, nullptr, w.qkv, nullptr, T, 3 * D, D,
                        3 * D, EPI_BIAS)))
        return rc;
    } else {
      LnArgs a{};
      a.x = w.x;
      a.in_bf16 = 1;
      a.y = hbuf;
      a.out_bf16 = 1;
      a.gamma = L.n1w;
      a.beta = L.n1b;
      a.rows = M;
      a.D = D;
      a.eps = c.ln_eps;
      a.imap = identity_map();
      a.omap = identity_map();
      if (!glob) {
        a.win_ws = ws;
        a.win_g = g;
      }
      {
        Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)M * D * 4);  // (own block: the profiler bracket must close before the GEMM)
        VDR_TRY(launch_layernorm(a, s), "layernorm(window)");
      }
      if ((rc = gemm(m, s, VDR_K_GEMM_QKV, hbuf, L.wqkv, L.bqkv, nullptr, nullptr, w.qkv, T, 3 * D, D, 3 * D, EPI_BIAS, LnFold(), 0, 0, nullptr,
                     nullptr, w.Mp)))
        return rc;
    }
    {
      Scope sc(m, s, VDR_K_ATTENTION, 4.0 * (double)S * S * S * S * 64.0 * H * nb + 2.0 * T * H * relpos_npad(S) * 64,
               2.0 * (double)T * 4 * D);
      VDR_TRY(relpos_products(w.qkv, L.reltab, w.rel, T, S, H, s), "relpos");
      VDR_TRY(launch_attention_relpos(w.qkv, w.rel, w.o, nb, S, H, s), "attention_relpos");
    }
    {
      GemmArgs ga{};
      ga.A = w.o;
      ga.W = L.wproj;
      ga.bias = L.bproj;
      ga.resid = w.x;
      ga.C = w.x;
      ga.M = T;
      ga.N = D;
      ga.K = D;
      ga.lda = D;
      ga.ldw = D;
      ga.ldc = D;
      ga.ldr = D;
      ga.omap = identity_map();
      if (!glob) {
        ga.win_ws = ws;
        ga.win_g = g;
      }
      if (m->ln_fuse) {
        ga.ln_part = w.part;
        ga.part_stride = w.Mp;
      }
      Scope sc(m, s, VDR_K_GEMM_PROJ, 2.0 * T * D * D, 2.0 * ((double)T * D + (double)D * D + 2.0 * M * D));
      VDR_TRY(launch_gemm_w(m, ga, EPI_BIAS_RESID, gemm_variant_for(VDR_K_GEMM_PROJ, ga.M, ga.N), s), "proj gemm");
      m->stats_fresh = false;  // (window un-partition scatters the rows: their statistics are finalised by ln_consumer's launch)
    }
    if (fp8) {
      {
        Scope sc(m, s, VDR_K_LAYERNORM, 0.0, (double)M * D * 3);
        VDR_TRY(launch_ln_mx(w.x, L.n2w, L.n2b, c.ln_eps, M, D, w.hg, w.hs, s), "layernorm_mx");
      }
      if ((rc = gemm_mx(m, s, VDR_K_GEMM_FC1, w.hg, w.hs, L.w1_q, L.w1_s, L.b1, nullptr, nullptr, w.u, w.us, M, F, D, F,
                        EPI_BIAS_GELU)))
        return rc;
      if ((rc = gemm_mx(m, s, VDR_K_GEMM_FC2, w.u, w.us, L.w2_q, L.w2_s, L.b2, w.x, nullptr, w.x, nullptr, M, D, F, D,
                        EPI_BIAS_RESID)))
        return rc;
      continue;
    }
    if (m->ln_fuse) {
      LnFold cons;
      if ((rc = ln_consumer(m, s, VDR_K_GEMM_FC1, M, F, D, w, &cons))) return rc;
      cons.colsum = L.s1;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC1, w.x, L.w1_f, L.t1, nullptr, nullptr, w.u, M, F, D, F, EPI_BIAS_GELU, cons, 0, 0, nullptr, nullptr, w.Mp))) return rc;
    } else {
      if ((rc = layernorm(m, s, VDR_K_LAYERNORM, w.x, 1, w.hg, 1, L.n2w, L.n2b, M, identity_map()))) return rc;
      if ((rc = gemm(m, s, VDR_K_GEMM_FC1, w.hg, L.w1, L.b1, nullptr, nullptr, w.u, M, F, D, F, EPI_BIAS_GELU, LnFold(), 0, 0, nullptr, nullptr,
                     w.Mp)))
        return rc;
    }
    if ((rc = gemm(m, s, VDR_K_GEMM_FC2, w.u, L.w2, L.b2, w.x, nullptr, w.x, M, D, F, D, EPI_BIAS_RESID))) return rc;
  }
  if (tokens_only) {
    Scope sc(m, s, VDR_K_FINAL_LN, 0.0, (double)M * D * 6);
    VDR_TRY(launch_gather_rows(w.x, out, out_dtype == VDR_BF16, M, D, identity_map(), s), "gather_rows");
    return VDR_OK;
  }
  // neck: 1x1 conv (no bias) -> LayerNorm2d -> 3x3 conv pad 1 (no bias) -> LayerNorm2d, all on NHWC tokens
  if ((rc = gemm(m, s, VDR_K_GEMM_PATCH, w.x, m->w_neck0, nullptr, nullptr, nullptr, w.qkv, M, C, D, C, EPI_BIAS))) return rc;
  if ((rc = layernorm(m, s, VDR_K_FINAL_LN, w.qkv, 1, w.o, 1, m->neck1w, m->neck1b, M, identity_map(), nullptr, 0, C))) return rc;
  {
    Scope sc(m, s, VDR_K_IM2COL, 0.0, (double)M * C * 2 * 10);
    VDR_TRY(launch_im2col3(w.o, w.u, mb, g, C, s), "im2col3");
  }
  if ((rc = gemm(m, s, VDR_K_GEMM_PATCH, w.u, m->w_neck2, nullptr, nullptr, nullptr, w.hg, M, C, 9 * C, C, EPI_BIAS))) return rc;
  return layernorm(m, s, VDR_K_FINAL_LN, w.hg, 1, out, out_dtype == VDR_BF16, m->neck3w, m->neck3b, M, identity_map(), nullptr,
                   0, C);
}

// slice (and final-normalise) the token buffer into the caller's output
int emit(vdr_model* m, hipStream_t s, const Carve& w, int mb, int ntok, int out_mode, int out_dtype, char* out,
         bool compact = false) {  // compact: the CLS rows are in w.h [mb, D] (block_tail_cls)
  const vdr_config& c = m->cfg;
  const int D = c.dim;
  RowMap im;
  int64_t rows;
  const int ncls = c.has_cls ? 1 : 0;
  if (compact) {
    const int ob = out_dtype == VDR_BF16;
    if (w.x32) return layernorm(m, s, VDR_K_FINAL_LN, w.xc32, 0, out, ob, m->normw, m->normb, mb, identity_map());
    return layernorm(m, s, VDR_K_FINAL_LN, w.h, 1, out, ob, m->normw, m->normb, mb, identity_map());
  }
  if (out_mode == VDR_OUT_CLS) {
    im = RowMap{1, ntok, 0};
    rows = mb;
  } else if (out_mode == VDR_OUT_DENSE) {
    im = RowMap{ntok - ncls, ntok, ncls};
    rows = (int64_t)mb * (ntok - ncls);
  } else {
    im = identity_map();
    rows = (int64_t)mb * ntok;
  }
  const int ob = out_dtype == VDR_BF16;
  if (c.pre_ln) {
    if (w.x32) return layernorm(m, s, VDR_K_FINAL_LN, w.x32, 0, out, ob, m->normw, m->normb, rows, im);
    return layernorm(m, s, VDR_K_FINAL_LN, w.x, 1, out, ob, m->normw, m->normb, rows, im);
  }
  Scope sc(m, s, VDR_K_FINAL_LN, 0.0, (double)rows * D * (2 + (ob ? 2 : 4)));
  VDR_TRY(launch_gather_rows(w.x, out, ob, rows, D, im, s), "gather_rows");
  return VDR_OK;
}

size_t out_row_bytes(const vdr_model* m, int out_dtype) { return (size_t)m->cfg.dim * (out_dtype == VDR_BF16 ? 2 : 4); }

int check_device(vdr_handle h) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(h, VDR_ERR_NO_DEVICE, "no HIP device visible: libvdr has no CPU path");
  }
  return VDR_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int vdr_abi_version(void) { return VDR_ABI_VERSION; }

int vdr_tuning_build(void) {
#ifdef VDR_TUNING
  return 1;
#else
  return 0;
#endif
}

int vdr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* vdr_last_error(vdr_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

const char* vdr_kernel_class_name(int k) {
  static const char* names[VDR_K_COUNT] = {"im2col",   "gemm_patch", "layernorm", "gemm_qkv", "attention", "gemm_proj",
                                           "gemm_fc1", "gemm_fc2",   "final_ln",  "assemble", "cls_tail"};
  return (k >= 0 && k < VDR_K_COUNT) ? names[k] : "?";
}

int vdr_create(const vdr_config* cfg, int device, vdr_handle* out) {
  if (!cfg || !out) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  const vdr_config& c = *cfg;
  if (c.dim <= 0 || c.heads <= 0 || c.layers < 0 || c.mlp_hidden <= 0)
    return fail(nullptr, VDR_ERR_INVALID, "dim/heads/layers/mlp_hidden must be positive");
  if (c.dim != c.heads * 64) return fail(nullptr, VDR_ERR_UNSUPPORTED, "head dim must be 64 (dim == 64*heads)");
  if (c.dim % 64 || c.mlp_hidden % 64 || c.dim > 2048)
    return fail(nullptr, VDR_ERR_UNSUPPORTED, "dim and mlp_hidden must be multiples of 64, dim <= 2048");
  if (c.patch) {
    if (c.img <= 0 || c.img % c.patch || c.in_chans <= 0) return fail(nullptr, VDR_ERR_INVALID, "img must be a multiple of patch");
  } else if (c.has_pos) {
    return fail(nullptr, VDR_ERR_UNSUPPORTED, "token models carry no learned pos_embed");
  }
  if (c.act != VDR_ACT_GELU && c.act != VDR_ACT_SWIGLU) return fail(nullptr, VDR_ERR_INVALID, "unknown activation");
  if (c.fp8 && !c.pre_ln) return fail(nullptr, VDR_ERR_UNSUPPORTED, "fp8 weights: pre-LN models only");
  if (c.fp8_cls_bf16 < 0 || c.fp8_cls_bf16 > 1) return fail(nullptr, VDR_ERR_INVALID, "fp8_cls_bf16 must be 0 or 1");
  if (c.resid_fp32 < 0 || c.resid_fp32 > 1) return fail(nullptr, VDR_ERR_INVALID, "resid_fp32 must be 0 or 1");
  if (c.fp8 < 0 || c.fp8 > 1)
    return fail(nullptr, VDR_ERR_UNSUPPORTED,
                "fp8 must be 0 or 1 (a level that also quantised the out-projection measured 0.987 row cosine at 40 "
                "blocks, below the 0.99 gate, and is not shipped)");
  if (c.window > 0) {
    const int g = c.patch ? c.img / c.patch : 0;
    auto side_ok = [](int v) { return v == 4 || v == 7 || v == 10 || v == 14 || v == 64; };
    if (!c.patch || c.has_cls || !c.has_pos || !c.pre_ln || c.input_ln || c.layerscale || c.act != VDR_ACT_GELU)
      return fail(nullptr, VDR_ERR_INVALID, "SAM encoder: needs patch > 0, has_cls = 0, has_pos = 1, pre_ln = 1, GELU, no LayerScale");
    if (c.neck_chans <= 0 || c.neck_chans % 64 || c.neck_chans > 2048)
      return fail(nullptr, VDR_ERR_UNSUPPORTED, "SAM encoder: neck_chans must be a positive multiple of 64");
    if (!side_ok(c.window) || c.window == 64 || (c.global_mask && !side_ok(g)))
      return fail(nullptr, VDR_ERR_UNSUPPORTED, "SAM encoder: window side in {4,7,10,14}, grid side of global blocks in {4,7,10,14,64}");
    if (c.layers > 31) return fail(nullptr, VDR_ERR_UNSUPPORTED, "SAM encoder: at most 31 blocks");
  }
  int rc = check_device(nullptr);
  if (rc) return rc;
  int ndev = 0;
  hipGetDeviceCount(&ndev);
  if (device < 0 || device >= ndev) return fail(nullptr, VDR_ERR_INVALID, "device index out of range");
  std::unique_ptr<vdr_model> m(new vdr_model());
  m->cfg = c;
  m->device = device;
  if (c.patch) {
    const int g = c.img / c.patch;
    m->n_patches = g * g;
    m->n_tokens = m->n_patches + (c.has_cls ? 1 : 0);
    m->Kp = round_up(c.in_chans * c.patch * c.patch, 64);
  }
  build_slots(m.get());
  *out = m.release();
  return VDR_OK;
}

void vdr_destroy(vdr_handle h) {
  if (!h) return;
  DeviceGuard dg(h->device);
  for (auto& s : h->slots)
    if (s.dev) hipFree(s.dev);
  for (auto& L : h->layers) {
    if (L.wqkv_f) hipFree(L.wqkv_f);
    if (L.reltab) hipFree(L.reltab);
    for (void* q : {L.qkv_q, L.qkv_s, L.w1_q, L.w1_s, L.w2_q, L.w2_s})
      if (q) hipFree(q);
    if (L.w1_f) hipFree(L.w1_f);
    if (L.sqkv) hipFree(L.sqkv);
    if (L.tqkv) hipFree(L.tqkv);
    if (L.s1) hipFree(L.s1);
    if (L.t1) hipFree(L.t1);
  }
  for (auto& kv : h->w_il)
    if (kv.second) hipFree(kv.second);
  if (h->fin_cnt) hipFree(h->fin_cnt);
  for (auto st : h->streams) hipStreamDestroy(st);
  for (auto st : h->aux) hipStreamDestroy(st);
  for (auto e : h->aux_fork) hipEventDestroy(e);
  for (auto e : h->aux_join) hipEventDestroy(e);
  for (auto e : h->ev_join) hipEventDestroy(e);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  for (auto& e : h->ev_used) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  for (auto& e : h->ev_free) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  delete h;
}

int vdr_num_weights(vdr_handle h) { return h ? (int)h->slots.size() : 0; }

const char* vdr_weight_name(vdr_handle h, int i) {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
}

int vdr_set_weight(vdr_handle m, const char* name, const float* host, const int64_t* shape, int ndim) {
  if (!m || !name || !host || (ndim > 0 && !shape)) return fail(m, VDR_ERR_INVALID, "null argument");
  auto it = m->index.find(name);
  if (it == m->index.end()) return fail(m, VDR_ERR_UNKNOWN_NAME, std::string("unknown weight name: ") + name);
  WSlot& s = m->slots[it->second];
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= shape[i];
  if (numel != s.numel)
    return fail(m, VDR_ERR_INVALID, std::string(name) + ": expected " + std::to_string(s.numel) + " elements, got " +
                                        std::to_string(numel));
  DeviceGuard dg(m->device);
  if (!dg.ok) return fail(m, VDR_ERR_HIP, "hipSetDevice failed");
  std::vector<uint16_t> bf;
  std::vector<float> fv;
  const void* src = host;
  size_t bytes = 0;
  const int F = m->cfg.mlp_hidden;
  switch (s.kind) {
    case W_VEC_F32:
      bytes = (size_t)numel * 4;
      break;
    case W_MAT_BF16:
      bf.resize(numel);
      for (int64_t i = 0; i < numel; ++i) bf[i] = f32_to_bf16(host[i]);
      src = bf.data();
      bytes = (size_t)numel * 2;
      break;
    case W_PATCH_BF16: {
      // [D, C*p*p] -> [D, Kp] zero padded along k
      const int64_t K = s.cols, Kp = m->Kp;
      bf.assign((size_t)s.rows * Kp, 0);
      for (int64_t r = 0; r < s.rows; ++r)
        for (int64_t k = 0; k < K; ++k) bf[r * Kp + k] = f32_to_bf16(host[r * K + k]);
      src = bf.data();
      bytes = bf.size() * 2;
      break;
    }
    case W_W12_BF16: {
      // SwiGLU: interleave x1/x2 rows in blocks of 32 so one wave tile holds a gate pair:
      // packed row 64*blk + t = (t < 32 ? x1[32*blk + t] : x2[32*blk + t - 32])
      if (F % 32) return fail(m, VDR_ERR_UNSUPPORTED, "SwiGLU hidden must be a multiple of 32");
      const int64_t K = s.cols;
      bf.resize(numel);
      for (int64_t pr = 0; pr < 2 * F; ++pr) {
        const int64_t blk = pr / 64, t = pr % 64;
        const int64_t srow = t < 32 ? blk * 32 + t : F + blk * 32 + (t - 32);
        for (int64_t k = 0; k < K; ++k) bf[pr * K + k] = f32_to_bf16(host[srow * K + k]);
      }
      src = bf.data();
      bytes = (size_t)numel * 2;
      break;
    }
    case W_CONV3_BF16: {
      // [C_out, C_in, 3, 3] -> [C_out][j * C_in + c], j = ky*3 + kx (tap-major so that im2col3 moves
      // 8 channels of one tap with a single 16-byte load)
      const int64_t Cin = s.cols / 9;
      bf.resize(numel);
      for (int64_t co = 0; co < s.rows; ++co)
        for (int64_t ci = 0; ci < Cin; ++ci)
          for (int64_t j = 0; j < 9; ++j) bf[co * s.cols + j * Cin + ci] = f32_to_bf16(host[(co * Cin + ci) * 9 + j]);
      src = bf.data();
      bytes = (size_t)numel * 2;
      break;
    }
    case W_W12_BIAS: {
      fv.resize(numel);
      for (int64_t pr = 0; pr < 2 * F; ++pr) {
        const int64_t blk = pr / 64, t = pr % 64;
        fv[pr] = host[t < 32 ? blk * 32 + t : F + blk * 32 + (t - 32)];
      }
      src = fv.data();
      bytes = (size_t)numel * 4;
      break;
    }
  }
  s.host.assign(host, host + numel);
  if (!s.dev) VDR_TRY(hipMalloc(&s.dev, bytes + 256), "hipMalloc(weight)");
  VDR_TRY(hipMemcpy(s.dev, src, bytes, hipMemcpyHostToDevice), "hipMemcpy(weight)");
  s.set = true;
  m->resolved = false;
  return VDR_OK;
}

int vdr_finalize(vdr_handle m) {
  if (!m) return fail(m, VDR_ERR_INVALID, "null handle");
  int rc = check_device(m);
  if (rc) return rc;
  DeviceGuard dg(m->device);
  if (!dg.ok) return fail(m, VDR_ERR_HIP, "hipSetDevice failed");
  if (m->resolved) return VDR_OK;
  return resolve(m);
}

int vdr_workspace_bytes(vdr_handle m, int batch, int seq, size_t* out) {
  if (!m || !out || batch <= 0) return fail(m, VDR_ERR_INVALID, "bad argument");
  const int ntok = m->cfg.patch ? m->n_tokens : seq + (m->cfg.has_cls ? 1 : 0);
  if (ntok <= 0) return fail(m, VDR_ERR_INVALID, "seq must be positive for a token model");
  const int mb = default_micro_batch(m, batch);
  *out = carve(m, nullptr, mb, ntok).total * num_streams(m);
  return VDR_OK;
}

int vdr_forward(vdr_handle m, const void* images, int in_dtype, int batch, void* out, int out_mode, int out_dtype,
                void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !images || !out || !workspace || batch <= 0) return fail(m, VDR_ERR_INVALID, "null/invalid argument");
  const vdr_config& c = m->cfg;
  if (!c.patch) return fail(m, VDR_ERR_INVALID, "vdr_forward needs an image model (patch > 0)");
  if (in_dtype != VDR_F32 && in_dtype != VDR_BF16) return fail(m, VDR_ERR_INVALID, "in_dtype");
  if (out_dtype != VDR_F32 && out_dtype != VDR_BF16) return fail(m, VDR_ERR_INVALID, "out_dtype");
  if (out_mode < VDR_OUT_CLS || out_mode > VDR_OUT_ENCODER) return fail(m, VDR_ERR_INVALID, "out_mode");
  if (out_mode == VDR_OUT_CLS && !c.has_cls) return fail(m, VDR_ERR_INVALID, "model has no cls token");
  if ((out_mode == VDR_OUT_ENCODER) != (c.window > 0 && out_mode != VDR_OUT_PATCH_EMBED && out_mode != VDR_OUT_TOKENS))
    return fail(m, VDR_ERR_INVALID, "VDR_OUT_ENCODER is the output of a SAM encoder (window > 0); other models use CLS/DENSE/TOKENS");
  int rc = check_device(m);
  if (rc) return rc;
  if (!m->resolved) return fail(m, VDR_ERR_INCOMPLETE, "vdr_finalize has not run since the last vdr_set_weight");
  DeviceGuard dg(m->device);
  if (!dg.ok) return fail(m, VDR_ERR_HIP, "hipSetDevice failed");
  const int mb_max = default_micro_batch(m, batch);
  const int ntok = m->n_tokens, n = m->n_patches, D = c.dim;
  const int ns = num_streams(m);
  const size_t per_ws = carve(m, nullptr, mb_max, ntok).total;
  if (per_ws * ns > workspace_bytes)
    return fail(m, VDR_ERR_WORKSPACE, "workspace too small: need " + std::to_string(per_ws * ns) + " bytes");
  hipStream_t caller = (hipStream_t)stream;
  if (fork_streams(m, caller)) return fail(m, VDR_ERR_HIP, "internal stream setup failed");
  const size_t img_elems = (size_t)c.in_chans * c.img * c.img;
  const size_t in_es = in_dtype == VDR_BF16 ? 2 : 4;
  const int ncls = c.has_cls ? 1 : 0;
  int chunk = 0;
  for (int b0 = 0; b0 < batch; b0 += mb_max, ++chunk) {
    const int mb = batch - b0 < mb_max ? batch - b0 : mb_max;
    const int si = chunk % ns;
    m->cur_aux = si;
    m->stats_fresh = false;  // (nothing has finalised the statistics of this micro-batch yet)
    hipStream_t s = ns == 1 ? caller : m->streams[si];
    const Carve w = carve(m, (char*)workspace + si * per_ws, mb_max, ntok);
    const char* img = (const char*)images + (size_t)b0 * img_elems * in_es;
    // bf16 images with a patch side of 8 / 16 / 32: the patch GEMM's operand loader gathers 16-byte runs of pixels
    // straight from the NCHW images (ring4 tile variants) -- no col buffer, no im2col launch.  fp32 images (the loader
    // is an LDS-DMA: it cannot convert) and p = 14 (runs of 14 pixels are not 16-byte chunks) go through im2col.
    const int pvar = gemm_variant_for(VDR_K_GEMM_PATCH, (int64_t)mb * n, D);
    const bool fused_patch = patch_gather_ok(in_dtype, c.patch, pvar, img);
    const bool pe_only = out_mode == VDR_OUT_PATCH_EMBED;
    if (!fused_patch) {
      Scope sc(m, s, VDR_K_IM2COL, 0.0, (double)mb * img_elems * in_es + 2.0 * mb * n * m->Kp);
      VDR_TRY(launch_im2col(img, in_dtype == VDR_BF16, w.u, mb, c.in_chans, c.img, c.patch, m->Kp, s), "im2col");
    }
    {
      GemmArgs g{};
      g.A = fused_patch ? (const void*)img : (const void*)w.u;
      if (fused_patch) {
        g.patch_p = c.patch;
        g.patch_g = c.img / c.patch;
        g.patch_C = c.in_chans;
      }
      g.W = m->w_patch;
      g.bias = m->b_patch;
      g.pos = pe_only ? nullptr : m->pos;
      g.M = (int64_t)mb * n;
      g.N = D;
      g.K = m->Kp;
      g.lda = m->Kp;
      g.ldw = m->Kp;
      g.ldc = D;
      g.ldr = D;
      if (m->ln_fuse && !pe_only) {
        g.ln_part = w.part;
        g.part_stride = w.Mp;
      }
      if (pe_only) {
        // model.patch_embed(x) (tfds_dense_descriptor.py:128): the GEMM epilogue writes the caller's [B, n, D] buffer
        // directly, bf16 or fp32 (no conversion pass)
        g.C = (char*)out + (size_t)b0 * n * D * (out_dtype == VDR_BF16 ? 2 : 4);
        g.out_f32 = out_dtype != VDR_BF16;
        g.omap = RowMap{n, n, 0};
      } else {
        g.C = w.x;
        g.omap = RowMap{n, ntok, ncls};
      }
      Scope sc(m, s, VDR_K_GEMM_PATCH, 2.0 * g.M * D * c.in_chans * c.patch * c.patch,
               2.0 * ((double)g.M * m->Kp + (double)D * m->Kp + (double)g.M * D));
      VDR_TRY(launch_gemm_w(m, g, EPI_PATCH, pvar, s), "patch gemm");
    }
    if (pe_only) continue;
    if (c.window > 0) {
      const bool tok = out_mode == VDR_OUT_TOKENS;
      const size_t orow_b = tok ? out_row_bytes(m, out_dtype) : (size_t)c.neck_chans * (out_dtype == VDR_BF16 ? 2 : 4);
      if ((rc = run_sam(m, s, w, mb, out_dtype, (char*)out + (size_t)b0 * n * orow_b, tok))) return rc;
      continue;
    }
    if (c.has_cls) {
      Scope sc(m, s, VDR_K_ASSEMBLE, 0.0, (double)mb * D * 2);
      if (m->ln_fuse)
        VDR_TRY(launch_cls_rows_stats(m->cls, m->pos, w.x, w.part, w.Mp, mb, ntok, D, s), "cls rows");
      else
        VDR_TRY(launch_cls_rows(m->cls, m->pos, w.x, mb, ntok, D, s), "cls rows");
    }
    if (c.input_ln) {
      if ((rc = layernorm(m, s, VDR_K_LAYERNORM, w.x, 1, w.x, 1, m->inw, m->inb, (int64_t)mb * ntok, identity_map())))
        return rc;
    }
    if (w.x32) {  // resid_fp32: the stream's fp32 master copy starts from the assembled tokens (their one bf16 rounding stays)
      Scope sc(m, s, VDR_K_ASSEMBLE, 0.0, (double)mb * ntok * D * 6);
      VDR_TRY(launch_gather_rows(w.x, w.x32, 0, (int64_t)mb * ntok, D, identity_map(), s), "residual stream -> fp32");
    }
    bool compact = false;
    if ((rc = run_blocks(m, s, w, mb, ntok, nullptr, 0, out_mode == VDR_OUT_CLS, &compact))) return rc;
    const int64_t rows_per_img = out_mode == VDR_OUT_CLS ? 1 : (out_mode == VDR_OUT_DENSE ? ntok - ncls : ntok);
    char* o = (char*)out + (size_t)b0 * rows_per_img * out_row_bytes(m, out_dtype);
    if ((rc = emit(m, s, w, mb, ntok, out_mode, out_dtype, o, compact))) return rc;
  }
  if (join_streams(m, caller)) return fail(m, VDR_ERR_HIP, "internal stream join failed");
  return VDR_OK;
}

static int forward_tokens_impl(vdr_handle m, const void* tokens, int in_dtype, int batch, int seq, const int32_t* seq_lens,
                               void* out, int out_mode, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

int vdr_forward_tokens(vdr_handle m, const void* tokens, int in_dtype, int batch, int seq, void* out, int out_mode,
                       int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_tokens_impl(m, tokens, in_dtype, batch, seq, nullptr, out, out_mode, out_dtype, workspace, workspace_bytes, stream);
}

int vdr_forward_tokens_varlen(vdr_handle m, const void* tokens, int in_dtype, int batch, int max_seq, const int32_t* seq_lens,
                              void* out, int out_mode, int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!seq_lens) return fail(m, VDR_ERR_INVALID, "seq_lens is null");
  return forward_tokens_impl(m, tokens, in_dtype, batch, max_seq, seq_lens, out, out_mode, out_dtype, workspace, workspace_bytes,
                             stream);
}

static int forward_tokens_impl(vdr_handle m, const void* tokens, int in_dtype, int batch, int seq, const int32_t* seq_lens,
                               void* out, int out_mode, int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !tokens || !out || !workspace || batch <= 0 || seq <= 0) return fail(m, VDR_ERR_INVALID, "null/invalid argument");
  const vdr_config& c = m->cfg;
  if (c.patch) return fail(m, VDR_ERR_INVALID, "vdr_forward_tokens needs a token model (patch == 0)");
  if (in_dtype != VDR_F32 && in_dtype != VDR_BF16) return fail(m, VDR_ERR_INVALID, "in_dtype");
  if (out_dtype != VDR_F32 && out_dtype != VDR_BF16) return fail(m, VDR_ERR_INVALID, "out_dtype");
  if (out_mode != VDR_OUT_CLS && out_mode != VDR_OUT_TOKENS && out_mode != VDR_OUT_DENSE)
    return fail(m, VDR_ERR_INVALID, "out_mode");
  if (out_mode == VDR_OUT_CLS && !c.has_cls) return fail(m, VDR_ERR_INVALID, "model has no cls token");
  int rc = check_device(m);
  if (rc) return rc;
  if (!m->resolved) return fail(m, VDR_ERR_INCOMPLETE, "vdr_finalize has not run since the last vdr_set_weight");
  DeviceGuard dg(m->device);
  if (!dg.ok) return fail(m, VDR_ERR_HIP, "hipSetDevice failed");
  const int ncls = c.has_cls ? 1 : 0;
  const int ntok = seq + ncls, D = c.dim;
  const int mb_max = default_micro_batch(m, batch);
  const int ns = num_streams(m);
  const size_t per_ws = carve(m, nullptr, mb_max, ntok).total;
  if (per_ws * ns > workspace_bytes)
    return fail(m, VDR_ERR_WORKSPACE, "workspace too small: need " + std::to_string(per_ws * ns) + " bytes");
  hipStream_t caller = (hipStream_t)stream;
  if (fork_streams(m, caller)) return fail(m, VDR_ERR_HIP, "internal stream setup failed");
  const size_t in_es = in_dtype == VDR_BF16 ? 2 : 4;
  int chunk = 0;
  for (int b0 = 0; b0 < batch; b0 += mb_max, ++chunk) {
    const int mb = batch - b0 < mb_max ? batch - b0 : mb_max;
    const int si = chunk % ns;
    m->cur_aux = si;
    m->stats_fresh = false;  // (nothing has finalised the statistics of this micro-batch yet)
    hipStream_t s = ns == 1 ? caller : m->streams[si];
    const Carve w = carve(m, (char*)workspace + si * per_ws, mb_max, ntok);
    const char* tok = (const char*)tokens + (size_t)b0 * seq * D * in_es;
    const int64_t M = (int64_t)mb * ntok;
    if (c.input_ln) {
      // LayerNorm([cls ; tokens]) straight from the caller's buffer (models_archs.py:143-145)
      const RowMap im = c.has_cls ? RowMap{ntok, seq, -1} : identity_map();
      if ((rc = layernorm(m, s, VDR_K_ASSEMBLE, tok, in_dtype == VDR_BF16, w.x, 1, m->inw, m->inb, M, im,
                          c.has_cls ? m->cls : nullptr, ntok)))
        return rc;
    } else {
      Scope sc(m, s, VDR_K_ASSEMBLE, 0.0, (double)M * D * (in_es + 2));
      VDR_TRY(launch_assemble_tokens(tok, in_dtype == VDR_BF16, m->cls, nullptr, w.x, mb, seq, D, ncls, s), "assemble");
    }
    if ((rc = run_blocks(m, s, w, mb, ntok, seq_lens ? seq_lens + b0 : nullptr, ncls))) return rc;
    const int64_t rows_per = out_mode == VDR_OUT_CLS ? 1 : (out_mode == VDR_OUT_DENSE ? seq : ntok);
    char* o = (char*)out + (size_t)b0 * rows_per * out_row_bytes(m, out_dtype);
    if ((rc = emit(m, s, w, mb, ntok, out_mode, out_dtype, o))) return rc;
  }
  if (join_streams(m, caller)) return fail(m, VDR_ERR_HIP, "internal stream join failed");
  return VDR_OK;
}

// ---- single operators -----------------------------------------------------------------------------
#define OP_TRY(expr, what)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) return fail(nullptr, VDR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(_e)); \
  } while (0)

int vdr_op_layernorm(const void* x, int in_dtype, void* y, int out_dtype, const float* gamma, const float* beta,
                     int64_t rows, int D, float eps, void* stream) {
  if (!x || !y || !gamma || !beta) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  int rc = check_device(nullptr);
  if (rc) return rc;
  LnArgs a{};
  a.x = x;
  a.in_bf16 = in_dtype == VDR_BF16;
  a.y = y;
  a.out_bf16 = out_dtype == VDR_BF16;
  a.gamma = gamma;
  a.beta = beta;
  a.rows = rows;
  a.D = D;
  a.eps = eps;
  a.imap = identity_map();
  a.omap = identity_map();
  OP_TRY(launch_layernorm(a, (hipStream_t)stream), "layernorm");
  return VDR_OK;
}

static int op_linear_impl(const void* x, const void* W, int packed, const float* bias, const void* resid, const float* gamma,
                          void* y, int64_t M, int N, int K, int epilogue, int variant, void* stream) {
  if (!x || !W || !y) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (epilogue < VDR_EPI_BIAS || epilogue > VDR_EPI_SWIGLU) return fail(nullptr, VDR_ERR_INVALID, "epilogue");
  if (epilogue == VDR_EPI_BIAS_RESID && !resid) return fail(nullptr, VDR_ERR_INVALID, "resid required");
  if (K % 64 || N % 8) return fail(nullptr, VDR_ERR_UNSUPPORTED, "K % 64 == 0 and N % 8 == 0 required");
#ifndef VDR_TUNING
  if (variant < 0 || variant >= 100) return fail(nullptr, VDR_ERR_INVALID, "variant");  // (ablation encodings: tuning builds only)
#endif
  int rc = check_device(nullptr);
  if (rc) return rc;
  GemmArgs g{};
  g.A = x;
  g.W = W;
  g.w_interleaved = packed;
  g.bias = bias;
  g.resid = resid;
  g.gamma = gamma;
  g.C = y;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldw = K;
  g.ldc = epilogue == VDR_EPI_SWIGLU ? N / 2 : N;
  g.ldr = g.ldc;
  g.omap = identity_map();
  if (variant == 0)  // library default: what the forward itself would pick for this shape
    variant = gemm_variant_for(N >= 2304 ? VDR_K_GEMM_QKV : VDR_K_GEMM_FC1, M, N);
  const hipError_t e = launch_gemm(g, epilogue, variant, (hipStream_t)stream);
  if (e == hipErrorInvalidValue) {
    (void)hipGetLastError();
    return fail(nullptr, VDR_ERR_INVALID, "gemm: unknown tile variant or unsupported shape");
  }
  OP_TRY(e, "gemm");
  return VDR_OK;
}

int vdr_op_linear(const void* x, const void* W, const float* bias, const void* resid, const float* gamma, void* y,
                  int64_t M, int N, int K, int epilogue, int variant, void* stream) {
  return op_linear_impl(x, W, 0, bias, resid, gamma, y, M, N, K, epilogue, variant, stream);
}

int vdr_op_pack_linear_weight(const void* W, int N, int K, void* packed, void* stream) {
  if (!W || !packed) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (N <= 0 || (N & 1) || K <= 0 || (K & 31)) return fail(nullptr, VDR_ERR_UNSUPPORTED, "N even and K % 32 == 0 required");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_w_interleave(W, packed, N, K, K, (hipStream_t)stream), "w_interleave");
  return VDR_OK;
}

int vdr_op_linear_packed(const void* x, const void* Wp, const float* bias, const void* resid, const float* gamma, void* y,
                         int64_t M, int N, int K, int epilogue, int variant, void* stream) {
  return op_linear_impl(x, Wp, 1, bias, resid, gamma, y, M, N, K, epilogue, variant, stream);
}

size_t vdr_prepare_scratch_bytes(int batch, int h, int w, int channels, int out_side) {
  if (batch <= 0 || h <= 0 || w <= 0 || channels <= 0 || out_side <= 0) return 0;
  return prepare_scratch_bytes(batch, h, w, channels, out_side);
}

int vdr_op_prepare_image(const void* src, int src_dtype, int batch, int h, int w, int channels, int64_t stride_b,
                         int64_t stride_y, int64_t stride_x, int64_t stride_c, int flip, int out_side, void* out,
                         int out_dtype, void* scratch, void* stream) {
  if (!src || !out) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (src_dtype != VDR_F32 && src_dtype != VDR_F64) return fail(nullptr, VDR_ERR_UNSUPPORTED, "prepare_image: fp32 or fp64 input");
  if (out_dtype != VDR_F32 && out_dtype != VDR_BF16) return fail(nullptr, VDR_ERR_UNSUPPORTED, "prepare_image: fp32 or bf16 output");
  if (prepare_scratch_bytes(batch, h, w, channels, out_side) && !scratch)
    return fail(nullptr, VDR_ERR_INVALID, "prepare_image: down-scaling needs the scratch buffer");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_prepare(src, src_dtype == VDR_F32, batch, h, w, channels, stride_b, stride_y, stride_x, stride_c, flip, out_side,
                        out, out_dtype == VDR_BF16, (float*)scratch, (hipStream_t)stream),
         "prepare_image");
  return VDR_OK;
}

int vdr_op_window_ct(const void* ct, int in_dtype, int64_t n, double width, double level, float* out, void* stream) {
  if (!ct || !out) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (in_dtype != VDR_F32 && in_dtype != VDR_I16) return fail(nullptr, VDR_ERR_UNSUPPORTED, "window_ct: fp32 or int16 input");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_window_ct(ct, in_dtype == VDR_I16, n, width, level, out, (hipStream_t)stream), "window_ct");
  return VDR_OK;
}

int vdr_op_hu_to_rgb(const void* hu, int in_dtype, int64_t n, void* rgb, void* stream) {
  if (!hu || !rgb) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  const int dt = in_dtype == VDR_F32 ? 0 : in_dtype == VDR_I16 ? 1 : in_dtype == VDR_F64 ? 2 : -1;
  if (dt < 0) return fail(nullptr, VDR_ERR_UNSUPPORTED, "hu_to_rgb: fp32, int16 or fp64 input");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_hu_to_rgb(hu, dt, n, rgb, (hipStream_t)stream), "hu_to_rgb");
  return VDR_OK;
}

int vdr_op_crop_hwc(const float* src, float* dst, int batch, int H, int W, int C, int y0, int x0, int crop_h, int crop_w,
                    void* stream) {
  if (!src || !dst) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_crop_hwc(src, dst, batch, H, W, C, y0, x0, crop_h, crop_w, (hipStream_t)stream), "crop_hwc");
  return VDR_OK;
}

int vdr_op_voxel_sequence(const float* feat, const int64_t* index, const double* xyz, const double* expo, int64_t n, int D,
                          void* out, int out_dtype, void* stream) {
  if (n < 0 || D < 6) return fail(nullptr, VDR_ERR_INVALID, "voxel_sequence: n >= 0, D >= 6");
  if (n == 0) return VDR_OK;
  if (!feat || !index || !xyz || !expo || !out) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  const int dt = out_dtype == VDR_F32 ? 0 : out_dtype == VDR_BF16 ? 1 : out_dtype == VDR_F64 ? 2 : -1;
  if (dt < 0) return fail(nullptr, VDR_ERR_UNSUPPORTED, "voxel_sequence: fp32, bf16 or fp64 output");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_voxel_sequence(feat, index, xyz, expo, n, D, out, dt, (hipStream_t)stream), "voxel_sequence");
  return VDR_OK;
}

size_t vdr_affine_cubic_scratch_bytes(int h, int w, int64_t planes) {
  if (h <= 0 || w <= 0 || planes <= 0) return 0;
  return affine_cubic_scratch_bytes(h, w, planes);
}

int vdr_op_affine_cubic(const void* src, int dtype, int h, int w, int64_t planes, const double* matrix, const double* offset,
                        void* out, int clip01, void* scratch, void* stream) {
  if (!src || !out || !matrix || !offset || !scratch) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (h < 2 || w < 2 || planes <= 0) return fail(nullptr, VDR_ERR_INVALID, "affine_cubic: planes of at least 2 x 2");
  const int dt = dtype == VDR_F64 ? 0 : dtype == VDR_F32 ? 1 : dtype == VDR_U8 ? 2 : -1;
  if (dt < 0) return fail(nullptr, VDR_ERR_UNSUPPORTED, "affine_cubic: fp64, fp32 or uint8 (boolean mask) volume");
  if ((int64_t)(h + 24) * (w + 24) * planes >= ((int64_t)1 << 31) * 256)
    return fail(nullptr, VDR_ERR_UNSUPPORTED, "affine_cubic: volume too large for one launch");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_affine_cubic(src, dt, h, w, planes, matrix, offset, out, clip01, (double*)scratch, (hipStream_t)stream),
         "affine_cubic");
  return VDR_OK;
}

size_t vdr_mx_scale_bytes(int64_t rows, int K) { return rows > 0 && K > 0 ? mx_scale_bytes(rows, K) : 0; }

int vdr_op_mx_quantize(const void* x, int64_t rows, int K, void* q, void* scales, void* stream) {
  if (!x || !q || !scales) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_mx_quant(x, rows, K, K, q, scales, (hipStream_t)stream), "mx_quantize");
  return VDR_OK;
}

int vdr_op_mx_dequantize(const void* q, const void* scales, int64_t rows, int K, float* y, void* stream) {
  if (!q || !scales || !y) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_mx_dequant(q, scales, rows, K, y, (hipStream_t)stream), "mx_dequantize");
  return VDR_OK;
}

int vdr_op_layernorm_mx(const void* x, const float* gamma, const float* beta, float eps, int64_t rows, int D, void* q,
                        void* scales, void* stream) {
  if (!x || !gamma || !beta || !q || !scales) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_ln_mx(x, gamma, beta, eps, rows, D, q, scales, (hipStream_t)stream), "layernorm_mx");
  return VDR_OK;
}

int vdr_op_linear_mx(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias, const void* resid,
                     const float* gamma, void* y, void* yscales, int64_t M, int N, int K, int epilogue, int variant,
                     void* stream) {
  if (!xq || !xs || !wq || !ws || !y) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (epilogue == VDR_EPI_BIAS_RESID && !resid) return fail(nullptr, VDR_ERR_INVALID, "EPI_BIAS_RESID needs resid");
  int rc = check_device(nullptr);
  if (rc) return rc;
  GemmArgs g{};
  g.A = xq;
  g.a_scale = xs;
  g.W = wq;
  g.w_scale = ws;
  g.bias = bias;
  g.resid = resid;
  g.gamma = gamma;
  g.C = y;
  g.c_scale = yscales;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldw = K;
  g.ldc = epilogue == VDR_EPI_SWIGLU ? N / 2 : N;
  g.ldr = g.ldc;
  g.omap = identity_map();
  OP_TRY(launch_gemm_mx(g, epilogue, variant, (hipStream_t)stream), "gemm_mx");
  return VDR_OK;
}

int vdr_op_attention(const void* qkv, void* out, int batch, int seq, int heads, int variant, void* stream) {
  if (!qkv || !out) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  int rc = check_device(nullptr);
  if (rc) return rc;
  OP_TRY(launch_attention(qkv, out, batch, seq, heads, variant, (hipStream_t)stream), "attention");
  return VDR_OK;
}

int vdr_op_attention_relpos(const void* qkv, const float* rel_pos_h, const float* rel_pos_w, float* rel, void* out,
                            int batch, int S, int heads, void* stream) {
  if (!qkv || !rel_pos_h || !rel_pos_w || !rel || !out) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (batch <= 0 || S <= 0 || heads <= 0) return fail(nullptr, VDR_ERR_INVALID, "bad shape");
  int rc = check_device(nullptr);
  if (rc) return rc;
  const int64_t tokens = (int64_t)batch * S * S;
  const int npad = relpos_npad(S);
  void* table = rel + tokens * heads * npad;  // packed bf16 tables behind the products
  OP_TRY(launch_relpos_pack(rel_pos_h, rel_pos_w, table, S, (hipStream_t)stream), "relpos_pack");
  OP_TRY(relpos_products(qkv, table, rel, tokens, S, heads, (hipStream_t)stream), "relpos");
  OP_TRY(launch_attention_relpos(qkv, rel, out, batch, S, heads, (hipStream_t)stream), "attention_relpos");
  return VDR_OK;
}

int vdr_op_patch_embed(const void* images, int in_dtype, const void* W, const float* bias, const float* pos, void* col,
                       void* y, int batch, int C, int img, int p, int D, int row_stride, int row_offset, void* stream) {
  if (!images || !W || !col || !y) return fail(nullptr, VDR_ERR_INVALID, "null argument");
  if (p <= 0 || img % p) return fail(nullptr, VDR_ERR_INVALID, "img must be a multiple of p");
  int rc = check_device(nullptr);
  if (rc) return rc;
  const int g = img / p, n = g * g, Kp = round_up(C * p * p, 64);
  const int variant = gemm_variant_for(VDR_K_GEMM_PATCH, (int64_t)batch * n, D);
  const bool fused = patch_gather_ok(in_dtype, p, variant, images);  // (see vdr_forward: no im2col pass, `col` untouched)
  if (!fused) OP_TRY(launch_im2col(images, in_dtype == VDR_BF16, col, batch, C, img, p, Kp, (hipStream_t)stream), "im2col");
  GemmArgs a{};
  a.A = fused ? images : col;
  if (fused) {
    a.patch_p = p;
    a.patch_g = g;
    a.patch_C = C;
  }
  a.W = W;
  a.bias = bias;
  a.pos = pos;
  a.C = y;
  a.M = (int64_t)batch * n;
  a.N = D;
  a.K = Kp;
  a.lda = Kp;
  a.ldw = Kp;
  a.ldc = D;
  a.ldr = D;
  a.omap = RowMap{n, row_stride, row_offset};
  OP_TRY(launch_gemm(a, EPI_PATCH, variant, (hipStream_t)stream), "patch gemm");
  return VDR_OK;
}

// ---- profiler ---------------------------------------------------------------------------------------
int vdr_profile_enable(vdr_handle m, int on) {
  if (!m) return fail(m, VDR_ERR_INVALID, "null handle");
  m->prof = on != 0;
  return VDR_OK;
}

int vdr_profile_mask(vdr_handle m, uint32_t class_mask) {
  if (!m) return fail(m, VDR_ERR_INVALID, "null handle");
  m->prof_mask = class_mask;
  return VDR_OK;
}

int vdr_profile_read(vdr_handle m, double* ms, int64_t* launches, double* flops, double* bytes, int n) {
  if (!m || !ms || n < VDR_K_COUNT) return fail(m, VDR_ERR_INVALID, "bad argument");
  for (int k = 0; k < VDR_K_COUNT; ++k) ms[k] = 0.0;
  for (auto& e : m->ev_used) {
    VDR_TRY(hipEventSynchronize(e.b), "hipEventSynchronize");
    float t = 0.0f;
    VDR_TRY(hipEventElapsedTime(&t, e.a, e.b), "hipEventElapsedTime");
    ms[e.cls] += t;
    m->ev_free.push_back(e);
  }
  m->ev_used.clear();
  for (int k = 0; k < VDR_K_COUNT; ++k) {
    if (launches) launches[k] = m->p_launch[k];
    if (flops) flops[k] = m->p_flops[k];
    if (bytes) bytes[k] = m->p_bytes[k];
    m->p_launch[k] = 0;
    m->p_flops[k] = 0;
    m->p_bytes[k] = 0;
  }
  return VDR_OK;
}

}  // extern "C"
