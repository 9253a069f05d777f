"""Host-side mirror of the reference's call boundary for the hot path.

    R1  load_model(model_name, model_path)              src/tfds_dense_descriptor.py:51-67
    R2  model.image_encoder(x) / model.patch_embed(x)   src/tfds_dense_descriptor.py:122-129
        get_dense_descriptor(model, img) -> (h, w, D)   src/tfds_dense_descriptor.py:110-139
    R3  model(x[B,S,D]) -> (logits[B,C], cls[B,D])      src/models_archs.py:141-147

Same names, argument meaning and error behaviour (Python exceptions); everything below these
methods runs in libvdr.so on the MI355X.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .engine import Engine, VdrConfig

# geometries BASELINE.json names + the two the reference itself loads
ARCHS = {
    "vit_tiny16_224": VdrConfig(224, 16, 3, 192, 3, 12, 768),
    "vit_base16_224": VdrConfig(224, 16, 3, 768, 12, 12, 3072),
    "vit_large14_336": VdrConfig(336, 14, 3, 1024, 16, 24, 4096),
    "dinov2_giant14_224": VdrConfig(224, 14, 3, 1536, 24, 40, 4096, act="swiglu", layerscale=True),
    # reference default for model_name='dinov2': load_dinov2('small') = dinov2_vits14, of which the hot loop runs
    # `model.patch_embed(x)` ONLY, at 896x896 (tfds_dense_descriptor.py:128-133).  So the drop-in for that name is the
    # patch embedding alone: no blocks, no cls / pos / norm -- a real dinov2_vits14 state_dict loads as it is (its
    # pos_embed is [1, 1370, 384] for 518^2 and would need DINOv2's run-time interpolation for any other use; its
    # cls_token / mask_token / blocks.* / norm.* keys are ignored here exactly as the reference ignores them).
    "dinov2": VdrConfig(896, 14, 3, 384, 6, 0, 1536, pre_ln=False, has_cls=False, has_pos=False),
    # the whole ViT-S/14 at 896^2 (pos_embed must already be [1, 4097, 384])
    "dinov2_small14_896": VdrConfig(896, 14, 3, 384, 6, 12, 1536, layerscale=True),
    # reference default backbone: sam_model_registry['vit_b'] image encoder (MedSAM checkpoint), 1024x1024
    "medsam": VdrConfig(1024, 16, 3, 768, 12, 12, 3072, has_cls=False, window=14, global_blocks=(2, 5, 8, 11),
                        neck_chans=256),
}


def from_sam_state_dict(sd):
    """segment_anything checkpoint keys (image_encoder.* of sam_model_registry['vit_b'](path),
    tfds_dense_descriptor.py:104) -> the canonical names vdr_set_weight understands."""
    out = {}
    for k, v in sd.items():
        if not k.startswith("image_encoder."):
            continue
        k = k[len("image_encoder."):].replace(".mlp.lin1.", ".mlp.fc1.").replace(".mlp.lin2.", ".mlp.fc2.")
        out[k] = v
    return out


class VitDescriptorModel:
    """Frozen-ViT feature extractor with the attributes the reference's hot loop dispatches on."""

    def __init__(self, cfg: VdrConfig, weights: "dict[str, torch.Tensor]", model_name: str = "vit", device=None):
        self.cfg = cfg
        self.model_name = model_name  # tfds_dense_descriptor.py:66 assigns this attribute
        self.engine = Engine(cfg, device)
        self.engine.load_weights(weights)
        self.device = self.engine.device

    # -- nn.Module-style no-ops so reference code such as model.eval().cuda() keeps working
    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    # -- R2 -------------------------------------------------------------------------------------
    def patch_embed(self, x: torch.Tensor) -> torch.Tensor:
        """DINOv2 PatchEmbed: [B,3,H,W] -> [B,n,D] (tfds_dense_descriptor.py:128)."""
        return self.engine.forward(x, L.OUT_PATCH_EMBED, torch.float32)

    def image_encoder(self, x: torch.Tensor) -> torch.Tensor:
        """Channel-first dense map [B,D,h,w], the layout tfds_dense_descriptor.py:123-126 squeezes and
        transposes to (h,w,D).  SAM / MedSAM models return the conv-neck output [B,256,64,64]."""
        B = x.shape[0]
        g = self.cfg.img // self.cfg.patch
        if self.cfg.window > 0:
            return self.engine.forward(x, L.OUT_ENCODER, torch.float32).permute(0, 3, 1, 2)
        dense = self.engine.forward(x, L.OUT_DENSE, torch.float32)
        return dense.reshape(B, g, g, self.cfg.dim).permute(0, 3, 1, 2)

    def forward_features(self, x: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
        """[B,3,H,W] -> CLS features [B,D] (the [N,D] matrix embedding_classifier.py consumes)."""
        return self.engine.forward(x, L.OUT_CLS, out_dtype)

    def dense_tokens(self, x: torch.Tensor, out_dtype=torch.bfloat16) -> torch.Tensor:
        return self.engine.forward(x, L.OUT_DENSE, out_dtype)

    def __call__(self, x):
        return self.image_encoder(x) if self.cfg.window > 0 else self.forward_features(x)


def load_model(model_name: str, model_path=None, weights=None, device=None, micro_batch: int = 0, streams: int = 0,
               fp8: int = 0, full_last_block: bool = False, ln_fold: bool = True, fp8_cls_bf16: bool = False,
               resid_fp32: bool = False, ln_fin_fused: bool = False):
    """R1.  model_name: 'dinov2' | 'medsam' (reference names) or any key of ARCHS.
    model_path: a PyTorch state_dict file with the canonical key names; loaded with
    torch.load(weights_only=True).  weights: the same dict passed directly.
    fp8=True keeps the qkv / fc1 / fc2 weights as MX-fp8 and runs them on the block-scaled fp8 MFMA
    (BASELINE config 5; pre-LN models).  full_last_block=True: `model(x)` computes every token of the last block
    like the reference does before it keeps x[:, 0] (default: the CLS rows only, same bits).  fp8_cls_bf16=True (fp8 models
    with a CLS token): the MLP of the CLS rows runs on the bf16 weights (vdr_config.fp8_cls_bf16).  resid_fp32=True: the
    residual stream keeps an fp32 master copy (vdr_config.resid_fp32; bf16 path of the plain ViTs).  ln_fin_fused=True:
    the LayerNorm fold's row statistics are finalised inside the residual GEMMs instead of by a launch of their own (A/B)."""
    if model_name not in ARCHS:
        raise KeyError(f"unknown model_name {model_name!r}; known: {sorted(ARCHS)} + 'medsam'")
    cfg = VdrConfig(**{**ARCHS[model_name].__dict__, "micro_batch": micro_batch, "streams": streams,
                       "fp8": int(fp8) or int(ARCHS[model_name].fp8), "full_last_block": bool(full_last_block),
                       "ln_fold": bool(ln_fold), "fp8_cls_bf16": bool(fp8_cls_bf16), "resid_fp32": bool(resid_fp32),
                       "ln_fin_fused": bool(ln_fin_fused)})
    if weights is None:
        if model_path is None:
            raise ValueError("load_model needs model_path or weights (no network: nothing is downloaded)")
        weights = torch.load(model_path, map_location="cpu", weights_only=True)
    if model_name == "medsam" and any(k.startswith("image_encoder.") for k in weights):
        weights = from_sam_state_dict(weights)
    model = VitDescriptorModel(cfg, weights, model_name, device)
    model.model_name = model_name
    return model


def get_dense_descriptor(model, img) -> np.ndarray:
    """R2: the reference's function (tfds_dense_descriptor.py:110-139), same argument, same result layout.
    img: the RAW slice exactly as the reference passes it -- (h, w) gray or (h, w, 3) colour, values in [0, 1]; it is
    prepared here as `prepare_image` does (tfds_dense_descriptor.py:30-48: gray2rgb + resize to 1024^2 for gray,
    resize to 896^2 for colour, CHW, float32, on the device: vdr.prep.prepare_image) and run through
    `model.image_encoder` ('medsam') or `model.patch_embed` (anything else), returning (h, w, D) float32 numpy.
    Also accepted, for callers that prepared the image themselves: a (3, S, S) or (1, 3, S, S) array / tensor with S
    the model's input side (taken as it is)."""
    from . import prep
    t = torch.as_tensor(img)
    side = model.cfg.img
    prepared = t.dim() == 4 or (t.dim() == 3 and t.shape[0] == 3 and tuple(t.shape[1:]) == (side, side))
    if prepared:
        t = t.to(torch.float32)
        if t.dim() == 3:
            t = t.unsqueeze(0)
        t = t.to(model.device)
    else:
        if t.dim() not in (2, 3) or (t.dim() == 3 and t.shape[2] != 3):
            raise ValueError(f"get_dense_descriptor: a raw (h, w) / (h, w, 3) slice or a prepared (3, {side}, {side}) image, "
                             f"got {tuple(t.shape)}")
        t = prep.prepare_image(t, device=model.device)  # [1, 3, 1024 | 896, .] float32 on the device
        if t.shape[-1] != side:
            raise ValueError(f"prepare_image gives a {t.shape[-1]}^2 image for a {'gray' if torch.as_tensor(img).dim() == 2 else 'colour'} "
                             f"slice, model '{model.model_name}' takes {side}^2 (the reference pairs gray slices with "
                             "'medsam' and colour slices with 'dinov2')")
    if model.model_name == "medsam":
        f = model.image_encoder(t).cpu().numpy()
        return np.transpose(np.squeeze(f), (1, 2, 0))
    f = np.squeeze(model.patch_embed(t).cpu().numpy())
    s = int(np.sqrt(f.shape[0]))
    return f.reshape(s, s, f.shape[1])


def extract_dense(model, images: torch.Tensor, encoder: bool = True) -> np.ndarray:
    """Batched counterpart of the reference's per-slice loop (tfds_dense_descriptor.py:271-281):
    [B,3,H,W] -> (B, h, w, D) float32 numpy in one call."""
    g = model.cfg.img // model.cfg.patch
    f = model.engine.forward(images, L.OUT_DENSE if encoder else L.OUT_PATCH_EMBED, torch.float32)
    return f.reshape(images.shape[0], g, g, model.cfg.dim).cpu().numpy()


class _DropIn:
    """nn.Module-style surface the reference's checkpoint helpers use (models_archs.py:14-35: `model.state_dict()`,
    `model.load_state_dict(torch.load(path))`, `model.to(device)`, `model.eval()`)."""

    _sd = None

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, device=None):
        return self

    def state_dict(self):
        self._need_weights()
        return dict(self._sd)

    def _need_weights(self):
        if self._sd is None:
            raise RuntimeError(f"{type(self).__name__}: no weights yet — call load_state_dict(state_dict) first "
                               "(the reference's load(), models_archs.py:32-35)")


class TransformerNoduleClassifier(_DropIn):
    """R3: drop-in for models_archs.TransformerNoduleClassifier in eval mode: the reference's constructor signature
    (models_archs.py:128), then `load_state_dict(sd)` as models_archs.load does (:32-35); `state_dict=` in the
    constructor is a shortcut for the two steps.  model(x[B,S,D]) -> (logits [B,C], cls [B,D])."""

    def __init__(self, input_dim, dim_feedforward, num_heads, num_classes, num_layers, state_dict=None, device=None):
        self.cfg = VdrConfig(img=0, patch=0, in_chans=0, dim=input_dim, heads=num_heads, layers=num_layers,
                             mlp_hidden=dim_feedforward, act="gelu", pre_ln=False, layerscale=False, has_cls=True,
                             has_pos=False, input_ln=True, ln_eps=1e-5)
        self.num_classes, self.num_layers = num_classes, num_layers
        self.engine = Engine(self.cfg, device)
        self.device = self.engine.device
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        """state_dict with the reference class's key names (cls_token, norm.*, transformer_encoder.layers.{i}.*,
        classifier.dense1/2.*)."""
        from .weights import from_torch_encoder_state_dict
        sd = {k: v.detach() for k, v in state_dict.items()}
        self.engine.load_weights(from_torch_encoder_state_dict(sd, self.num_layers), strict=strict)
        # MLPLayer head (models_archs.py:186-200): dense1 -> GELU -> dense2, through vdr_op_linear
        self.head = _MlpHead(sd, "classifier", self.device)
        if self.head.n != self.num_classes:
            raise ValueError(f"classifier.dense2 has {self.head.n} rows, model was built for {self.num_classes} classes")
        self._sd = sd
        return self

    def __call__(self, x, lengths=None):
        """x [B,S,D] -> (logits, cls) as the reference.  Variable-length batches (the reference runs batch_size 1
        because its masked-voxel sequences differ in length): pass x padded to the longest sequence plus
        `lengths` [B], or a list of [S_i, D] tensors (padded here)."""
        self._need_weights()
        if isinstance(x, (list, tuple)):
            lengths = [int(t.shape[0]) for t in x]
            x = torch.nn.utils.rnn.pad_sequence([torch.as_tensor(t).float() for t in x], batch_first=True)
        cls = self.engine.forward_tokens(x, L.OUT_CLS, torch.float32, lengths=lengths)
        return self.head(cls), cls


class _MlpHead:
    """models_archs.MLPLayer (:186-200) in eval mode through vdr_op_linear; dense2's rows zero-padded to a multiple
    of 8 (the GEMM's N rule)."""

    def __init__(self, sd, prefix, dev):
        self.w1 = sd[prefix + ".dense1.weight"].to(dev, torch.bfloat16).contiguous()
        self.b1 = sd[prefix + ".dense1.bias"].to(dev, torch.float32).contiguous()
        w2, b2 = sd[prefix + ".dense2.weight"].float().cpu(), sd[prefix + ".dense2.bias"].float().cpu()
        self.n = w2.shape[0]
        npad = (self.n + 7) // 8 * 8
        wp = torch.zeros((npad, w2.shape[1]), dtype=torch.float32)
        wp[: self.n] = w2
        bp = torch.zeros((npad,), dtype=torch.float32)
        bp[: self.n] = b2
        self.w2, self.b2 = wp.to(dev, torch.bfloat16).contiguous(), bp.to(dev)

    def __call__(self, x):
        from . import ops
        hid = ops.linear(x.to(torch.bfloat16).contiguous(), self.w1, self.b1, epilogue=L.EPI_BIAS_GELU)
        out = ops.linear(hid, self.w2, self.b2, epilogue=L.EPI_BIAS)[:, : self.n].float()
        # The device erf-GELU (csrc/vdr_dev.h: max(x, 0) - a 2^Q(a), a = min(|x|, 5.7)) maps a NaN pre-activation to
        # -3e-8 -- v_min / v_max return their non-NaN operand, and a NaN-preserving form costs the VALU-bound fc1 epilogue
        # of the ViTs one to two more instructions per value.  Inside a transformer block the residual stream carries the
        # NaN on; this head has no residual, so a non-finite feature row is handed on here, as torch's gelu would:
        bad = ~torch.isfinite(x.float()).all(dim=-1, keepdim=True)
        return torch.where(bad, torch.full_like(out, float("nan")), out)


class _CrossAttentionCls:
    """CrossAttentionLayer (models_archs.py:174-183 = nn.MultiheadAttention, batch_first, no key padding mask) of
    which the bimodal forward consumes only the CLS query row (`x_attn[:, 0, :]`, :101-102).  Runs on the
    self-attention kernel: keys / values of the other modality are projected into the k | v columns of a packed
    [B*S, 3D] activation whose q columns hold the projected CLS query in row 0 of every sequence; row 0 of the
    kernel's output is then exactly softmax(q_cls K^T / sqrt(dh)) V."""

    def __init__(self, sd, prefix, heads, dev):
        W = sd[prefix + ".multihead_attn.in_proj_weight"].float()
        b = sd[prefix + ".multihead_attn.in_proj_bias"].float()
        D = W.shape[1]
        if D != heads * 64:
            raise ValueError(f"cross attention needs head dim 64 (dim {D}, heads {heads})")
        self.D, self.heads = D, heads
        self.wq, self.bq = W[:D].to(dev, torch.bfloat16).contiguous(), b[:D].to(dev).contiguous()
        self.wkv, self.bkv = W[D:].to(dev, torch.bfloat16).contiguous(), b[D:].to(dev).contiguous()
        self.wo = sd[prefix + ".multihead_attn.out_proj.weight"].to(dev, torch.bfloat16).contiguous()
        self.bo = sd[prefix + ".multihead_attn.out_proj.bias"].to(dev, torch.float32).contiguous()

    def __call__(self, xq, xkv):
        """xq [B, Sq, D], xkv [B, Sk, D] bf16 token sequences -> [B, D] fp32 (row 0 of the attention output)."""
        from . import ops
        B, Sk, D = xkv.shape
        q = ops.linear(xq[:, 0, :].contiguous(), self.wq, self.bq)                      # [B, D]
        qkv = torch.zeros((B, Sk, 3 * D), dtype=torch.bfloat16, device=xkv.device)
        qkv[:, :, D:] = ops.linear(xkv.reshape(B * Sk, D).contiguous(), self.wkv, self.bkv).view(B, Sk, 2 * D)
        qkv[:, 0, :D] = q
        o = ops.attention(qkv.view(B * Sk, 3 * D), B, Sk, self.heads).view(B, Sk, D)[:, 0, :].contiguous()
        return ops.linear(o, self.wo, self.bo).float()


class TransformerNoduleBimodalClassifier(_DropIn):
    """Drop-in for models_archs.TransformerNoduleBimodalClassifier (:38-124) in eval mode: the reference's constructor
    arguments, then `load_state_dict(sd)` (reference key names; `state_dict=` in the constructor is a shortcut);
    `model(x_ct, x_pet)` -> (logits_petct, petct_cls_token, logits_ct, logits_pet); either modality may be None, as in
    the reference."""

    def __init__(self, input_dim, mlp_ratio_ct, mlp_ratio_pet, num_heads_ct, num_heads_pet, num_layers_ct, num_layers_pet,
                 num_classes, state_dict=None, device=None):
        self.engines, self._layers = {}, {}
        for m, ratio, heads, layers in (("ct", mlp_ratio_ct, num_heads_ct, num_layers_ct),
                                        ("pet", mlp_ratio_pet, num_heads_pet, num_layers_pet)):
            cfg = VdrConfig(img=0, patch=0, in_chans=0, dim=input_dim, heads=heads, layers=layers,
                            mlp_hidden=int(ratio * input_dim), act="gelu", pre_ln=False, layerscale=False, has_cls=True,
                            has_pos=False, input_ln=True, ln_eps=1e-5)
            self.engines[m] = Engine(cfg, device)
            self._layers[m] = layers
        self.device = self.engines["ct"].device
        self.num_classes = num_classes
        self._cross_heads = num_heads_ct  # the reference constructs BOTH cross-attention layers with num_heads_ct (:71-72)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        from .weights import from_torch_encoder_state_dict
        sd = {k: v.detach() for k, v in state_dict.items()}
        for m, eng in self.engines.items():
            sub = {"cls_token": sd[f"cls_token_{m}"], "norm.weight": sd[f"norm_{m}.weight"], "norm.bias": sd[f"norm_{m}.bias"]}
            pre = f"transformer_encoder_{m}."
            sub.update({"transformer_encoder." + k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
            eng.load_weights(from_torch_encoder_state_dict(sub, self._layers[m]), strict=strict)
        dev = self.device
        self.cross = {m: _CrossAttentionCls(sd, f"cross_attention_{m}", self._cross_heads, dev) for m in ("ct", "pet")}
        self.heads = {n: _MlpHead(sd, n, dev) for n in ("classifier_ct", "classifier_pet", "projection_petct", "classifier_petct")}
        self._sd = sd
        return self

    def __call__(self, x_ct=None, x_pet=None):
        self._need_weights()
        if x_ct is None and x_pet is None:
            raise AssertionError("At least one modality should be used")  # the reference's assert
        t = {}
        for m, x in (("ct", x_ct), ("pet", x_pet)):
            if x is not None:
                mode = L.OUT_TOKENS if (x_ct is not None and x_pet is not None) else L.OUT_CLS
                t[m] = self.engines[m].forward_tokens(x, mode, torch.bfloat16 if mode == L.OUT_TOKENS else torch.float32)
        if len(t) == 2:
            ct_cls = self.cross["ct"](t["ct"], t["pet"])
            pet_cls = self.cross["pet"](t["pet"], t["ct"])
            logits_ct, logits_pet = self.heads["classifier_ct"](ct_cls), self.heads["classifier_pet"](pet_cls)
            fused = self.heads["projection_petct"](torch.cat([ct_cls, pet_cls], dim=1))
            return self.heads["classifier_petct"](fused), fused, logits_ct, logits_pet
        m = "ct" if "ct" in t else "pet"
        cls = t[m]
        lg = self.heads["classifier_" + m](cls)
        return lg, cls, lg, lg
