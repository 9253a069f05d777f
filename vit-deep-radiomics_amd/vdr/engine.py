"""Thin owner of a vdr_handle: weights in, device tensors in/out.  PyTorch supplies device memory,
the current HIP stream and (elsewhere) torch.distributed; all compute is in libvdr.so."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L


@dataclass
class VdrConfig:
    """Mirror of vdr_config (include/vdr.h)."""
    img: int = 224
    patch: int = 16
    in_chans: int = 3
    dim: int = 768
    heads: int = 12
    layers: int = 12
    mlp_hidden: int = 3072
    act: str = "gelu"
    pre_ln: bool = True
    layerscale: bool = False
    has_cls: bool = True
    has_pos: bool = True
    input_ln: bool = False
    ln_eps: float = 1e-6
    micro_batch: int = 0
    streams: int = 0
    window: int = 0            # SAM: windowed attention side (14) + decomposed rel-pos
    global_blocks: tuple = ()  # SAM: blocks with global attention (2, 5, 8, 11)
    neck_chans: int = 0        # SAM: conv neck output channels (256)
    fp8: int = 0               # 1: qkv / fc1 / fc2 as MX-fp8 (BASELINE config 5)
    ln_fold: bool = True       # pre-LN image models: LayerNorm folded into the qkv / fc1 GEMMs (False: explicit kernel)
    full_last_block: bool = False  # CLS output: True keeps every row of the last block (default: its CLS rows only,
                               # the same features bit for bit; see vdr_config.full_last_block)
    fp8_cls_bf16: bool = False  # fp8 = 1: the MLP of the CLS rows on the bf16 weights (vdr_config.fp8_cls_bf16)
    resid_fp32: bool = False  # bf16 path: fp32 master copy of the residual stream (vdr_config.resid_fp32)
    ln_fin_fused: bool = False  # LayerNorm fold: the residual GEMMs finalise the row statistics themselves (vdr_config.ln_fin_fused)

    @property
    def n_patches(self):
        return (self.img // self.patch) ** 2 if self.patch else 0

    @property
    def n_tokens(self):
        return self.n_patches + (1 if self.has_cls else 0)

    def to_c(self) -> L.vdr_config:
        c = L.vdr_config()
        c.img, c.patch, c.in_chans, c.dim, c.heads, c.layers = self.img, self.patch, self.in_chans, self.dim, self.heads, self.layers
        c.mlp_hidden = self.mlp_hidden
        c.act = L.ACT_SWIGLU if self.act == "swiglu" else L.ACT_GELU
        c.pre_ln, c.layerscale, c.has_cls, c.has_pos = int(self.pre_ln), int(self.layerscale), int(self.has_cls), int(self.has_pos)
        c.input_ln, c.ln_eps, c.micro_batch = int(self.input_ln), float(self.ln_eps), int(self.micro_batch)
        c.streams = int(self.streams)
        c.window, c.neck_chans = int(self.window), int(self.neck_chans)
        c.global_mask = sum(1 << int(i) for i in self.global_blocks)
        c.fp8 = int(self.fp8)
        c.no_ln_fold = int(not self.ln_fold)
        c.full_last_block = int(self.full_last_block)
        c.fp8_cls_bf16 = int(self.fp8_cls_bf16)
        c.resid_fp32 = int(self.resid_fp32)
        c.ln_fin_fused = int(self.ln_fin_fused)
        return c


_DT = {torch.float32: L.VDR_F32, torch.bfloat16: L.VDR_BF16}


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Engine:
    def __init__(self, cfg: VdrConfig, device=None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.VdrError(-2, "no HIP device visible: libvdr has no CPU path")
        self.cfg = cfg
        dev = torch.device("cuda") if device is None else torch.device(device)
        if dev.type != "cuda":
            raise L.VdrError(-1, f"libvdr runs on a HIP device, got {dev}")
        # always an indexed device: torch.device("cuda") means the current one
        self.device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        h = C.c_void_p()
        cc = cfg.to_c()
        L.check(self.lib.vdr_create(C.byref(cc), self.device.index, C.byref(h)))
        self.h = h
        self._ws = None
        self._loaded = False

    def close(self):
        if getattr(self, "h", None):
            self.lib.vdr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def weight_names(self):
        return [self.lib.vdr_weight_name(self.h, i).decode() for i in range(self.lib.vdr_num_weights(self.h))]

    def load_weights(self, weights: "dict[str, torch.Tensor]", strict=True):
        """weights: canonical (timm/DINOv2-style) names -> tensors in the PyTorch layout."""
        names = self.weight_names()
        missing = [n for n in names if n not in weights]
        if strict and missing:
            raise KeyError(f"missing weights: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for n in names:
            if n not in weights:
                continue
            t = weights[n].detach().to("cpu", torch.float32).contiguous()
            a = t.numpy()
            shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            L.check(self.lib.vdr_set_weight(self.h, n.encode(), a.ctypes.data_as(C.c_void_p), shape, max(a.ndim, 1)), self.h)
        if not missing:
            self.finalize()

    def finalize(self):
        """vdr_finalize: LayerNorm folding, packed GEMM layouts, fp8 copies, rel-pos tables.  Load time (synchronous);
        after it the forward calls neither allocate nor synchronise, so even the first one can be graph-captured."""
        L.check(self.lib.vdr_finalize(self.h), self.h)
        self._loaded = True

    # ---- workspace ------------------------------------------------------------------------------
    def _workspace(self, batch: int, seq: int = 0) -> torch.Tensor:
        need = C.c_size_t()
        L.check(self.lib.vdr_workspace_bytes(self.h, batch, seq, C.byref(need)), self.h)
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            # held by the engine: survives torch.cuda.empty_cache() between calls
            # (the reference calls it after every slice, tfds_dense_descriptor.py:137)
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- hot path --------------------------------------------------------------------------------
    def forward(self, images: torch.Tensor, out_mode: int = L.OUT_CLS, out_dtype=torch.float32) -> torch.Tensor:
        cfg = self.cfg
        if images.dim() != 4 or images.shape[1] != cfg.in_chans or images.shape[2] != cfg.img or images.shape[3] != cfg.img:
            raise ValueError(f"images must be [B,{cfg.in_chans},{cfg.img},{cfg.img}], got {tuple(images.shape)}")
        if images.dtype not in _DT:
            images = images.float()
        images = images.to(self.device).contiguous()
        B = images.shape[0]
        n, N, D = cfg.n_patches, cfg.n_tokens, cfg.dim
        g = cfg.img // cfg.patch
        shape = {L.OUT_CLS: (B, D), L.OUT_DENSE: (B, n, D), L.OUT_PATCH_EMBED: (B, n, D), L.OUT_TOKENS: (B, N, D),
                 L.OUT_ENCODER: (B, g, g, cfg.neck_chans)}[out_mode]
        out = torch.empty(shape, dtype=out_dtype, device=self.device)
        ws = self._workspace(B)
        L.check(self.lib.vdr_forward(self.h, images.data_ptr(), _DT[images.dtype], B, out.data_ptr(), out_mode,
                                     _DT[out_dtype], ws.data_ptr(), ws.numel(), _stream_ptr(self.device)), self.h)
        return out

    def forward_into(self, images: torch.Tensor, out: torch.Tensor, out_mode: int = L.OUT_CLS):
        """As forward(), writing into a caller-owned buffer (e.g. this rank's slice of the all-gather buffer).
        Nothing is converted or copied here, so everything is checked: a wrong dtype / layout / size is an error, never
        an out-of-bounds write."""
        cfg = self.cfg
        if images.dim() != 4 or tuple(images.shape[1:]) != (cfg.in_chans, cfg.img, cfg.img):
            raise ValueError(f"images must be [B,{cfg.in_chans},{cfg.img},{cfg.img}], got {tuple(images.shape)}")
        if images.dtype not in _DT or out.dtype not in _DT:
            raise TypeError(f"images / out must be float32 or bfloat16, got {images.dtype} / {out.dtype}")
        if images.device != self.device or out.device != self.device:
            raise ValueError(f"images and out must live on {self.device}")
        if not images.is_contiguous() or not out.is_contiguous():
            raise ValueError("images and out must be contiguous")
        B = images.shape[0]
        g = cfg.img // cfg.patch
        per_image = {L.OUT_CLS: cfg.dim, L.OUT_DENSE: cfg.n_patches * cfg.dim, L.OUT_PATCH_EMBED: cfg.n_patches * cfg.dim,
                     L.OUT_TOKENS: cfg.n_tokens * cfg.dim, L.OUT_ENCODER: g * g * cfg.neck_chans}[out_mode]
        if out.numel() != B * per_image:
            raise ValueError(f"out has {out.numel()} elements, the forward writes {B} x {per_image}")
        ws = self._workspace(B)
        L.check(self.lib.vdr_forward(self.h, images.data_ptr(), _DT[images.dtype], B, out.data_ptr(), out_mode,
                                     _DT[out.dtype], ws.data_ptr(), ws.numel(), _stream_ptr(self.device)), self.h)
        return out

    def forward_tokens(self, tokens: torch.Tensor, out_mode: int = L.OUT_CLS, out_dtype=torch.float32,
                       lengths=None) -> torch.Tensor:
        """tokens [B,S,D]; lengths (optional, int [B]): sequence b is tokens[b, :lengths[b]], the rest padding."""
        cfg = self.cfg
        if tokens.dim() != 3 or tokens.shape[2] != cfg.dim:
            raise ValueError(f"tokens must be [B,S,{cfg.dim}], got {tuple(tokens.shape)}")
        if tokens.dtype not in _DT:
            tokens = tokens.float()
        tokens = tokens.to(self.device).contiguous()
        B, S, D = tokens.shape
        c = 1 if cfg.has_cls else 0
        shape = {L.OUT_CLS: (B, D), L.OUT_DENSE: (B, S, D), L.OUT_TOKENS: (B, S + c, D)}[out_mode]
        out = torch.empty(shape, dtype=out_dtype, device=self.device)
        ws = self._workspace(B, S)
        if lengths is None:
            L.check(self.lib.vdr_forward_tokens(self.h, tokens.data_ptr(), _DT[tokens.dtype], B, S, out.data_ptr(), out_mode,
                                                _DT[out_dtype], ws.data_ptr(), ws.numel(), _stream_ptr(self.device)), self.h)
            return out
        lens = torch.as_tensor(lengths, dtype=torch.int32)
        if lens.shape != (B,) or int(lens.min()) < 1 or int(lens.max()) > S:
            raise ValueError(f"lengths must be [B] with 1 <= length <= {S}")
        lens = lens.to(self.device).contiguous()
        L.check(self.lib.vdr_forward_tokens_varlen(self.h, tokens.data_ptr(), _DT[tokens.dtype], B, S, lens.data_ptr(),
                                                   out.data_ptr(), out_mode, _DT[out_dtype], ws.data_ptr(), ws.numel(),
                                                   _stream_ptr(self.device)), self.h)
        return out

    # ---- profiler ---------------------------------------------------------------------------------
    def profile(self, on: bool, classes=None):
        """Bracket kernel launches with HIP events; classes = iterable of class names to restrict to."""
        mask = 0xFFFFFFFF
        if classes is not None:
            names = [self.lib.vdr_kernel_class_name(k).decode() for k in range(L.K_COUNT)]
            mask = 0
            for c in classes:
                mask |= 1 << names.index(c)
        L.check(self.lib.vdr_profile_mask(self.h, mask), self.h)
        L.check(self.lib.vdr_profile_enable(self.h, int(on)), self.h)

    def profile_read(self):
        ms = (C.c_double * L.K_COUNT)()
        ln = (C.c_int64 * L.K_COUNT)()
        fl = (C.c_double * L.K_COUNT)()
        by = (C.c_double * L.K_COUNT)()
        L.check(self.lib.vdr_profile_read(self.h, ms, ln, fl, by, L.K_COUNT), self.h)
        out = {}
        for k in range(L.K_COUNT):
            if ln[k]:
                out[self.lib.vdr_kernel_class_name(k).decode()] = {"ms": ms[k], "launches": int(ln[k]), "flops": fl[k], "bytes": by[k]}
        return out
