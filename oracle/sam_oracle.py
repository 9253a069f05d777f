"""CPU fp32 oracle for the SAM / MedSAM ViT image encoder — the backbone the reference runs by default.

TEST INFRASTRUCTURE ONLY (same rules as vit_oracle.py: never imported by the product package).

Restates the call at ``src/tfds_dense_descriptor.py:123`` ``model.image_encoder(x)`` for
``sam_model_registry['vit_b']`` (``src/tfds_dense_descriptor.py:104``).  The third-party package
``segment_anything`` is NOT in /root/reference (no requirements file, unpinned) and is not installed
offline, so this is a restatement of its published ``ImageEncoderViT``:

    patch_embed Conv2d(3, D, 16, 16) -> NHWC tokens + abs pos_embed [1, g, g, D]
    L x Block:  x = x + unwindow(Attn(window(norm1(x))))      window 14 (zero-padded to a multiple
                x = x + MLP(norm2(x))                          of 14 AFTER norm1), global at 2,5,8,11
        Attn: qkv Linear(+bias), scale = dh^-0.5, decomposed relative position bias
              attn[q,k] += q . Rh[qh - kh] + q . Rw[qw - kw]   (unscaled q), softmax, proj
    neck: Conv2d(D, C, 1, bias=False) -> LayerNorm2d -> Conv2d(C, C, 3, pad=1, bias=False) -> LayerNorm2d
    output [B, C, g, g]

Parity pinning: unpinned by the reference (package absent); cross-checked against the in-container
``transformers`` ``SamVisionModel`` (tests/golden/make_golden.py -> tests/golden/sam_hf_*.npz).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from .vit_oracle import _q, _r, _wq, gelu_erf, layer_norm


@dataclass
class SamCfg:
    img: int = 1024
    patch: int = 16
    in_chans: int = 3
    dim: int = 768
    heads: int = 12
    layers: int = 12
    mlp_hidden: int = 3072
    window: int = 14
    global_idx: tuple = (2, 5, 8, 11)
    out_chans: int = 256
    ln_eps: float = 1e-6

    @property
    def grid(self):
        return self.img // self.patch


SAM_VIT_B = SamCfg()


def weight_shapes(cfg: SamCfg):
    D, Fh, C, g = cfg.dim, cfg.mlp_hidden, cfg.out_chans, cfg.grid
    dh = D // cfg.heads
    s = {"patch_embed.proj.weight": (D, cfg.in_chans, cfg.patch, cfg.patch), "patch_embed.proj.bias": (D,),
         "pos_embed": (1, g, g, D)}
    for i in range(cfg.layers):
        p = f"blocks.{i}."
        size = g if i in cfg.global_idx else cfg.window
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.rel_pos_h"] = (2 * size - 1, dh)
        s[p + "attn.rel_pos_w"] = (2 * size - 1, dh)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Fh, D)
        s[p + "mlp.fc1.bias"] = (Fh,)
        s[p + "mlp.fc2.weight"] = (D, Fh)
        s[p + "mlp.fc2.bias"] = (D,)
    s["neck.0.weight"] = (C, D, 1, 1)
    s["neck.1.weight"] = (C,)
    s["neck.1.bias"] = (C,)
    s["neck.2.weight"] = (C, C, 3, 3)
    s["neck.3.weight"] = (C,)
    s["neck.3.bias"] = (C,)
    return s


def make_weights(cfg: SamCfg, seed: int = 1, scale: float = 0.02):
    out = {}
    for idx, (name, shape) in enumerate(weight_shapes(cfg).items()):
        rng = np.random.Generator(np.random.PCG64([seed, 1000 + idx]))
        z = rng.standard_normal(size=shape, dtype=np.float32)
        if "norm" in name and name.endswith(".weight") or name in ("neck.1.weight", "neck.3.weight"):
            z = 1.0 + 0.1 * z
        elif ("norm" in name and name.endswith(".bias")) or name in ("neck.1.bias", "neck.3.bias"):
            z = 0.1 * z
        elif name == "pos_embed":
            z = 0.02 * z
        elif "rel_pos" in name:
            z = 0.1 * z
        else:
            z = scale * z
        out[name] = torch.from_numpy(np.ascontiguousarray(z.astype(np.float32)))
    return out


def make_images(cfg: SamCfg, batch: int, seed: int = 0):
    rng = np.random.Generator(np.random.PCG64([seed, 17]))
    return torch.from_numpy(rng.random(size=(batch, cfg.in_chans, cfg.img, cfg.img), dtype=np.float32))


def window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def rel_table(size: int, rel_pos):
    """Rh[q, k, :] = rel_pos[q - k + size - 1]  (get_rel_pos with q_size == k_size: no interpolation)."""
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + (size - 1)
    return rel_pos[idx]


def sam_attention(x, w, p, heads, emulate=False):
    """x [Bw, S, S, D] (one window or the whole grid per batch entry)."""
    Bw, S, _, D = x.shape
    dh = D // heads
    qkv = _r(x.reshape(Bw, S * S, D) @ _wq(w[p + "qkv.weight"], emulate).t() + w[p + "qkv.bias"], emulate)
    q, k, v = qkv.reshape(Bw, S * S, 3, heads, dh).permute(2, 0, 3, 1, 4)  # [Bw, H, N, dh]
    attn = (q * (dh ** -0.5)) @ k.transpose(-1, -2)
    Rh, Rw = rel_table(S, w[p + "rel_pos_h"]), rel_table(S, w[p + "rel_pos_w"])
    rq = q.reshape(Bw, heads, S, S, dh)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw)
    attn = attn.view(Bw, heads, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]
    attn = attn.view(Bw, heads, S * S, S * S)
    attn = attn - attn.amax(dim=-1, keepdim=True)
    pexp = torch.exp(attn)
    l = pexp.sum(dim=-1, keepdim=True)
    o = (_r(pexp, emulate) @ v) / l if emulate else (pexp / l) @ v
    o = _r(o.transpose(1, 2).reshape(Bw, S, S, D), emulate)
    return o @ _r(w[p + "proj.weight"], bool(emulate)).t() + w[p + "proj.bias"]  # the out-projection stays bf16 (fp8 level 1)


def layer_norm_2d(x_nhwc, gamma, beta, eps):
    """segment_anything LayerNorm2d: normalise over channels per pixel (NHWC here)."""
    return layer_norm(x_nhwc, gamma, beta, eps)


@torch.no_grad()
def sam_forward(cfg: SamCfg, w, images, emulate_bf16=False):
    """[B,3,img,img] -> dict(tokens [B,g,g,D] after the blocks, out [B,C,g,g] after the neck).
    emulate_bf16="mx": additionally the MX-fp8 quantisation points of the fp8 path (qkv / fc1 / fc2 operands)."""
    em = emulate_bf16
    images = images.to(torch.float32)
    B = images.shape[0]
    g, D, p = cfg.grid, cfg.dim, cfg.patch
    cols = images.reshape(B, cfg.in_chans, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
    wmat = w["patch_embed.proj.weight"].reshape(D, -1)
    x = _r(cols, em) @ _r(wmat, em).t() + w["patch_embed.proj.bias"]
    x = _r(x.reshape(B, g, g, D) + w["pos_embed"], em)
    for i in range(cfg.layers):
        pfx = f"blocks.{i}."
        h = _q(layer_norm(x, w[pfx + "norm1.weight"], w[pfx + "norm1.bias"], cfg.ln_eps), em)
        if i in cfg.global_idx:
            a = sam_attention(h, w, pfx + "attn.", cfg.heads, em)
        else:
            hw, pad_hw = window_partition(h, cfg.window)
            a = window_unpartition(sam_attention(hw, w, pfx + "attn.", cfg.heads, em), cfg.window, pad_hw, (g, g))
        x = _r(x + a, em)
        h = _q(layer_norm(x, w[pfx + "norm2.weight"], w[pfx + "norm2.bias"], cfg.ln_eps), em)
        u = _q(gelu_erf(h @ _wq(w[pfx + "mlp.fc1.weight"], em).t() + w[pfx + "mlp.fc1.bias"]), em)
        x = _r(x + u @ _wq(w[pfx + "mlp.fc2.weight"], em).t() + w[pfx + "mlp.fc2.bias"], em)
    tokens = x
    C = cfg.out_chans
    y = _r(x, em) @ _r(w["neck.0.weight"].reshape(C, D), em).t()
    y = _r(layer_norm_2d(_r(y, em), w["neck.1.weight"], w["neck.1.bias"], cfg.ln_eps), em)
    y = F.conv2d(y.permute(0, 3, 1, 2), _r(w["neck.2.weight"], em), None, padding=1).permute(0, 2, 3, 1)
    y = layer_norm_2d(_r(y, em), w["neck.3.weight"], w["neck.3.bias"], cfg.ln_eps)
    return {"tokens": tokens, "out": y.permute(0, 3, 1, 2).contiguous()}


def flops_per_image(cfg: SamCfg) -> float:
    g, D, Fh, C = cfg.grid, cfg.dim, cfg.mlp_hidden, cfg.out_chans
    n = g * g
    ws = cfg.window
    gp = (g + ws - 1) // ws * ws
    nwin = (gp // ws) ** 2
    lin = 2.0 * n * D * (3 * D + D + 2 * Fh)
    tot = 2.0 * n * cfg.in_chans * cfg.patch ** 2 * D
    for i in range(cfg.layers):
        if i in cfg.global_idx:
            tot += lin + 4.0 * n * n * D
        else:
            tot += lin + (2.0 * (gp * gp - n) * D * 3 * D) + 4.0 * nwin * (ws * ws) ** 2 * D
    tot += 2.0 * n * D * C + 2.0 * n * C * C * 9
    return tot
