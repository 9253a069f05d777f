"""state_dict key translation (host logic, CPU-testable)."""
from __future__ import annotations

import torch


def from_torch_encoder_state_dict(sd, layers: int):
    """nn.TransformerEncoder / TransformerNoduleClassifier keys (models_archs.py:127-139; SURVEY.md §8a
    weight-name map) -> the canonical names vdr_set_weight understands."""
    out = {"cls_token": sd["cls_token"], "input_norm.weight": sd["norm.weight"], "input_norm.bias": sd["norm.bias"]}
    for i in range(layers):
        s, d = f"transformer_encoder.layers.{i}.", f"blocks.{i}."
        out[d + "attn.qkv.weight"] = sd[s + "self_attn.in_proj_weight"]
        out[d + "attn.qkv.bias"] = sd[s + "self_attn.in_proj_bias"]
        out[d + "attn.proj.weight"] = sd[s + "self_attn.out_proj.weight"]
        out[d + "attn.proj.bias"] = sd[s + "self_attn.out_proj.bias"]
        out[d + "mlp.fc1.weight"] = sd[s + "linear1.weight"]
        out[d + "mlp.fc1.bias"] = sd[s + "linear1.bias"]
        out[d + "mlp.fc2.weight"] = sd[s + "linear2.weight"]
        out[d + "mlp.fc2.bias"] = sd[s + "linear2.bias"]
        for n in ("norm1", "norm2"):
            out[d + n + ".weight"] = sd[s + n + ".weight"]
            out[d + n + ".bias"] = sd[s + n + ".bias"]
    return {k: torch.as_tensor(v).detach().to(torch.float32).contiguous() for k, v in out.items()}


def expected_weight_shapes(cfg) -> "dict[str, tuple]":
    """Names and PyTorch shapes a config expects (same list vdr_weight_name enumerates)."""
    D, Fh = cfg.dim, cfg.mlp_hidden
    s = {}
    if cfg.patch:
        s["patch_embed.proj.weight"] = (D, cfg.in_chans, cfg.patch, cfg.patch)
        s["patch_embed.proj.bias"] = (D,)
    if cfg.has_cls:
        s["cls_token"] = (1, 1, D)
    if cfg.has_pos:
        s["pos_embed"] = (1, cfg.n_tokens, D)
    if cfg.input_ln:
        s["input_norm.weight"] = (D,)
        s["input_norm.bias"] = (D,)
    for i in range(cfg.layers):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        if cfg.layerscale:
            s[p + "ls1.gamma"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        if cfg.act == "swiglu":
            s[p + "mlp.w12.weight"] = (2 * Fh, D)
            s[p + "mlp.w12.bias"] = (2 * Fh,)
            s[p + "mlp.w3.weight"] = (D, Fh)
            s[p + "mlp.w3.bias"] = (D,)
        else:
            s[p + "mlp.fc1.weight"] = (Fh, D)
            s[p + "mlp.fc1.bias"] = (Fh,)
            s[p + "mlp.fc2.weight"] = (D, Fh)
            s[p + "mlp.fc2.bias"] = (D,)
        if cfg.layerscale:
            s[p + "ls2.gamma"] = (D,)
    if cfg.pre_ln:
        s["norm.weight"] = (D,)
        s["norm.bias"] = (D,)
    return s
