"""Generate the golden vectors under tests/golden/ (run ONCE in the authoring container).

    python tests/golden/make_golden.py

* ``postln_*.npz`` — expected outputs of the REFERENCE's own class
  ``models_archs.TransformerNoduleClassifier`` (/root/reference/src/models_archs.py:127-147),
  imported here and run on CPU fp32 in eval()/no_grad.  Its parameters are overwritten with
  the oracle's seeded numpy weights (``oracle.vit_oracle.make_weights``) through the inverse of
  ``from_torch_encoder_state_dict`` so the fixtures only need to store seeds + expected
  outputs; the classifier-head weights (MLPLayer, models_archs.py:186-200) are stored in full
  because the oracle has no generator for them.
* ``vit_hf_*.npz`` / ``dinov2_hf_*.npz`` — architecture cross-checks (SURVEY.md §8c "O3") from
  the in-container ``transformers`` ``ViTModel`` / ``Dinov2Model`` built from local Config
  objects (no download) with the same seeded weights.  These are NOT the reference; they guard
  the pre-LN restatement against being self-consistent but wrong.

/root/reference never travels to the GPU box: only the .npz files (inputs/seeds and expected
outputs — data, no source) are committed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

from oracle import vit_oracle as vo  # noqa: E402


def to_torch_encoder_state_dict(w, layers):
    sd = {"cls_token": w["cls_token"], "norm.weight": w["input_norm.weight"], "norm.bias": w["input_norm.bias"]}
    for i in range(layers):
        s, d = f"blocks.{i}.", f"transformer_encoder.layers.{i}."
        sd[d + "self_attn.in_proj_weight"] = w[s + "attn.qkv.weight"]
        sd[d + "self_attn.in_proj_bias"] = w[s + "attn.qkv.bias"]
        sd[d + "self_attn.out_proj.weight"] = w[s + "attn.proj.weight"]
        sd[d + "self_attn.out_proj.bias"] = w[s + "attn.proj.bias"]
        sd[d + "linear1.weight"] = w[s + "mlp.fc1.weight"]
        sd[d + "linear1.bias"] = w[s + "mlp.fc1.bias"]
        sd[d + "linear2.weight"] = w[s + "mlp.fc2.weight"]
        sd[d + "linear2.bias"] = w[s + "mlp.fc2.bias"]
        for n in ("norm1", "norm2"):
            sd[d + n + ".weight"] = w[s + n + ".weight"]
            sd[d + n + ".bias"] = w[s + n + ".bias"]
    return sd


def gen_postln(tag, dim, heads, layers, ffn, batch, seq, wseed, xseed, wscale):
    import models_archs  # the reference module (torch only)

    cfg = vo.postln_cfg(dim, heads, layers, ffn)
    w = vo.make_weights(cfg, seed=wseed, scale=wscale)
    torch.manual_seed(1234)
    ref = models_archs.TransformerNoduleClassifier(input_dim=dim, dim_feedforward=ffn, num_heads=heads,
                                                   num_classes=2, num_layers=layers)
    sd = ref.state_dict()
    new = to_torch_encoder_state_dict(w, layers)
    for k, v in new.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = v.clone()
    ref.load_state_dict(sd)
    ref.eval()
    x = vo.make_tokens(batch, seq, dim, seed=xseed)
    with torch.no_grad():
        logits, cls = ref(x)
    head = {k: sd[k].numpy() for k in sd if k.startswith("classifier.")}
    np.savez_compressed(os.path.join(HERE, f"postln_{tag}.npz"),
                        dim=dim, heads=heads, layers=layers, ffn=ffn, batch=batch, seq=seq,
                        wseed=wseed, xseed=xseed, wscale=wscale,
                        logits=logits.numpy(), cls=cls.numpy(),
                        x_probe=x[0, :2, :8].numpy(),
                        w_probe=w["blocks.0.attn.qkv.weight"][:2, :8].numpy(),
                        **{"head." + k: v for k, v in head.items()})
    # sanity: the restatement agrees with the reference right now
    o = vo.forward_tokens(cfg, w, x)
    err = (o["cls"] - cls).abs().max().item()
    print(f"postln_{tag}: cls {tuple(cls.shape)} max|oracle-ref| = {err:.3e}")
    assert err < 2e-5, err


def gen_vit_hf(tag, img, patch, dim, heads, layers, ffn, batch, wseed, xseed):
    from transformers import ViTConfig, ViTModel

    cfg = vo.VitCfg(img, patch, 3, dim, heads, layers, ffn, ln_eps=1e-6)
    w = vo.make_weights(cfg, seed=wseed, scale=0.05)
    hc = ViTConfig(hidden_size=dim, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn,
                   image_size=img, patch_size=patch, layer_norm_eps=1e-6, hidden_act="gelu",
                   hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = ViTModel(hc, add_pooling_layer=False)
    sd = m.state_dict()
    sd["embeddings.cls_token"] = w["cls_token"]
    sd["embeddings.position_embeddings"] = w["pos_embed"]
    sd["embeddings.patch_embeddings.projection.weight"] = w["patch_embed.proj.weight"]
    sd["embeddings.patch_embeddings.projection.bias"] = w["patch_embed.proj.bias"]
    for i in range(layers):
        s, d = f"blocks.{i}.", f"layers.{i}."
        q, k, v = w[s + "attn.qkv.weight"].chunk(3, dim=0)
        qb, kb, vb = w[s + "attn.qkv.bias"].chunk(3, dim=0)
        for nm, ww, bb in (("q_proj", q, qb), ("k_proj", k, kb), ("v_proj", v, vb)):
            sd[d + f"attention.{nm}.weight"] = ww.clone()
            sd[d + f"attention.{nm}.bias"] = bb.clone()
        sd[d + "attention.o_proj.weight"] = w[s + "attn.proj.weight"]
        sd[d + "attention.o_proj.bias"] = w[s + "attn.proj.bias"]
        sd[d + "layernorm_before.weight"] = w[s + "norm1.weight"]
        sd[d + "layernorm_before.bias"] = w[s + "norm1.bias"]
        sd[d + "layernorm_after.weight"] = w[s + "norm2.weight"]
        sd[d + "layernorm_after.bias"] = w[s + "norm2.bias"]
        sd[d + "mlp.fc1.weight"] = w[s + "mlp.fc1.weight"]
        sd[d + "mlp.fc1.bias"] = w[s + "mlp.fc1.bias"]
        sd[d + "mlp.fc2.weight"] = w[s + "mlp.fc2.weight"]
        sd[d + "mlp.fc2.bias"] = w[s + "mlp.fc2.bias"]
    sd["layernorm.weight"] = w["norm.weight"]
    sd["layernorm.bias"] = w["norm.bias"]
    m.load_state_dict(sd)
    m.eval()
    x = vo.make_images(cfg, batch, seed=xseed)
    with torch.no_grad():
        hs = m(pixel_values=x).last_hidden_state
    np.savez_compressed(os.path.join(HERE, f"vit_hf_{tag}.npz"), img=img, patch=patch, dim=dim, heads=heads,
                        layers=layers, ffn=ffn, batch=batch, wseed=wseed, xseed=xseed, wscale=0.05,
                        tokens=hs.numpy())
    o = vo.forward_images(cfg, w, x)
    err = (o["tokens"] - hs).abs().max().item()
    print(f"vit_hf_{tag}: tokens {tuple(hs.shape)} max|oracle-hf| = {err:.3e}")
    assert err < 5e-5, err


def gen_dinov2_hf(tag, img, patch, dim, heads, layers, batch, wseed, xseed):
    from transformers import Dinov2Config, Dinov2Model

    ffn = (int(dim * 4 * 2 / 3) + 7) // 8 * 8
    cfg = vo.VitCfg(img, patch, 3, dim, heads, layers, ffn, act="swiglu", layerscale=True, ln_eps=1e-6)
    w = vo.make_weights(cfg, seed=wseed, scale=0.05)
    hc = Dinov2Config(hidden_size=dim, num_hidden_layers=layers, num_attention_heads=heads, mlp_ratio=4,
                      image_size=img, patch_size=patch, use_swiglu_ffn=True, layer_norm_eps=1e-6,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
    m = Dinov2Model(hc)
    sd = m.state_dict()
    sd["embeddings.cls_token"] = w["cls_token"]
    sd["embeddings.position_embeddings"] = w["pos_embed"]
    sd["embeddings.patch_embeddings.projection.weight"] = w["patch_embed.proj.weight"]
    sd["embeddings.patch_embeddings.projection.bias"] = w["patch_embed.proj.bias"]
    for i in range(layers):
        s, d = f"blocks.{i}.", f"encoder.layer.{i}."
        q, k, v = w[s + "attn.qkv.weight"].chunk(3, dim=0)
        qb, kb, vb = w[s + "attn.qkv.bias"].chunk(3, dim=0)
        for nm, ww, bb in (("query", q, qb), ("key", k, kb), ("value", v, vb)):
            sd[d + f"attention.attention.{nm}.weight"] = ww.clone()
            sd[d + f"attention.attention.{nm}.bias"] = bb.clone()
        sd[d + "attention.output.dense.weight"] = w[s + "attn.proj.weight"]
        sd[d + "attention.output.dense.bias"] = w[s + "attn.proj.bias"]
        sd[d + "layer_scale1.lambda1"] = w[s + "ls1.gamma"]
        sd[d + "layer_scale2.lambda1"] = w[s + "ls2.gamma"]
        for n in ("norm1", "norm2"):
            sd[d + n + ".weight"] = w[s + n + ".weight"]
            sd[d + n + ".bias"] = w[s + n + ".bias"]
        sd[d + "mlp.weights_in.weight"] = w[s + "mlp.w12.weight"]
        sd[d + "mlp.weights_in.bias"] = w[s + "mlp.w12.bias"]
        sd[d + "mlp.weights_out.weight"] = w[s + "mlp.w3.weight"]
        sd[d + "mlp.weights_out.bias"] = w[s + "mlp.w3.bias"]
    sd["layernorm.weight"] = w["norm.weight"]
    sd["layernorm.bias"] = w["norm.bias"]
    m.load_state_dict(sd)
    m.eval()
    x = vo.make_images(cfg, batch, seed=xseed)
    with torch.no_grad():
        hs = m(pixel_values=x).last_hidden_state
    np.savez_compressed(os.path.join(HERE, f"dinov2_hf_{tag}.npz"), img=img, patch=patch, dim=dim, heads=heads,
                        layers=layers, ffn=ffn, batch=batch, wseed=wseed, xseed=xseed, wscale=0.05,
                        tokens=hs.numpy())
    o = vo.forward_images(cfg, w, x)
    err = (o["tokens"] - hs).abs().max().item()
    print(f"dinov2_hf_{tag}: tokens {tuple(hs.shape)} max|oracle-hf| = {err:.3e}")
    assert err < 5e-5, err


def gen_sam_hf(tag, img, patch, dim, heads, layers, ffn, window, global_idx, out_chans, batch, wseed, xseed):
    """Architecture cross-check of oracle/sam_oracle.py against transformers.SamVisionModel (not the reference:
    segment_anything is absent from /root/reference and from this container)."""
    from transformers import SamVisionConfig, SamVisionModel
    from oracle import sam_oracle as so

    cfg = so.SamCfg(img, patch, 3, dim, heads, layers, ffn, window, tuple(global_idx), out_chans, 1e-6)
    w = so.make_weights(cfg, seed=wseed, scale=0.05)
    hc = SamVisionConfig(hidden_size=dim, output_channels=out_chans, num_hidden_layers=layers, num_attention_heads=heads,
                         image_size=img, patch_size=patch, window_size=window, global_attn_indexes=list(global_idx),
                         mlp_dim=ffn, layer_norm_eps=1e-6, use_abs_pos=True, use_rel_pos=True, qkv_bias=True,
                         hidden_act="gelu", attention_dropout=0.0)
    m = SamVisionModel(hc)
    sd = m.state_dict()
    pre = "vision_encoder."
    sd[pre + "pos_embed"] = w["pos_embed"]
    sd[pre + "patch_embed.projection.weight"] = w["patch_embed.proj.weight"]
    sd[pre + "patch_embed.projection.bias"] = w["patch_embed.proj.bias"]
    for i in range(layers):
        s_, d = f"blocks.{i}.", pre + f"layers.{i}."
        sd[d + "layer_norm1.weight"], sd[d + "layer_norm1.bias"] = w[s_ + "norm1.weight"], w[s_ + "norm1.bias"]
        sd[d + "layer_norm2.weight"], sd[d + "layer_norm2.bias"] = w[s_ + "norm2.weight"], w[s_ + "norm2.bias"]
        for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "rel_pos_h", "rel_pos_w"):
            sd[d + "attn." + k] = w[s_ + "attn." + k]
        sd[d + "mlp.lin1.weight"], sd[d + "mlp.lin1.bias"] = w[s_ + "mlp.fc1.weight"], w[s_ + "mlp.fc1.bias"]
        sd[d + "mlp.lin2.weight"], sd[d + "mlp.lin2.bias"] = w[s_ + "mlp.fc2.weight"], w[s_ + "mlp.fc2.bias"]
    sd[pre + "neck.conv1.weight"] = w["neck.0.weight"]
    sd[pre + "neck.layer_norm1.weight"], sd[pre + "neck.layer_norm1.bias"] = w["neck.1.weight"], w["neck.1.bias"]
    sd[pre + "neck.conv2.weight"] = w["neck.2.weight"]
    sd[pre + "neck.layer_norm2.weight"], sd[pre + "neck.layer_norm2.bias"] = w["neck.3.weight"], w["neck.3.bias"]
    for k, v in sd.items():
        assert m.state_dict()[k].shape == v.shape, (k, m.state_dict()[k].shape, v.shape)
    m.load_state_dict(sd)
    m.eval()
    x = so.make_images(cfg, batch, seed=xseed)
    with torch.no_grad():
        out = m(pixel_values=x).last_hidden_state
    np.savez_compressed(os.path.join(HERE, f"sam_hf_{tag}.npz"), img=img, patch=patch, dim=dim, heads=heads, layers=layers,
                        ffn=ffn, window=window, global_idx=np.array(global_idx), out_chans=out_chans, batch=batch,
                        wseed=wseed, xseed=xseed, wscale=0.05, out=out.numpy())
    o = so.sam_forward(cfg, w, x)
    err = (o["out"] - out).abs().max().item()
    print(f"sam_hf_{tag}: out {tuple(out.shape)} max|oracle-hf| = {err:.3e}")
    assert err < 1e-4, err


def gen_e4m3fn_table():
    """(iv) fp8 tables: the 256-entry e4m3fn decode table and a quantise/dequantise vector, both from
    torch.float8_e4m3fn casts on the CPU (the reference has no fp8 path; this pins oracle/mx_oracle.py)."""
    from oracle import mx_oracle as mx

    tab = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(7, 96, generator=g) * torch.logspace(-6, 5, 7)[:, None]
    x[2, 32:64] = 0
    q, e = mx.mx_quantize(x)
    d = mx.mx_dequantize(q, e)
    np.savez(os.path.join(HERE, "e4m3fn_table.npz"), decode=tab, x=x.numpy(), payload=q.view(torch.uint8).numpy(),
             exponent=e.numpy(), dequant=d.numpy())


if __name__ == "__main__":
    torch.set_num_threads(8)
    # (i) tiny, (ii) the reference's configured dims (conf/parameters_models.yaml:4,14-16),
    # (iii) the BASELINE config-1 shape (ViT-Ti dims through models_archs.py, odd head count)
    gen_postln("tiny", 64, 1, 2, 128, 2, 5, wseed=11, xseed=3, wscale=0.08)
    gen_postln("refconf", 256, 4, 2, 1024, 2, 50, wseed=12, xseed=4, wscale=0.05)
    gen_postln("cfg1", 192, 3, 12, 768, 8, 196, wseed=13, xseed=5, wscale=0.05)
    gen_vit_hf("tiny", 32, 8, 64, 1, 2, 128, 2, wseed=21, xseed=6)
    gen_vit_hf("p16", 64, 16, 128, 2, 3, 512, 2, wseed=22, xseed=7)
    gen_dinov2_hf("tiny", 28, 14, 64, 1, 2, 2, wseed=31, xseed=8)
    # SAM geometry in miniature: 10x10 grid, window 4 (padded to 12 -> 9 windows), one global block
    gen_sam_hf("tiny", 160, 16, 64, 1, 3, 128, 4, (1,), 32, 2, wseed=41, xseed=9)
    gen_sam_hf("w7", 224, 16, 128, 2, 2, 256, 7, (1,), 64, 1, wseed=42, xseed=10)
    gen_e4m3fn_table()
