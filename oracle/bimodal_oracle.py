"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the bimodal Stage-C classifier:
`TransformerNoduleBimodalClassifier.forward` (reference src/models_archs.py:38-124) in eval mode — two post-LN
token encoders (CT, PET), cross attention between the two token sequences (`CrossAttentionLayer` :174-183 =
nn.MultiheadAttention, no key padding mask) of which only the CLS query row is consumed, and three `MLPLayer` heads
plus the fusion projection (:186-200).  fp32 torch restatement built on vit_oracle; pinned by
tests/golden/bimodal_*.npz, which hold the outputs of the REFERENCE's own class on seeded weights
(tests/golden/make_golden_bimodal.py imports models_archs from /root/reference)."""
import torch

from . import vit_oracle as vo


def state_dict_shapes(dim, ffn_ct, ffn_pet, layers_ct, layers_pet, num_classes):
    """Key -> shape of TransformerNoduleBimodalClassifier.state_dict() (printed from a live instance)."""
    s = {}
    for m, ffn, layers in (("ct", ffn_ct, layers_ct), ("pet", ffn_pet, layers_pet)):
        for i in range(layers):
            p = f"transformer_encoder_{m}.layers.{i}."
            s[p + "self_attn.in_proj_weight"] = (3 * dim, dim)
            s[p + "self_attn.in_proj_bias"] = (3 * dim,)
            s[p + "self_attn.out_proj.weight"] = (dim, dim)
            s[p + "self_attn.out_proj.bias"] = (dim,)
            s[p + "linear1.weight"] = (ffn, dim)
            s[p + "linear1.bias"] = (ffn,)
            s[p + "linear2.weight"] = (dim, ffn)
            s[p + "linear2.bias"] = (dim,)
            for n in ("norm1", "norm2"):
                s[p + n + ".weight"] = (dim,)
                s[p + n + ".bias"] = (dim,)
        s[f"norm_{m}.weight"] = (dim,)
        s[f"norm_{m}.bias"] = (dim,)
        s[f"cls_token_{m}"] = (1, 1, dim)
        s[f"cross_attention_{m}.multihead_attn.in_proj_weight"] = (3 * dim, dim)
        s[f"cross_attention_{m}.multihead_attn.in_proj_bias"] = (3 * dim,)
        s[f"cross_attention_{m}.multihead_attn.out_proj.weight"] = (dim, dim)
        s[f"cross_attention_{m}.multihead_attn.out_proj.bias"] = (dim,)
    for head, (i, h, o) in (("classifier_ct", (dim, 2 * dim, num_classes)), ("classifier_pet", (dim, 2 * dim, num_classes)),
                            ("projection_petct", (2 * dim, dim, dim)), ("classifier_petct", (dim, 2 * dim, num_classes))):
        s[head + ".dense1.weight"] = (h, i)
        s[head + ".dense1.bias"] = (h,)
        s[head + ".dense2.weight"] = (o, h)
        s[head + ".dense2.bias"] = (o,)
    return s


def make_state_dict(dim, ffn_ct, ffn_pet, layers_ct, layers_pet, num_classes, seed=1, scale=0.06):
    """Seeded synthetic weights in the reference's key naming (sorted key order, one generator)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in sorted(state_dict_shapes(dim, ffn_ct, ffn_pet, layers_ct, layers_pet, num_classes).items()):
        r = torch.randn(shp, generator=g)
        if ".norm" in k or k.startswith("norm_"):
            sd[k] = (1.0 + 0.1 * r) if k.endswith("weight") else 0.1 * r
        elif k.startswith("cls_token"):
            sd[k] = 0.5 * r
        elif k.endswith("bias"):
            sd[k] = 0.05 * r
        else:
            sd[k] = scale * r
    return sd


def encoder_weights(sd, modality, layers):
    """Keys of one modality's encoder -> vit_oracle's canonical names."""
    sub = {"cls_token": sd[f"cls_token_{modality}"], "norm.weight": sd[f"norm_{modality}.weight"],
           "norm.bias": sd[f"norm_{modality}.bias"]}
    pre = f"transformer_encoder_{modality}."
    for k, v in sd.items():
        if k.startswith(pre):
            sub["transformer_encoder." + k[len(pre):]] = v
    return vo.from_torch_encoder_state_dict(sub, layers)


def cross_attention_cls(sd, modality, heads, xq, xkv):
    """Row 0 of CrossAttentionLayer(query=xq, key=xkv, value=xkv): [B, D]."""
    p = f"cross_attention_{modality}.multihead_attn."
    W, b = sd[p + "in_proj_weight"].float(), sd[p + "in_proj_bias"].float()
    D = W.shape[1]
    q = xq[:, 0, :] @ W[:D].t() + b[:D]                       # [B, D]
    k = xkv @ W[D:2 * D].t() + b[D:2 * D]                     # [B, S, D]
    v = xkv @ W[2 * D:].t() + b[2 * D:]
    B, S, dh = xkv.shape[0], xkv.shape[1], D // heads
    qh = q.reshape(B, heads, 1, dh)
    kh = k.reshape(B, S, heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(B, S, heads, dh).permute(0, 2, 1, 3)
    a = torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, dim=-1) @ vh   # [B, H, 1, dh]
    o = a.permute(0, 2, 1, 3).reshape(B, D)
    return o @ sd[p + "out_proj.weight"].float().t() + sd[p + "out_proj.bias"].float()


def _head(sd, name, x):
    return vo.mlp_head(x, sd[name + ".dense1.weight"].float(), sd[name + ".dense1.bias"].float(),
                       sd[name + ".dense2.weight"].float(), sd[name + ".dense2.bias"].float())


def forward(sd, dim, ffn_ct, ffn_pet, heads_ct, heads_pet, layers_ct, layers_pet, x_ct=None, x_pet=None):
    """(logits_petct, petct_cls_token, logits_ct, logits_pet) as models_archs.py:76-124."""
    assert x_ct is not None or x_pet is not None
    t_ct = t_pet = None
    if x_ct is not None:
        t_ct = vo.forward_tokens(vo.postln_cfg(dim, heads_ct, layers_ct, ffn_ct), encoder_weights(sd, "ct", layers_ct), x_ct)["tokens"]
    if x_pet is not None:
        t_pet = vo.forward_tokens(vo.postln_cfg(dim, heads_pet, layers_pet, ffn_pet), encoder_weights(sd, "pet", layers_pet), x_pet)["tokens"]
    if t_ct is not None and t_pet is not None:
        ct_cls = cross_attention_cls(sd, "ct", heads_ct, t_ct, t_pet)
        pet_cls = cross_attention_cls(sd, "pet", heads_ct, t_pet, t_ct)  # the reference builds both with num_heads_ct
        logits_ct, logits_pet = _head(sd, "classifier_ct", ct_cls), _head(sd, "classifier_pet", pet_cls)
        fused = _head(sd, "projection_petct", torch.cat([ct_cls, pet_cls], dim=1))
        return _head(sd, "classifier_petct", fused), fused, logits_ct, logits_pet
    if t_ct is not None:
        cls = t_ct[:, 0, :]
        lg = _head(sd, "classifier_ct", cls)
        return lg, cls, lg, lg
    cls = t_pet[:, 0, :]
    lg = _head(sd, "classifier_pet", cls)
    return lg, cls, lg, lg
