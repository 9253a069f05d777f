"""Golden vectors for the bimodal Stage-C classifier: outputs of the REFERENCE's own
`TransformerNoduleBimodalClassifier` (src/models_archs.py:38-124; imports with torch alone) in eval mode on seeded
weights and inputs.      python tests/golden/make_golden_bimodal.py
Weights are regenerated from the seed by oracle.bimodal_oracle.make_state_dict (probe values are stored to catch a
generator change); only inputs, probes and expected outputs are stored (tests/golden/bimodal_*.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference/src")
from oracle import bimodal_oracle as bo  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402


def gen(tag, dim, ratio_ct, ratio_pet, heads_ct, heads_pet, layers_ct, layers_pet, classes, batch, s_ct, s_pet, seed):
    import models_archs  # the reference
    ref = models_archs.TransformerNoduleBimodalClassifier(dim, ratio_ct, ratio_pet, heads_ct, heads_pet, layers_ct,
                                                          layers_pet, classes)
    ffn_ct, ffn_pet = int(ratio_ct * dim), int(ratio_pet * dim)
    sd = bo.make_state_dict(dim, ffn_ct, ffn_pet, layers_ct, layers_pet, classes, seed=seed)
    have = ref.state_dict()
    assert set(have) == set(sd), set(have) ^ set(sd)
    for k in have:
        assert have[k].shape == sd[k].shape, k
    ref.load_state_dict(sd)
    ref.eval()
    x_ct = vo.make_tokens(batch, s_ct, dim, seed=seed + 1)
    x_pet = vo.make_tokens(batch, s_pet, dim, seed=seed + 2)
    out = dict(dim=dim, ratio_ct=ratio_ct, ratio_pet=ratio_pet, heads_ct=heads_ct, heads_pet=heads_pet,
               layers_ct=layers_ct, layers_pet=layers_pet, classes=classes, seed=seed, x_ct=x_ct.numpy(), x_pet=x_pet.numpy(),
               w_probe=sd["cross_attention_ct.multihead_attn.in_proj_weight"][:2, :8].numpy())
    with torch.no_grad():
        for mode, (a, b) in (("both", (x_ct, x_pet)), ("ct", (x_ct, None)), ("pet", (None, x_pet))):
            r = ref(a, b)
            o = bo.forward(sd, dim, ffn_ct, ffn_pet, heads_ct, heads_pet, layers_ct, layers_pet, a, b)
            for name, rv, ov in zip(("logits_petct", "cls_petct", "logits_ct", "logits_pet"), r, o):
                err = (rv - ov).abs().max().item()
                assert err < 2e-5, (mode, name, err)
                out[f"{mode}_{name}"] = rv.numpy()
    np.savez_compressed(os.path.join(HERE, f"bimodal_{tag}.npz"), **out)
    print(f"bimodal_{tag}: ok", {k: v.shape for k, v in out.items() if k.startswith("both_")})


if __name__ == "__main__":
    gen("tiny", 128, 2, 1.5, 2, 2, 2, 1, 2, batch=2, s_ct=9, s_pet=6, seed=31)
    gen("refdim", 256, 4, 4, 4, 4, 2, 2, 2, batch=3, s_ct=40, s_pet=23, seed=32)
