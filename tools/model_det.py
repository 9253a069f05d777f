#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch
import vdr
from oracle import vit_oracle as vo
cfg = vo.CONFIGS["vit_base16_224"]
w = vo.make_weights(cfg, seed=1)
m = vdr.load_model("vit_base16_224", weights=w)
g = torch.Generator().manual_seed(0)
x = torch.rand(256, 3, 224, 224, generator=g).to(torch.bfloat16).cuda()
for mode, name in ((vdr.OUT_PATCH_EMBED, "patch_embed"), (vdr.OUT_CLS, "cls"), (vdr.OUT_TOKENS, "tokens")):
    a = m.engine.forward(x, mode)
    b = m.engine.forward(x, mode)
    perm = torch.randperm(256, generator=g).cuda()
    c = m.engine.forward(x[perm].contiguous(), mode)
    ne = (c != a[perm])
    rows = ne.reshape(256, -1).any(dim=1).nonzero().flatten()
    print(f"{name}: repeat-equal {bool(torch.equal(a, b))}; perm-equal {bool(torch.equal(c, a[perm]))}; differing images {rows.numel()} e.g. {rows[:8].tolist()} maxdiff {(c.float()-a[perm].float()).abs().max().item():.3e}", flush=True)
