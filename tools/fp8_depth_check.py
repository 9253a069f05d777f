#!/usr/bin/env python3
"""fp8 path at FULL depth against the fp32 oracle (test-suite cases cut the depth for time): per-row cosine and
relative L2 of the CLS / dense outputs for ViT-B/16 (12 blocks) and DINOv2 ViT-g/14 (40 blocks), bf16 next to it."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch
import vdr
from oracle import vit_oracle as vo   # checker

torch.set_num_threads(16)
for name, B in (("vit_base16_224", 4), ("dinov2_giant14_224", 2)):
    cfg = vo.CONFIGS[name]
    w = vo.make_weights(cfg, seed=1)
    x = vo.make_images(cfg, B, seed=3)
    ref = vo.forward_images(cfg, w, x)
    for fp8 in (0, 1, 2):
        m = vdr.load_model(name, weights=w, fp8=fp8)
        for mode, key in ((vdr.OUT_CLS, "cls"), (vdr.OUT_DENSE, "dense")):
            got = m.engine.forward(x.cuda(), mode, torch.float32).float().cpu().reshape(ref[key].shape)
            r = ref[key]
            rel = float((got - r).norm() / r.norm())
            g2, r2 = got.reshape(-1, got.shape[-1]), r.reshape(-1, r.shape[-1])
            cos = float(torch.nn.functional.cosine_similarity(g2, r2, dim=-1).min())
            print(f"{name:20s} L={cfg.layers:2d} {('fp8=%d' % fp8) if fp8 else 'bf16 '} {key:5s}: rel L2 {rel:.3e}  min row cosine {cos:.5f}", flush=True)
        del m

# the reference's default backbone: SAM ViT-B image encoder at 1024^2, one slice
from oracle import sam_oracle as so
w = so.make_weights(so.SAM_VIT_B, seed=1)
x = so.make_images(so.SAM_VIT_B, 1, seed=3)
ref = so.sam_forward(so.SAM_VIT_B, w, x)["out"].permute(0, 2, 3, 1)
for fp8 in (0, 1):
    m = vdr.load_model("medsam", weights=w, fp8=fp8)
    got = m.engine.forward(x.cuda(), vdr.OUT_ENCODER, torch.float32).float().cpu()
    rel = float((got - ref).norm() / ref.norm())
    cos = float(torch.nn.functional.cosine_similarity(got.reshape(-1, 256), ref.reshape(-1, 256), dim=-1).min())
    print(f"medsam (SAM ViT-B 1024^2) L=12 {('fp8=%d' % fp8) if fp8 else 'bf16 '} neck : rel L2 {rel:.3e}  min pixel cosine {cos:.5f}", flush=True)
    del m
