#!/usr/bin/env python3
"""Race / hang hunt: many back-to-back forwards of every path (ViT-B bf16 + fp8, ViT-L chunked attention, MedSAM bf16
+ fp8, varlen classifier) must reproduce their first result bitwise.   python tools/stress_det.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch
import vdr
from oracle import vit_oracle as vo, sam_oracle as so   # weight generators only

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(0)
cases = []
for name, B, fp8, mode in (("vit_base16_224", 256, 0, vdr.OUT_CLS), ("vit_base16_224", 256, 2, vdr.OUT_CLS),
                           ("vit_large14_336", 16, 0, vdr.OUT_DENSE), ("dinov2_giant14_224", 8, 1, vdr.OUT_CLS)):
    cfg = vo.CONFIGS[name]
    m = vdr.load_model(name, weights=vo.make_weights(cfg, seed=1), fp8=fp8)
    x = torch.rand(B, 3, cfg.img, cfg.img, generator=g).to(torch.bfloat16).cuda()
    cases.append((f"{name} B={B} fp8={fp8}", m, lambda m=m, x=x, mode=mode: m.engine.forward(x, mode)))
for fp8 in (0, 1):
    m = vdr.load_model("medsam", weights=so.make_weights(so.SAM_VIT_B, seed=1), fp8=fp8)
    x = torch.rand(2, 3, 1024, 1024, generator=g).cuda()
    cases.append((f"medsam B=2 fp8={fp8}", m, lambda m=m, x=x: m.engine.forward(x, vdr.OUT_ENCODER)))
for label, m, fn in cases:
    ref = fn().clone()
    torch.cuda.synchronize()
    t0 = time.time()
    bad = 0
    for i in range(iters):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"{label:40s}: {iters} runs, {bad} differing, finite {bool(torch.isfinite(ref.float()).all())}, {time.time() - t0:.1f} s", flush=True)
