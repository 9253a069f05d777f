#!/usr/bin/env python3
"""Interleaved A/B of the engine's micro-batch / internal-stream configuration on the whole forward, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch, vdr
from oracle import vit_oracle as vo
cfg = vo.CONFIGS["vit_base16_224"]
w = vo.make_weights(cfg, seed=1)
x = torch.rand(256, 3, 224, 224).to(torch.bfloat16).cuda()
out = torch.empty(256, 768, dtype=torch.float32, device="cuda")
settings = [(0, 0), (128, 2), (64, 2), (64, 4), (128, 1), (86, 3)]
models = [vdr.load_model("vit_base16_224", weights=w, micro_batch=mb, streams=st) for mb, st in settings]
times = [[] for _ in settings]
for rnd in range(10):
    for i, m in enumerate(models):
        m.engine.forward_into(x, out, vdr.OUT_CLS)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.engine.forward_into(x, out, vdr.OUT_CLS)
        e1.record(); torch.cuda.synchronize()
        if rnd: times[i].append(e0.elapsed_time(e1) / 5)
for (mb, st), t in zip(settings, times):
    t = sorted(t); print(f"micro_batch {mb:3d} streams {st}: median {t[len(t)//2]:7.3f} ms/step -> {256 / t[len(t)//2] * 1e3:8.1f} img/s", flush=True)
