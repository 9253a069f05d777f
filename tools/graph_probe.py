#!/usr/bin/env python3
"""Does a whole forward replay from a HIP graph (captured through torch.cuda.CUDAGraph on the caller's stream), and
what does it buy at launch-bound sizes?   python tools/graph_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch
import vdr
from oracle import vit_oracle as vo, sam_oracle as so   # weight generators


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, B, mode in (("medsam", 1, vdr.OUT_ENCODER), ("vit_tiny16_224", 8, vdr.OUT_CLS), ("vit_base16_224", 16, vdr.OUT_CLS)):
    if name == "medsam":
        m = vdr.load_model(name, weights=so.make_weights(so.SAM_VIT_B, seed=1)); side = 1024; dt = torch.float32
    else:
        cfg = vo.CONFIGS[name]; m = vdr.load_model(name, weights=vo.make_weights(cfg, seed=1)); side = cfg.img; dt = torch.bfloat16
    x = torch.rand(B, 3, side, side, device="cuda").to(dt)
    out = m.engine.forward(x, mode, torch.float32)
    eager = bench(lambda: m.engine.forward_into(x, out, mode))
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    try:
        with torch.cuda.stream(s):
            m.engine.forward_into(x, out, mode)   # warm on the capture stream
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                m.engine.forward_into(x, out, mode)
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(out, ref)
        graph = bench(g.replay)
        print(f"{name:16s} B={B:2d}: eager {eager:.3f} ms, graph replay {graph:.3f} ms, identical output: {same}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: capture failed: {e!r}"[:300], flush=True)
