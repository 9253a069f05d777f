#!/usr/bin/env python3
"""vdr_config.stream_gemm on / off on the whole forward: outputs must be bitwise equal (ragged and full batches, LayerNorm
fold on and off); interleaved timing of the two engines in one process."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch  # noqa: E402

import vdr  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402  (weight / image generators only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_base16_224")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    cfg = vo.CONFIGS[a.model]
    w = vo.make_weights(cfg, seed=1)
    ok = True
    models = {}
    for fold in (True, False):
        base = vdr.load_model(a.model, weights=w, ln_fold=fold)
        strm = vdr.load_model(a.model, weights=w, ln_fold=fold, stream_gemm=True)
        if fold:
            models = {"ring4": base, "stream": strm}
        for B in (100, a.batch):
            x = torch.rand(B, 3, cfg.img, cfg.img).to(torch.bfloat16).cuda()
            for mode in (vdr.OUT_CLS, vdr.OUT_DENSE):
                ya = base.engine.forward(x, mode)
                yb = strm.engine.forward(x, mode)
                torch.cuda.synchronize()
                same = torch.equal(ya, yb)
                print(f"ln_fold={fold} B={B} out_mode={mode}: bitwise equal = {same}", flush=True)
                ok &= same
    print("ALL EQUAL" if ok else "MISMATCH", flush=True)
    x = torch.rand(a.batch, 3, cfg.img, cfg.img).to(torch.bfloat16).cuda()
    out = torch.empty(a.batch, cfg.dim, dtype=torch.float32, device="cuda")
    times = {k: [] for k in models}
    for rnd in range(a.rounds + 1):
        for k, m in models.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            m.engine.forward_into(x, out, vdr.OUT_CLS)
            e0.record()
            for _ in range(a.steps):
                m.engine.forward_into(x, out, vdr.OUT_CLS)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) / a.steps)
    for k, t in times.items():
        t = sorted(t)
        print(f"{k:8s}: median {t[len(t) // 2]:7.3f} ms/step  min {t[0]:7.3f}  -> {a.batch / t[len(t) // 2] * 1e3:8.1f} img/s", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
