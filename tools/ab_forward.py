#!/usr/bin/env python3
"""Interleaved A/B of tuning knobs on the WHOLE forward in one process (guide rule 24): a tuning build (make TUNING=1)
re-reads the VDR_* environment on every launch, so settings can alternate between forwards on the same device.
   python tools/ab_forward.py "VDR_GEMM_GN=0" "VDR_GEMM_GN=4" "" --rounds 9 --steps 5
Each argument is a space-separated list of NAME=VALUE (empty string = defaults).  Prints median ms/step per setting."""
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
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--model", default="vit_base16_224")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--fp8", type=int, default=0)
    a = ap.parse_args()
    assert vdr.load().vdr_tuning_build(), "needs a tuning build: make -C vit-deep-radiomics_amd/csrc TUNING=1"
    cfg = vo.CONFIGS[a.model]
    model = vdr.load_model(a.model, weights=vo.make_weights(cfg, seed=1), fp8=a.fp8)
    x = torch.rand(a.batch, 3, cfg.img, cfg.img).to(torch.bfloat16).cuda()
    out = torch.empty(a.batch, cfg.dim, dtype=torch.float32, device="cuda")
    knobs = sorted({kv.split("=")[0] for s in a.settings for kv in s.split() if kv})

    def apply(s):
        for k in knobs:
            os.environ.pop(k, None)
        for kv in s.split():
            k, v = kv.split("=")
            os.environ[k] = v

    times = [[] for _ in a.settings]
    for rnd in range(a.rounds + 1):
        for i, s in enumerate(a.settings):
            apply(s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            model.engine.forward_into(x, out, vdr.OUT_CLS)
            e0.record()
            for _ in range(a.steps):
                model.engine.forward_into(x, out, vdr.OUT_CLS)
            e1.record()
            torch.cuda.synchronize()
            if rnd:  # round 0 warms up
                times[i].append(e0.elapsed_time(e1) / a.steps)
    for s, t in zip(a.settings, times):
        t = sorted(t)
        print(f"{s or '(defaults)':40s}: median {t[len(t) // 2]:7.3f} ms/step  min {t[0]:7.3f}  -> {a.batch / t[len(t) // 2] * 1e3:8.1f} img/s", flush=True)


if __name__ == "__main__":
    main()
