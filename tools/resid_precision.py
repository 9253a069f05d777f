#!/usr/bin/env python3
"""What would an fp32 residual stream buy?  (round-3 review, item 6: SURVEY 8d states rel-L2 <= 1e-2 for the bf16 path
against the fp32 oracle; the tests gate at 4e-3 + 3e-3 sqrt(L).)  CPU emulation with oracle/vit_oracle.py: every rounding
point of the HIP path as it is (emulate_bf16=True), once with the residual stream rounded to bf16 after each sub-block as
the library stores it (buffer x) and once with it kept in fp32 (vo.RESID_FP32).  Output kept in
profiles/r04_resid_precision.txt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vit_oracle as vo  # noqa: E402

torch.set_num_threads(min(8, os.cpu_count() or 1))


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def mincos(a, b):
    return torch.nn.functional.cosine_similarity(a.double().reshape(-1, a.shape[-1]), b.double().reshape(-1, b.shape[-1]), dim=-1).min().item()


for name, n_img in (("vit_base16_224", 4), ("vit_large14_336", 2), ("dinov2_giant14_224", 2)):
    cfg = vo.CONFIGS[name]
    w = vo.make_weights(cfg, seed=1)
    x = vo.make_images(cfg, n_img, seed=3)
    ref = vo.forward_images(cfg, w, x)
    row = f"{name} L = {cfg.layers} ({n_img} images):"
    for fp32 in (False, True):
        vo.RESID_FP32 = fp32
        try:
            out = vo.forward_images(cfg, w, x, emulate_bf16=True)
        finally:
            vo.RESID_FP32 = False
        row += (f"  residual {'fp32' if fp32 else 'bf16'}: CLS relL2 {rel(out['cls'], ref['cls']):.3e} cos {mincos(out['cls'], ref['cls']):.6f},"
                f" dense relL2 {rel(out['dense'], ref['dense']):.3e} cos {mincos(out['dense'], ref['dense']):.6f} |")
    print(row, flush=True)
