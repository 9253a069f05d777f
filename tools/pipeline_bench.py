#!/usr/bin/env python3
"""End-to-end slices/s of the batched slice pipeline (vdr.pipeline.generate_features: crop -> GPU prepare_image ->
MedSAM encoder -> ROI crop -> D2H) on a synthetic CT-sized volume, next to the encoder-only rate of bench.py.
   python tools/pipeline_bench.py [--slices 64] [--side 512]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vdr  # noqa: E402
from vdr import pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--side", type=int, default=512)
    a = ap.parse_args()
    from oracle import sam_oracle as so  # weight generator only
    model = vdr.load_model("medsam", weights=so.make_weights(so.SAM_VIT_B, seed=1))
    rng = np.random.default_rng(0)
    img = rng.random((a.side, a.side, a.slices)).astype(np.float32)
    mask = np.zeros((a.side, a.side, a.slices), dtype=bool)
    c = a.side // 2
    mask[c - 20:c + 25, c - 30:c + 22, a.slices // 4: 3 * a.slices // 4] = True
    for mb in (1, 4, 8, 16):
        lo = a.slices // 4
        pipeline.generate_features(model, img[:, :, lo:lo + mb], mask[:, :, lo:lo + mb], max_batch=mb)  # warm-up / workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feats, masks = pipeline.generate_features(model, img, mask, max_batch=mb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"max_batch {mb:2d}: {a.slices / dt:7.1f} slices/s  ({dt * 1e3 / a.slices:.2f} ms/slice)  crop {feats[0].shape}", flush=True)


if __name__ == "__main__":
    main()
