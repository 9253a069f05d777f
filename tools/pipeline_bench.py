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
    ap.add_argument("--augment", action="store_true",
                    help="time the whole per-patient augmentation loop (3 flips x 4 angles, GPU rotations) instead")
    a = ap.parse_args()
    from oracle import sam_oracle as so  # weight generator only
    model = vdr.load_model("medsam", weights=so.make_weights(so.SAM_VIT_B, seed=1))
    rng = np.random.default_rng(0)
    img = rng.random((a.side, a.side, a.slices)).astype(np.float32)
    mask = np.zeros((a.side, a.side, a.slices), dtype=bool)
    c = a.side // 2
    mask[c - 20:c + 25, c - 30:c + 22, a.slices // 4: 3 * a.slices // 4] = True
    if a.augment:
        img64 = img.astype(np.float64)  # what apply_window_ct hands over
        pipeline.extract_patient_features(model, img64[:, :, :8], mask[:, :, a.slices // 4:a.slices // 4 + 8], "warm", 0,
                                          "x_dataset", "ct", np.ones(3), flips=(None,), angles=(0, 45))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feats, masks, df = pipeline.extract_patient_features(model, img64, mask, "P", 0, "x_dataset", "ct", np.ones(3))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"augmentation loop, {a.side}x{a.side}x{a.slices} float64 volume, 3 flips x 4 angles: {dt:.2f} s for "
              f"{len(feats)} feature maps = {len(feats) / dt:.1f} slices/s end to end (upload once, 9 GPU rotations of "
              f"image + mask, resize, encoder, ROI crop, D2H)", flush=True)
        from scipy.ndimage import rotate
        t0 = time.perf_counter()
        rotate(img64[:, :, :4], 45, axes=(0, 1), reshape=False, mode="nearest")
        rotate(mask[:, :, :4], 45, axes=(0, 1), reshape=False, mode="nearest")
        ds = (time.perf_counter() - t0) / 4 * a.slices * 9
        print(f"the same 9 rotations of image + mask with SciPy on one host core (4 slices timed, scaled): {ds:.1f} s")
        return
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
