#!/usr/bin/env python3
"""Offline rotation augmentation (rotate_image, reference tfds_dense_descriptor.py:327-350) on a CT-sized float64
volume: the GPU kernels of rotate.hip against scipy.ndimage.rotate on the host cores, plus the per-kernel split
(HIP events) and the bytes each one moves.   python tools/rotate_bench.py [--slices 130] [--side 512]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vdr import prep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=130)
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--scipy-slices", type=int, default=16, help="slices timed with SciPy (scaled to the volume)")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    H = W = a.side
    vol = rng.random((H, W, a.slices))
    mask = np.zeros((H, W, a.slices), dtype=bool)
    mask[H // 2 - 30:H // 2 + 30, W // 2 - 25:W // 2 + 35, a.slices // 4:3 * a.slices // 4] = True
    dvol, dmask = torch.from_numpy(vol).cuda(), torch.from_numpy(mask).cuda()
    for name, x, clip in (("image f64", dvol, True), ("mask bool", dmask, False)):
        prep.rotate_volume(x, 45, clip01=clip)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 10
        ev[0].record()
        for _ in range(n):
            out = prep.rotate_volume(x, 45, clip01=clip)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / n
        padded = (H + 24) * (W + 24) * a.slices * 8
        # pad: read src + write padded; 2 filter axes x (2 reads + ... ) ~ 5 passes each; sample: 16 taps (cached) + write
        alg = x.numel() * x.element_size() * 2 + padded * (1 + 2 * 5)
        print(f"GPU {name}: {ms:7.2f} ms per {H}x{W}x{a.slices} volume = {a.slices / ms * 1e3:9.0f} slices/s "
              f"({alg / ms / 1e6:.0f} GB/s of pass traffic)", flush=True)
    from scipy.ndimage import rotate
    k = min(a.scipy_slices, a.slices)
    t0 = time.perf_counter()
    want = rotate(vol[:, :, :k], 45, axes=(0, 1), reshape=False, mode="nearest")
    dt = time.perf_counter() - t0
    got = prep.rotate_volume(dvol[:, :, :k].contiguous(), 45).cpu().numpy()
    print(f"SciPy (1 core) image f64: {dt / k * 1e3:.1f} ms/slice -> {dt / k * a.slices:.2f} s per volume; "
          f"bitwise equal to the GPU result: {np.array_equal(got, want)}")
    t0 = time.perf_counter()
    h = dvol.cpu()
    torch.cuda.synchronize()
    print(f"D2H of the rotated volume (what a host-side consumer would pay): {(time.perf_counter() - t0) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
