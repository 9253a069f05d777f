"""CPU restatement of the numpy pre/post-processing either side of the encoder.

TEST INFRASTRUCTURE ONLY (never imported by the product package).

  crop_image / extract_coords / extract_roi      src/visualization_utils.py:93-125
  hu_to_rgb_vectorized                            src/visualization_utils.py:128-186
  windowing_ct / apply_window_ct                  src/tfds_dense_descriptor.py:204-237, 287-302
  prepare_image (gray2rgb + resize, CHW)          src/tfds_dense_descriptor.py:30-48 -> third-party
                                                  skimage.transform.resize (0.18.3 in the authoring container):
                                                  order 1, mode 'reflect' (numpy.pad sense: no edge repeat),
                                                  anti_aliasing Gaussian (sigma = (scale-1)/2, truncate 4,
                                                  scipy 'mirror' boundary) when down-scaling, pixel centres at +0.5

Pinned by tests/golden/prep_*.npz (made by tests/golden/make_golden_prep.py from the reference's own functions
and the same skimage calls).
"""
from __future__ import annotations

import numpy as np


# ---- ROI maths (pure integer / slicing) ----------------------------------------------------------------
def crop_image(img, xmin, ymin, xmax, ymax):
    h, w = img.shape[0:2]
    ymin, ymax = [max(0, min(v, h)) for v in (ymin, ymax)]
    xmin, xmax = [max(0, min(v, w)) for v in (xmin, xmax)]
    return img[ymin:ymax, xmin:xmax]


def extract_coords(mask, margin):
    """NB the reference's asymmetric margins (ymin - m, xmin + m, ymax - m, xmax + m) are part of the contract."""
    ys, xs = np.where(mask)
    ymin, xmin = int(ys.min()) - margin, int(xs.min()) + margin
    ymax, xmax = int(ys.max()) - margin, int(xs.max()) + margin
    h = max(ymax - ymin, margin)
    w = max(xmax - xmin, margin)
    return xmin, ymin, xmin + w, ymin + h


def roi_box(img_hw, mask, margin=1):
    """the (xmin, ymin, xmax, ymax) extract_roi crops an array of spatial shape img_hw to (before clamping)"""
    xmin, ymin, xmax, ymax = extract_coords(mask, margin)
    if tuple(img_hw) != tuple(mask.shape[0:2]):
        h = img_hw[0] / mask.shape[0]
        w = img_hw[1] / mask.shape[1]
        xmin, ymin, xmax, ymax = [int(v) for v in (xmin * w, ymin * h, xmax * w, ymax * h)]
        hh = max(ymax - ymin, margin)
        ww = max(xmax - xmin, margin)
        xmax, ymax = xmin + ww, ymin + hh
    return xmin, ymin, xmax, ymax


def extract_roi(img, mask, margin=1):
    return crop_image(img, *roi_box(img.shape[0:2], mask, margin))


# ---- intensity maps -------------------------------------------------------------------------------------
def apply_window_ct(ct, width, level):
    lo, hi = level - width / 2, level + width / 2
    return np.clip((ct - lo) / (hi - lo), 0, 1)


_HU_SEG = [  # (lo, hi, include_lo, include_hi, colour_a, colour_b, interp_min, interp_max); None = constant colour_a
    (-np.inf, -1000, True, True, (0, 0, 0), None, 0, 0),
    (-1000, -600, False, False, (0, 0, 0), (194, 105, 82), -1000, -600),
    (-600, -400, True, True, (194, 105, 82), None, 0, 0),
    (-400, -100, False, False, (194, 105, 82), (194, 166, 115), -400, -100),
    (-100, -60, True, True, (194, 166, 115), None, 0, 0),
    (-60, 40, False, False, (194, 166, 115), (102, 0, 0), -60, 40),
    (40, 80, True, True, (102, 0, 0), (153, 0, 0), 80, 400),   # the reference interpolates this band over [80, 400]
    (80, 400, False, False, (153, 0, 0), (255, 255, 255), 80, 400),
    (400, np.inf, True, True, (255, 255, 255), None, 0, 0),
]


def hu_to_rgb(hu):
    hu = np.asarray(hu)
    out = np.zeros(hu.shape + (3,), dtype=np.int64)
    for lo, hi, inc_lo, inc_hi, ca, cb, mn, mx in _HU_SEG:
        m = (hu >= lo if inc_lo else hu > lo) & (hu <= hi if inc_hi else hu < hi)
        if cb is None:
            out[m] = ca
        else:
            r = (hu[m] - mn) / (mx - mn)
            out[m] = np.array(ca) * (1 - r[..., None]) + np.array(cb) * r[..., None]  # float -> int: truncation
    return out.astype(np.uint8)


# ---- skimage.transform.resize(order=1, mode='reflect', anti_aliasing=True) ------------------------------
def _mirror(i, n):
    """numpy.pad 'reflect' / scipy 'mirror' index map (no edge repeat)."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.abs(i) % p
    return np.where(i >= n, p - i, i)


def _gauss1d(x, sigma, axis):
    if sigma <= 0:
        return x
    r = int(4.0 * sigma + 0.5)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    n = x.shape[axis]
    idx = _mirror(np.arange(n)[:, None] + np.arange(-r, r + 1)[None, :], n)  # [n, 2r+1]
    xm = np.moveaxis(x, axis, -1)
    y = (xm[..., idx] * k).sum(-1)
    return np.moveaxis(y, -1, axis)


def resize_hwc(img, out_h, out_w):
    """img [h, w, C] float -> [out_h, out_w, C]; the arithmetic runs in float64, the result keeps img's dtype."""
    dt = img.dtype
    x = img.astype(np.float64)
    h, w = x.shape[:2]
    sy, sx = h / out_h, w / out_w
    # anti-aliasing prefilter (each axis result stored back in the image dtype, as ndimage does)
    if sy > 1 or sx > 1:
        x = _gauss1d(x, max(0.0, (sy - 1) / 2), 0).astype(dt).astype(np.float64)
        x = _gauss1d(x, max(0.0, (sx - 1) / 2), 1).astype(dt).astype(np.float64)
    r = (np.arange(out_h) + 0.5) * sy - 0.5
    c = (np.arange(out_w) + 0.5) * sx - 0.5
    r0, c0 = np.floor(r), np.floor(c)
    dr, dc = (r - r0)[:, None, None], (c - c0)[None, :, None]
    r0i, r1i = _mirror(r0.astype(np.int64), h), _mirror(np.ceil(r).astype(np.int64), h)
    c0i, c1i = _mirror(c0.astype(np.int64), w), _mirror(np.ceil(c).astype(np.int64), w)
    top = (1 - dc) * x[r0i][:, c0i] + dc * x[r0i][:, c1i]
    bot = (1 - dc) * x[r1i][:, c0i] + dc * x[r1i][:, c1i]
    y = (1 - dr) * top + dr * bot
    lo, hi = img.min(), img.max()  # clip=True
    return np.clip(y, lo, hi).astype(dt)


def prepare_image(img, side=None):
    """(h, w) gray -> [3, 1024, 1024]; (h, w, 3) -> [3, 896, 896] (side overrides), CHW, image dtype kept."""
    if img.ndim < 3:
        x = np.repeat(img[..., None], 3, axis=2)
        side = side or 1024
    else:
        x = img
        side = side or 896
    return resize_hwc(x, side, side).transpose(2, 0, 1)


# ---- Stage-C input builder: train_models.py:30-44 (positional_encoding_3d) and :143-182 (_get_features) ------------
def resize_mask_nearest(mask, out_hw):
    """skimage.transform.resize(mask, out_hw, order=0) of a boolean mask: nearest sample of the pixel-centre
    mapping x = (o + 0.5) * in / out - 0.5 (pinned by skimage 0.18.3 values, tests/golden/sequence_cases.npz)."""
    mask = np.asarray(mask)
    H, W = mask.shape
    oh, ow = out_hw
    yi = np.clip(np.floor((np.arange(oh) + 0.5) * H / oh - 0.5 + 0.5).astype(np.int64), 0, H - 1)
    xi = np.clip(np.floor((np.arange(ow) + 0.5) * W / ow - 0.5 + 0.5).astype(np.int64), 0, W - 1)
    return mask[yi][:, xi].astype(bool)


def positional_encoding_3d(x, y, z, D, scale=10000):
    x, y, z = np.asarray(x), np.asarray(y), np.asarray(z)
    enc = np.zeros((x.shape[0], D))
    for i in range(D // 6):
        e = scale ** (6 * i / D)
        for off, v in ((0, x), (D // 3, y), (2 * D // 3, z)):
            enc[:, 2 * i + off] = np.sin(v / e)
            enc[:, 2 * i + 1 + off] = np.cos(v / e)
    return enc


def voxel_coordinates(h, w, S, orig_hw, spatial_res, noise):
    """(x, y, z) float64 [h*w*S] of every flat position of the (h, w, S) feature volume, exactly as :162-173 computes
    them: np.meshgrid's default 'xy' indexing returns (w, h, S)-shaped grids that the reference flattens against the
    (h, w, S)-ordered features, so for non-square maps x / y follow that flat order, not the voxel's own row / column
    (consumers were trained on it)."""
    x, y, z = np.meshgrid(np.arange(0, h), np.arange(0, w), np.arange(0, S))
    x = (x.flatten() / w).flatten() * orig_hw[1] * spatial_res[0]
    y = (y.flatten() / h).flatten() * orig_hw[0] * spatial_res[1]
    z = (z.flatten()).flatten() * spatial_res[2]
    return x - x.mean() + noise[0], y - y.mean() + noise[1], z - z.mean() + noise[2]


def masked_voxel_sequence(slice_features, slice_masks, spatial_res, noise=(0.0, 0.0, 0.0)):
    """The 'transformer' branch of _get_features: (seq [n, D] float64, keep [h*w*S] bool)."""
    feats = np.stack([np.asarray(f) for f in slice_features], axis=0)               # (S, h, w, D)
    S, h, w, D = feats.shape
    masks = np.stack([resize_mask_nearest(m, (h, w)) for m in slice_masks], axis=0)  # (S, h, w)
    keep = np.transpose(masks, (1, 2, 0)).reshape(-1)
    vol = np.transpose(feats, (1, 2, 0, 3)).reshape(-1, D)
    x, y, z = voxel_coordinates(h, w, S, np.asarray(slice_masks[-1]).shape[0:2], spatial_res, noise)
    pe = positional_encoding_3d(x[keep], y[keep], z[keep], D)
    return vol[keep, :] + pe / 4, keep
