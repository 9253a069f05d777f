"""GPU counterparts of the reference's per-slice numpy / skimage pre-processing.

Same names, argument meaning and return contracts as the functions they replace; inputs may be numpy arrays
(uploaded as they are: a 512x512 slice instead of the 12.6 MB resized tensor) or device tensors.

  prepare_image(img)                    src/tfds_dense_descriptor.py:30-48
  prepare_slices(volume, ...)           the same for all slices of an (H, W, S[, C]) volume in ONE launch
  apply_window_ct(ct, width, level)     src/tfds_dense_descriptor.py:287-302 (windowing_ct :204-237)
  hu_to_rgb_vectorized(hu)              src/visualization_utils.py:128-186
  flip_image(image, mask, flip_type)    src/tfds_dense_descriptor.py:305-324 (views; prepare_slices(flip=...) folds
                                        the image flip into the resize gather instead)

rotate_image (scipy.ndimage.rotate, cubic spline, :327-350) is NOT implemented on the GPU: it stays upstream.
There is no CPU fallback: without the HIP library every function raises.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L

_FLIP = {None: 0, "none": 0, "horizontal": 1, "vertical": 2}


def _s(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(x, device, dtypes):
    t = torch.as_tensor(x)
    if t.dtype not in dtypes:
        t = t.to(torch.float32)
    return t.to(device if device is not None else "cuda")


def prepare_slices(volume, side=None, flip=None, out_dtype=torch.float32, device=None) -> torch.Tensor:
    """(H, W, S) gray or (H, W, S, 3) colour volume -> [S, 3, side, side] device tensor, every slice prepared
    exactly as prepare_image does (gray2rgb + skimage resize semantics, CHW); side defaults to the reference's
    1024 (gray -> MedSAM) / 896 (colour -> DINOv2).  The volume is read strided: no transpose, no copy."""
    lib = L.load()
    t = _dev(volume, device, (torch.float32, torch.float64))
    if t.dim() not in (3, 4) or (t.dim() == 4 and t.shape[3] != 3):
        raise ValueError(f"volume must be (H, W, S) or (H, W, S, 3), got {tuple(t.shape)}")
    ch = 1 if t.dim() == 3 else 3
    side = side or (1024 if ch == 1 else 896)
    H, W, S = t.shape[:3]
    st = t.stride()
    out = torch.empty((S, 3, side, side), dtype=out_dtype, device=t.device)
    nb = int(lib.vdr_prepare_scratch_bytes(S, H, W, ch, side))
    scratch = torch.empty(nb, dtype=torch.uint8, device=t.device) if nb else None
    L.check(lib.vdr_op_prepare_image(t.data_ptr(), L.VDR_F32 if t.dtype == torch.float32 else L.VDR_F64, S, H, W, ch,
                                     st[2], st[0], st[1], st[3] if ch == 3 else 0, _FLIP[flip], side, out.data_ptr(),
                                     L.VDR_F32 if out_dtype == torch.float32 else L.VDR_BF16,
                                     scratch.data_ptr() if nb else None, _s(t)))
    return out


def prepare_image(img, device=None) -> torch.Tensor:
    """R2 input contract: img (h, w) or (h, w, 3) in [0, 1] -> float32 device tensor [1, 3, 1024|896, .]."""
    t = torch.as_tensor(img)
    v = t.unsqueeze(2)  # one-slice volume: (h, w, 1[, 3])
    return prepare_slices(v, device=device)


def apply_window_ct(ct, width, level, device=None) -> torch.Tensor:
    """HU volume (float32 or int16) -> float32 device tensor in [0, 1]."""
    lib = L.load()
    t = _dev(ct, device, (torch.float32, torch.int16)).contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    L.check(lib.vdr_op_window_ct(t.data_ptr(), L.VDR_F32 if t.dtype == torch.float32 else L.VDR_I16, t.numel(),
                                 float(width), float(level), out.data_ptr(), _s(t)))
    return out


def hu_to_rgb_vectorized(hu_matrix, device=None) -> torch.Tensor:
    """HU array (float32 / float64 / int16) -> uint8 device tensor hu.shape + (3,)."""
    lib = L.load()
    t = _dev(hu_matrix, device, (torch.float32, torch.float64, torch.int16)).contiguous()
    dt = {torch.float32: L.VDR_F32, torch.float64: L.VDR_F64, torch.int16: L.VDR_I16}[t.dtype]
    out = torch.empty(tuple(t.shape) + (3,), dtype=torch.uint8, device=t.device)
    L.check(lib.vdr_op_hu_to_rgb(t.data_ptr(), dt, t.numel(), out.data_ptr(), _s(t)))
    return out


def flip_image(image, mask, flip_type):
    """Views, as in the reference (no data moves until the consumer reads them)."""
    image, mask = torch.as_tensor(image), torch.as_tensor(mask)
    if flip_type == "horizontal":
        return image.flip(1), mask.flip(1)
    if flip_type == "vertical":
        return image.flip(0), mask.flip(0)
    return image, mask
