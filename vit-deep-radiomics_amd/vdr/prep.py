"""GPU counterparts of the reference's per-slice numpy / skimage pre-processing.

Same names, argument meaning and return contracts as the functions they replace; inputs may be numpy arrays
(uploaded as they are: a 512x512 slice instead of the 12.6 MB resized tensor) or device tensors.

  prepare_image(img)                    src/tfds_dense_descriptor.py:30-48
  prepare_slices(volume, ...)           the same for all slices of an (H, W, S[, C]) volume in ONE launch
  apply_window_ct(ct, width, level)     src/tfds_dense_descriptor.py:287-302 (windowing_ct :204-237)
  hu_to_rgb_vectorized(hu)              src/visualization_utils.py:128-186
  flip_image(image, mask, flip_type)    src/tfds_dense_descriptor.py:305-324 (views; prepare_slices(flip=...) folds
                                        the image flip into the resize gather instead)

  rotate_volume / rotate_image          src/tfds_dense_descriptor.py:327-350 (scipy.ndimage.rotate, cubic spline,
                                        mode 'nearest', in the (0, 1) plane): float64 on the GPU, bit-identical

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
    """HU volume (float32 or int16) -> float32 device tensor in [0, 1], bit-identical to the reference's numpy result
    for those dtypes.  A float64 volume is rounded to float32 FIRST (numpy would window in float64 and return float64):
    the only consumer, prepare_image, casts to float32 anyway; pass float32 when the last ulp has to match."""
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


def affine_cubic(vol: torch.Tensor, matrix, offset, clip01=False) -> torch.Tensor:
    """scipy.ndimage.affine_transform(plane, matrix, offset, order=3, mode='nearest') on every (H, W) plane of a
    device volume [H, W, ...] (float64, float32 or bool); same shape and dtype out."""
    lib = L.load()
    assert vol.is_cuda and vol.dim() >= 2
    dt = {torch.float64: L.VDR_F64, torch.float32: L.VDR_F32, torch.bool: L.VDR_U8}.get(vol.dtype)
    if dt is None:
        raise TypeError(f"affine_cubic: float64, float32 or bool volume, got {vol.dtype}")
    vol = vol.contiguous()
    H, W = vol.shape[:2]
    planes = vol.numel() // (H * W)
    out = torch.empty_like(vol)
    if planes == 0:
        return out
    scratch = torch.empty(lib.vdr_affine_cubic_scratch_bytes(H, W, planes), dtype=torch.uint8, device=vol.device)
    import ctypes as C
    m = (C.c_double * 4)(*[float(v) for v in np.asarray(matrix, dtype=np.float64).reshape(-1)])
    o = (C.c_double * 2)(*[float(v) for v in np.asarray(offset, dtype=np.float64).reshape(-1)])
    L.check(lib.vdr_op_affine_cubic(vol.data_ptr(), dt, H, W, planes, m, o, out.data_ptr(), int(bool(clip01)),
                                    scratch.data_ptr(), _s(vol)))
    return out


def rotation_matrix_offset(shape_hw, angle):
    """The (matrix, offset) scipy.ndimage.rotate(axes=(0, 1), reshape=False) passes to affine_transform: the same
    numpy / scipy.special expressions, so the six doubles are the ones SciPy computes."""
    from scipy import special
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    plane = np.asarray(shape_hw)
    out_center = rot @ ((plane - 1) / 2)
    in_center = (plane - 1) / 2
    return rot, in_center - out_center


def rotate_volume(vol, angle, clip01=False, device=None) -> torch.Tensor:
    """scipy.ndimage.rotate(vol, angle, axes=(0, 1), reshape=False, mode='nearest') for an (H, W, S[, C]) volume
    (numpy array or tensor; float64 / float32 / bool) on the GPU; returns a device tensor of the same dtype."""
    t = torch.as_tensor(vol)
    if t.dtype not in (torch.float64, torch.float32, torch.bool):
        # SciPy rounds and casts into an integer output array; the reference only rotates float images and boolean
        # masks, so that store is not restated: refuse rather than return something SciPy would not
        raise TypeError(f"rotate_volume: float64, float32 or bool volume, got {t.dtype}")
    t = t.to(device if device is not None else (t.device if t.is_cuda else "cuda"))
    rot, off = rotation_matrix_offset(t.shape[:2], angle)
    return affine_cubic(t, rot, off, clip01=clip01)


def rotate_image(image, mask, angle, axes=(0, 1), device=None):
    """rotate_image of the reference: image rotated and clipped to [0, 1], mask rotated and thresholded `> 0`.
    Returns device tensors (image dtype kept, mask bool); angle 0 returns copies, as the reference does."""
    if tuple(axes) != (0, 1):
        raise NotImplementedError("rotate_image: the reference only rotates in the (0, 1) plane")
    if angle == 0:
        dev = device if device is not None else "cuda"
        return torch.as_tensor(image).to(dev).clone(), torch.as_tensor(mask).to(dev).clone()
    img = rotate_volume(image, angle, clip01=True, device=device)
    m = torch.as_tensor(mask)
    m = rotate_volume(m if m.dtype == torch.bool else m.to(torch.bool), angle, device=device)
    return img, m
