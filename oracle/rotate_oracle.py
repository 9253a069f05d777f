"""CPU oracle (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product) for the in-plane rotation the
reference applies as offline augmentation: `rotate_image` (reference src/tfds_dense_descriptor.py:327-350) =
`scipy.ndimage.rotate(vol, angle, axes=(0, 1), reshape=False, mode='nearest')` (order 3, prefilter on) on the
image, clipped to [0, 1], and on the boolean mask followed by `> 0`.

The arithmetic lives in a third-party dependency that is not vendored in /root/reference: SciPy (the reference
pins no version; 1.15.3 is what this image holds).  This file restates SciPy's published algorithm
(scipy/ndimage/_interpolation.py `rotate` / `affine_transform`, src/ni_splines.c, src/ni_interpolation.c) in
numpy, operation for operation, so that the float64 results are BIT-IDENTICAL to SciPy's; it is pinned by
tests/test_oracle.py against scipy.ndimage itself (importable here and on the GPU box) and against the committed
fixtures tests/golden/rotate_*.npz generated from it.

  per (H, W) plane of the volume:
    1. edge-pad by 12 samples on both axes (`_prepad_for_spline_filter`, mode 'nearest')
    2. cubic B-spline prefilter along axis 0, then axis 1 (pole z = sqrt(3) - 2 as the literal SciPy uses, gain
       (1 - z)(1 - 1/z), 'reflect' boundary initialisation for mode 'nearest', in-place causal + anti-causal pass)
    3. for every output pixel o: x = ((M o) + offset) + 12 per axis (not clamped); 4 x 4 taps from floor(x) - 1,
       tap indices clamped to the padded array; weights of `get_spline_interpolation_weights`;
       t = sum_i sum_j (c[i][j] * w0[i]) * w1[j] accumulated in that order from 0.0
    4. store: float -> cast to the input dtype; bool -> (unsigned char) t (truncation), i.e. True iff t >= 1
"""
import math

import numpy as np

NPAD = 12
POLE = -0.267949192431122706472553658494127633  # SciPy's literal for sqrt(3) - 2 (correctly rounded, unlike sqrt(3.0) - 2.0)


def rotation_matrix_offset(shape_hw, angle):
    """The (matrix, offset) scipy.ndimage.rotate hands to affine_transform for axes=(0, 1), reshape=False."""
    from scipy import special
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    plane = np.asarray(shape_hw)
    out_center = rot @ ((plane - 1) / 2)
    in_center = (plane - 1) / 2
    return rot, in_center - out_center


def _filter_axis0(c):
    """In-place cubic prefilter along axis 0 of a float64 array [n, ...] (every column is one line)."""
    z = POLE
    n = c.shape[0]
    if n < 2:
        return
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    z_n = math.pow(z, n)
    c0 = c[0].copy()
    acc = c[0] + z_n * c[n - 1]
    z_i = z
    for i in range(1, n):
        # at i = n - 1 SciPy reads c[0] while it is being accumulated (in place)
        other = acc if i == n - 1 else c[n - 1 - i]
        acc = acc + z_i * (c[i] + z_n * other)
        z_i *= z
    acc = acc * (z / (1 - z_n * z_n))
    c[0] = acc + c0
    for i in range(1, n):
        c[i] = c[i] + z * c[i - 1]
    c[n - 1] = c[n - 1] * (z / (z - 1))
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])


def spline_coefficients(plane_stack):
    """[H, W, T] any real dtype -> float64 [H + 24, W + 24, T] prefiltered coefficients (steps 1-2)."""
    p = np.pad(np.asarray(plane_stack), ((NPAD, NPAD), (NPAD, NPAD), (0, 0)), mode="edge").astype(np.float64)
    _filter_axis0(p)
    q = np.ascontiguousarray(p.transpose(1, 0, 2))
    _filter_axis0(q)
    return np.ascontiguousarray(q.transpose(1, 0, 2))


def _axis_taps(x, n):
    """x float64 [..] coordinates in the padded array of length n -> (idx [4, ..] int, w [4, ..] float64).
    Mode 'nearest' with a spline order > 1: the coordinate itself is NOT clamped; the four tap indices are (the
    coefficient array is extended by its edge sample), the weights are those of the true coordinate."""
    fl = np.floor(x)
    start = fl.astype(np.int64) - 1
    y = x - fl
    zz = 1.0 - y
    w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
    w2 = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0
    w0 = zz * zz * zz / 6.0
    w3 = 1.0 - w0 - w1 - w2
    idx = [np.clip(start + k, 0, n - 1) for k in range(4)]
    return np.stack(idx), np.stack([w0, w1, w2, w3])


def affine_sample(coef, matrix, offset, out_hw):
    """Step 3 on coefficients [Hp, Wp, T]: float64 [H, W, T]."""
    Hp, Wp, T = coef.shape
    H, W = out_hw
    o0 = np.arange(H, dtype=np.float64)[:, None]
    o1 = np.arange(W, dtype=np.float64)[None, :]
    m = np.asarray(matrix, dtype=np.float64)
    off = np.asarray(offset, dtype=np.float64)
    x0 = (o0 * m[0, 0] + o1 * m[0, 1]) + off[0] + NPAD
    x1 = (o0 * m[1, 0] + o1 * m[1, 1]) + off[1] + NPAD
    i0, w0 = _axis_taps(x0, Hp)
    i1, w1 = _axis_taps(x1, Wp)
    t = np.zeros((H, W, T), dtype=np.float64)
    for a in range(4):
        for b in range(4):
            t = t + (coef[i0[a], i1[b]] * w0[a][..., None]) * w1[b][..., None]
    return t


def rotate_planes(vol, angle):
    """scipy.ndimage.rotate(vol, angle, axes=(0, 1), reshape=False, mode='nearest') for vol [H, W, ...]."""
    vol = np.asarray(vol)
    H, W = vol.shape[:2]
    v3 = vol.reshape(H, W, -1)
    rot, off = rotation_matrix_offset((H, W), angle)
    t = affine_sample(spline_coefficients(v3), rot, off, (H, W))
    if vol.dtype == np.bool_:
        out = t.astype(np.int64).astype(np.uint8).astype(np.bool_)  # (npy_bool) t: truncation toward zero
    else:
        out = t.astype(vol.dtype)
    return out.reshape(vol.shape)


def rotate_image(image, mask, angle):
    """The reference's rotate_image (tfds_dense_descriptor.py:327-350)."""
    image, mask = np.asarray(image), np.asarray(mask)
    if angle == 0:
        return image.copy(), mask.copy()
    return np.clip(rotate_planes(image, angle), 0, 1), rotate_planes(mask, angle) > 0
