#!/opt/conda/bin/python3.9
"""Golden vectors for the pre/post-processing either side of the encoder.

Run with the conda interpreter of the authoring container (it has skimage 0.18.3, which the main interpreter
lacks):   /opt/conda/bin/python3.9 tests/golden/make_golden_prep.py

What produces the expected values:
  * crop_image / extract_coords / extract_roi / hu_to_rgb_vectorized: the reference's OWN functions, imported
    from /root/reference/src/visualization_utils.py (that module imports under this interpreter);
  * prepare_image's numpy part (tfds_dense_descriptor.py:43-47: gray2rgb + skimage.transform.resize): the
    same skimage calls made directly, because tfds_dense_descriptor.py itself needs torch / tensorflow_datasets
    / segment_anything, which this interpreter does not have (ordinary ModuleNotFoundError);
  * apply_window_ct / windowing_ct (tfds_dense_descriptor.py:204-237, 287-302): the reference's OWN functions, their
    definitions read from the reference file with `ast` and executed with numpy in scope.
Only numeric inputs / outputs are stored (tests/golden/prep_*.npz).
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")

import visualization_utils as vu  # noqa: E402  (the reference)
from skimage.color import gray2rgb  # noqa: E402
from skimage.transform import resize  # noqa: E402


def blob_mask(rng, h, w, n=3):
    m = np.zeros((h, w), dtype=bool)
    for _ in range(n):
        cy, cx = rng.integers(3, h - 3), rng.integers(3, w - 3)
        ry, rx = rng.integers(1, max(2, h // 6)), rng.integers(1, max(2, w // 6))
        yy, xx = np.ogrid[:h, :w]
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    return m


def gen_roi():
    rng = np.random.default_rng(11)
    out = {}
    cases = [(40, 40, (64, 64, 5)), (37, 53, (64, 64, 3)), (64, 64, (64, 64, 4)), (30, 30, (16, 16, 2)), (50, 20, (7, 9, 3)),
             (12, 12, (64, 64, 2))]
    for i, (h, w, fshape) in enumerate(cases):
        mask = blob_mask(rng, h, w)
        if i == 5:  # a single pixel: the margin clamps
            mask = np.zeros((h, w), dtype=bool)
            mask[4, 7] = True
        feat = rng.standard_normal(fshape).astype(np.float32)
        for margin in (1, 2):
            out[f"c{i}_coords_m{margin}"] = np.array(vu.extract_coords(mask, margin), dtype=np.int64)
        out[f"c{i}_mask"] = mask
        out[f"c{i}_feat"] = feat
        out[f"c{i}_roi_feat"] = vu.extract_roi(feat, mask)            # feature map cropped to the nodule box
        out[f"c{i}_roi_mask"] = vu.extract_roi(mask, mask)            # same-shape path
        xmin, ymin, xmax, ymax = [int(v) for v in rng.integers(-10, 70, 4)]
        out[f"c{i}_crop_args"] = np.array([xmin, ymin, xmax, ymax], dtype=np.int64)
        out[f"c{i}_crop"] = vu.crop_image(feat, xmin, ymin, xmax, ymax)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "prep_roi.npz"), **out)


def gen_hu():
    rng = np.random.default_rng(12)
    edges = np.array([-1500, -1000, -999.5, -800, -600, -500, -400, -399, -250, -100, -80, -60, -59, 0, 39.9, 40, 60, 80, 81,
                      200, 399.9, 400, 1000], dtype=np.float64)
    hu = np.concatenate([edges, rng.uniform(-1200, 600, 2025)]).reshape(32, 64)
    hu_i = rng.integers(-1100, 500, (16, 16)).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "prep_hu.npz"), hu=hu, rgb=vu.hu_to_rgb_vectorized(hu), hu_i16=hu_i,
                        rgb_i16=vu.hu_to_rgb_vectorized(hu_i), hu_f32=hu.astype(np.float32),
                        rgb_f32=vu.hu_to_rgb_vectorized(hu.astype(np.float32)))


def gen_resize():
    rng = np.random.default_rng(13)
    out = {}
    cases = [("up_gray", (20, 31), None, 64), ("up2_gray", (32, 32), None, 64), ("same_gray", (64, 64), None, 64),
             ("down_gray", (100, 90), None, 64), ("down_big_gray", (150, 150), None, 48), ("up_rgb", (30, 30), 3, 56),
             ("down_rgb", (80, 70), 3, 56), ("mixed_gray", (40, 100), None, 64)]
    for name, hw, ch, side in cases:
        img = rng.random(hw if ch is None else hw + (ch,)).astype(np.float32)
        x = gray2rgb(img) if ch is None else img          # prepare_image: tfds_dense_descriptor.py:43-47
        y = resize(x, (side, side))
        out[name + "_in"] = img
        out[name + "_out"] = y.transpose(2, 0, 1).astype(np.float32)  # CHW, as prepare_image returns it
        out[name + "_out_dtype"] = np.array(str(y.dtype))
    # float64 input (what apply_window_ct hands over for integer CT volumes)
    img = rng.random((25, 25))
    out["up_gray64_in"] = img
    out["up_gray64_out"] = resize(gray2rgb(img), (64, 64)).transpose(2, 0, 1)
    out["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "prep_resize.npz"), **out)


def gen_window():
    rng = np.random.default_rng(14)
    ct = rng.integers(-1200, 1200, (24, 24)).astype(np.int16)
    ctf = rng.uniform(-1200, 1200, (24, 24)).astype(np.float32)
    out = {"ct_i16": ct, "ct_f32": ctf}
    # the reference's OWN apply_window_ct / windowing_ct: their definitions are read from the reference file and executed
    # with numpy in scope (the module itself needs torch / tensorflow_datasets / segment_anything to import)
    import ast
    path = "/root/reference/src/tfds_dense_descriptor.py"
    scope = {"np": np}
    body = [n for n in ast.parse(open(path).read()).body
            if isinstance(n, ast.FunctionDef) and n.name in ("windowing_ct", "apply_window_ct")]
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), scope)
    for tag, (w, l) in {"w800_l40": (800, 40), "w1500_lm600": (1500, -600), "w350_l50": (350, 50)}.items():
        for nm, a in (("i16", ct), ("f32", ctf)):
            out[f"{tag}_{nm}"] = scope["apply_window_ct"](a, width=w, level=l)
    np.savez_compressed(os.path.join(HERE, "prep_window.npz"), **out)


if __name__ == "__main__":
    gen_roi()
    gen_hu()
    gen_resize()
    gen_window()
    print("wrote prep_roi.npz prep_hu.npz prep_resize.npz prep_window.npz")
