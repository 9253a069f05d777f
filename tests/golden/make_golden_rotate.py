"""Golden vectors for the offline rotation augmentation (`rotate_image`,
reference src/tfds_dense_descriptor.py:327-350).

    python tests/golden/make_golden_rotate.py          (main interpreter: scipy 1.15.3)

tfds_dense_descriptor.py itself does not import here (skimage / tensorflow_datasets / segment_anything are missing:
ordinary ModuleNotFoundError), so the definition of the reference's OWN rotate_image is read from its file with `ast`
and executed with numpy and scipy.ndimage.rotate in scope (all it needs): rotate(vol, angle, axes=(0, 1),
reshape=False, mode='nearest'), np.clip(., 0, 1) on the image and `> 0` on the rotated boolean mask.  Only numeric inputs / outputs are stored (tests/golden/rotate_cases.npz).
"""
import os

import numpy as np
from scipy.ndimage import rotate

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_function(path, name, scope):
    """The reference's OWN `name`, taken from its source file at generation time (the module itself does not import
    here) and executed with `scope` as its globals; nothing of it is stored in this repository."""
    import ast
    node = next(n for n in ast.parse(open(path).read()).body if isinstance(n, ast.FunctionDef) and n.name == name)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scope)
    return scope[name]


rotate_image = reference_function("/root/reference/src/tfds_dense_descriptor.py", "rotate_image", {"np": np, "rotate": rotate})


def main():
    rng = np.random.default_rng(2024)
    out = {}
    names = []
    # (H, W, S) CT-like volume in [0, 1] (float64, what apply_window_ct hands over), non-square, blob mask
    for name, shape, dtype in (("ct64", (29, 23, 3), np.float64), ("pet32", (24, 24, 2), np.float32),
                               ("rgb64", (21, 26, 2, 3), np.float64)):
        img = rng.random(shape).astype(dtype)
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        mask = np.zeros(shape[:3], dtype=bool)
        for s in range(shape[2]):
            mask[:, :, s] = (yy - shape[0] * 0.45 - s) ** 2 + (xx - shape[1] * 0.55) ** 2 < (4 + s) ** 2
        out[name + "_img"], out[name + "_mask"] = img, mask
        for angle in (45, 90, 135):
            ri, rm = rotate_image(img, mask, angle)
            out[f"{name}_img_{angle}"], out[f"{name}_mask_{angle}"] = ri, rm
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "rotate_cases.npz"), **out)
    print("wrote rotate_cases.npz", {k: v.shape for k, v in out.items() if k.endswith("_45")})


if __name__ == "__main__":
    main()
