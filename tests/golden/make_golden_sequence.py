#!/opt/conda/bin/python3.9
"""Golden vectors for the Stage-C input builder: the 'transformer' branch of
PETCTDataset3D._get_features (reference src/train_models.py:143-182) with positional_encoding_3d (:30-44).

    /opt/conda/bin/python3.9 tests/golden/make_golden_sequence.py        (skimage 0.18.3 lives there)

train_models.py itself does not import under this interpreter (no torch / h5py here: ordinary ModuleNotFoundError),
so: `positional_encoding_3d` is the reference's OWN function (its definition is read from the reference file with `ast`
and executed here; it needs numpy only), and the few array steps around it follow the reference's order with the same
numpy / skimage calls, on the arrays it would have read from the HDF5 file.  skimage 0.18.3 returns the order-0 resized boolean mask as float 0/1
(newer releases keep bool, which is what the reference's boolean indexing needs): the values are cast to bool here.
Original mask sizes are odd so that no sample falls exactly between two source pixels: on such ties 0.18.3's warp and
the ndimage.zoom(grid_mode=True) that newer skimage releases call disagree (the oracle follows zoom, cross-checked
against scipy.ndimage.zoom in tests/test_oracle.py).
Only numeric inputs / outputs are stored (tests/golden/sequence_cases.npz)."""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.transform import resize  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_function(path, name):
    """The reference's OWN `name`, taken from its source file at generation time (the module itself does not import
    here) and executed with numpy in scope; nothing of it is stored in this repository."""
    import ast
    src = open(path).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    scope = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scope)
    return scope[name]


positional_encoding_3d = reference_function("/root/reference/src/train_models.py", "positional_encoding_3d")


def get_features(maps, nodule_masks, noise, spacing, width):
    """What the 'transformer' branch of _get_features (train_models.py:143-182) yields for the arrays it would have
    read from the HDF5 file: the skimage order-0 mask resize, its (h, w, slice) flattening, the np.meshgrid (default
    'xy') coordinate grids scaled to millimetres, centred and offset by `noise`, the reference's own
    positional_encoding_3d, and `features + pe / 4` on the kept voxels."""
    vol = np.stack(maps, axis=0).transpose(1, 2, 0, 3)                                   # (h, w, slice, width)
    h, w, n_slices = vol.shape[:3]
    small = [resize(m, (h, w), order=0).astype(bool) for m in nodule_masks]
    keep = np.stack(small, axis=0).transpose(1, 2, 0).reshape(-1)
    src_h, src_w = nodule_masks[-1].shape[0:2]
    gx, gy, gz = np.meshgrid(np.arange(0, h), np.arange(0, w), np.arange(0, n_slices))
    cx = gx.flatten() / w * src_w * spacing[0]
    cy = gy.flatten() / h * src_h * spacing[1]
    cz = gz.flatten() * spacing[2]
    coords = [(c - c.mean() + d)[keep] for c, d in zip((cx, cy, cz), noise)]
    pe = positional_encoding_3d(*coords, D=width, scale=10000)
    return vol.reshape(-1, width)[keep, :] + pe / 4, keep


def main():
    rng = np.random.default_rng(77)
    out = {}
    cases = [("sq", 12, 12, 5, 48, (41, 41)), ("rect", 9, 14, 3, 48, (33, 51)), ("ref256", 10, 10, 3, 256, (63, 65)),
             ("up", 20, 18, 3, 24, (11, 9))]
    for name, h, w, S, D, (h0, w0) in cases:
        feats = [rng.standard_normal((h, w, D)).astype(np.float32) for _ in range(S)]
        yy, xx = np.mgrid[0:h0, 0:w0]
        masks = [(((yy - h0 * 0.5 - s) / (h0 * 0.3)) ** 2 + ((xx - w0 * 0.45) / (w0 * 0.25)) ** 2) < 1.0 for s in range(S)]
        res = np.array([0.7, 0.8, 1.25]) if name != "sq" else np.array([1.0, 1.0, 3.0])
        noise = np.array([0.0, 0.0, 0.0]) if name in ("sq", "ref256") else rng.standard_normal(3)
        seq, m = get_features(feats, masks, noise, res, D)
        out[f"{name}_feats"], out[f"{name}_masks"] = np.stack(feats), np.stack(masks)
        out[f"{name}_res"], out[f"{name}_noise"], out[f"{name}_seq"], out[f"{name}_keep"] = res, noise, seq, m
    out["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "sequence_cases.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("_seq")})


if __name__ == "__main__":
    main()
