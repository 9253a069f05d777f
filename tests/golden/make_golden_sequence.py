#!/opt/conda/bin/python3.9
"""Golden vectors for the Stage-C input builder: the 'transformer' branch of
PETCTDataset3D._get_features (reference src/train_models.py:143-182) with positional_encoding_3d (:30-44).

    /opt/conda/bin/python3.9 tests/golden/make_golden_sequence.py        (skimage 0.18.3 lives there)

train_models.py itself does not import under this interpreter (no torch / h5py here: ordinary ModuleNotFoundError),
so the expected values come from the same numpy / skimage calls made directly, in the reference's order, on the
arrays it would have read from the HDF5 file.  skimage 0.18.3 returns the order-0 resized boolean mask as float 0/1
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


def positional_encoding_3d(x, y, z, D, scale=10000):
    x, y, z = np.asarray(x), np.asarray(y), np.asarray(z)
    encoding = np.zeros((x.shape[0], D))
    for i in range(D // 6):
        exponent = scale ** (6 * i / D)
        encoding[:, 2 * i] = np.sin(x / exponent)
        encoding[:, 2 * i + 1] = np.cos(x / exponent)
        encoding[:, 2 * i + D // 3] = np.sin(y / exponent)
        encoding[:, 2 * i + 1 + D // 3] = np.cos(y / exponent)
        encoding[:, 2 * i + 2 * D // 3] = np.sin(z / exponent)
        encoding[:, 2 * i + 1 + 2 * D // 3] = np.cos(z / exponent)
    return encoding


def get_features(slice_features_list, slice_masks_list, noise, spatial_res, feature_dim):
    features, masks = [], []
    for slice_features, slice_mask_orig in zip(slice_features_list, slice_masks_list):
        slice_mask = resize(slice_mask_orig, slice_features.shape[0:2], order=0).astype(bool)
        masks.append(np.expand_dims(slice_mask, axis=-1))
        features.append(slice_features)
    features = np.transpose(np.stack(features, axis=0), axes=(3, 0, 1, 2))
    masks = np.transpose(np.stack(masks, axis=0), axes=(1, 2, 0, 3))
    h_orig, w_orig = slice_mask_orig.shape[0:2]
    features = np.transpose(features, axes=(2, 3, 1, 0))
    h_new, w_new = features.shape[0], features.shape[1]
    x, y, z = np.meshgrid(np.arange(0, features.shape[0]), np.arange(0, features.shape[1]), np.arange(0, features.shape[2]))
    x = (x.flatten() / w_new).flatten() * w_orig * spatial_res[0]
    y = (y.flatten() / h_new).flatten() * h_orig * spatial_res[1]
    z = (z.flatten()).flatten() * spatial_res[2]
    masks = masks.flatten()
    x = (x - x.mean() + noise[0])[masks]
    y = (y - y.mean() + noise[1])[masks]
    z = (z - z.mean() + noise[2])[masks]
    pe = positional_encoding_3d(x, y, z, D=feature_dim, scale=10000)
    return features.reshape(-1, feature_dim)[masks, :] + pe / 4, masks


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
