"""On-disk feature store of the reference's Stage A (src/tfds_dense_descriptor.py:142-165 `save_features`), read back
by Stage B at src/train_models.py:147-157: one HDF5 file per modality, group `{patient_id}`, datasets
`features/{i}` (fp32 (h', w', D) ROI-cropped descriptor map of feature i) and `masks/{i}` (boolean nodule mask crop),
lzf-compressed, ONE chunk per dataset (chunks == shape), a patient's group replaced when it already exists.

Deliberately free of torch / GPU imports (numpy + h5py only): the writer also runs under interpreters that have h5py but
no torch (this image's /opt/conda/bin/python3.9), e.g. as a writer process beside the GPU extraction."""
from __future__ import annotations

import numpy as np


def _h5py():
    try:
        import h5py
    except ImportError as e:  # not installed in every interpreter; nothing else can write this format
        raise RuntimeError("save_features needs h5py (the reference's on-disk format is HDF5)") from e
    return h5py


def save_features(filename, all_features, all_masks, patient_id):
    """tfds_dense_descriptor.py:142-165.  all_features / all_masks: one array per feature id, in feature_id order."""
    h5py = _h5py()
    with h5py.File(filename, "a") as h5f:
        if patient_id in h5f:  # :153-155: an existing patient is overwritten, not appended to
            del h5f[patient_id]
        grp = h5f.create_group(patient_id)
        for i, (feature, mask) in enumerate(zip(all_features, all_masks)):
            feature, mask = np.asarray(feature), np.asarray(mask)
            grp.create_dataset(f"features/{i}", compression="lzf", data=feature, chunks=feature.shape)
            grp.create_dataset(f"masks/{i}", compression="lzf", data=mask, chunks=mask.shape)


def read_features(filename, patient_id, feature_ids):
    """The access pattern of PETCTDataset3D._get_features (train_models.py:147-151): `h5f[f'{patient}/features/{id}'][()]`
    and the matching mask, for the listed feature ids."""
    h5py = _h5py()
    feats, masks = [], []
    with h5py.File(filename, "r") as h5f:
        for fid in feature_ids:
            feats.append(h5f[f"{patient_id}/features/{fid}"][()])
            masks.append(h5f[f"{patient_id}/masks/{fid}"][()])
    return feats, masks
