"""Batched slice pipeline around the encoder: the reference's per-slice hot loop
`generate_features` (src/tfds_dense_descriptor.py:242-284) with every slice of a volume in one batch, the ROI
maths of src/visualization_utils.py:93-125, and the HDF5 layout `save_features` (:142-165) writes, so
`train_models.py:147-157` reads the files unchanged.

Host-side integer logic (boxes, clamping) is restated here; pixels only move on the GPU (prep.prepare_slices ->
encoder -> vdr_op_crop_hwc) and come back once per volume, already cropped to the nodule box.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import prep


# ---- visualization_utils.py:93-125 ------------------------------------------------------------------------
def crop_box(shape_hw, xmin, ymin, xmax, ymax):
    """crop_image's clamping: the (y0, y1, x0, x1) slice bounds it applies to an array of spatial shape shape_hw."""
    h, w = shape_hw
    y0, y1 = [max(0, min(int(v), h)) for v in (ymin, ymax)]
    x0, x1 = [max(0, min(int(v), w)) for v in (xmin, xmax)]
    return y0, y1, x0, x1


def crop_image(img, xmin, ymin, xmax, ymax):
    y0, y1, x0, x1 = crop_box(img.shape[0:2], xmin, ymin, xmax, ymax)
    return img[y0:y1, x0:x1]


def extract_coords(mask, margin):
    """Box of a boolean mask; the reference's margins are asymmetric (ymin - m, xmin + m, ymax - m, xmax + m)."""
    m = np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)
    ys, xs = np.nonzero(m)
    if ys.size == 0:
        raise ValueError("empty mask")  # the reference raises here too (np.min of an empty array)
    ymin, xmin = int(ys.min()) - margin, int(xs.min()) + margin
    ymax, xmax = int(ys.max()) - margin, int(xs.max()) + margin
    h = max(ymax - ymin, margin)
    w = max(xmax - xmin, margin)
    return xmin, ymin, xmin + w, ymin + h


def roi_box(img_hw, mask, margin=1):
    """The box extract_roi crops an array of spatial shape img_hw to (rescaled when img and mask differ in size)."""
    xmin, ymin, xmax, ymax = extract_coords(mask, margin)
    mh, mw = mask.shape[0:2]
    if tuple(img_hw) != (mh, mw):
        h = img_hw[0] / mh
        w = img_hw[1] / mw
        xmin, ymin, xmax, ymax = [int(v) for v in (xmin * w, ymin * h, xmax * w, ymax * h)]
        hh = max(ymax - ymin, margin)
        ww = max(xmax - xmin, margin)
        xmax, ymax = xmin + ww, ymin + hh
    return xmin, ymin, xmax, ymax


def extract_roi(img, mask, margin=1):
    return crop_image(img, *roi_box(img.shape[0:2], mask, margin))


def crop_maps(maps: torch.Tensor, box) -> torch.Tensor:
    """maps fp32 [B, H, W, C] on the device -> [B, h', w', C] cropped to `box` = (xmin, ymin, xmax, ymax) with
    crop_image's clamping, by the HIP crop kernel."""
    lib = L.load()
    assert maps.is_cuda and maps.dtype == torch.float32 and maps.is_contiguous() and maps.dim() == 4
    B, H, W, C = maps.shape
    y0, y1, x0, x1 = crop_box((H, W), *box)
    if y1 <= y0 or x1 <= x0:
        return torch.empty((B, max(y1 - y0, 0), max(x1 - x0, 0), C), dtype=torch.float32, device=maps.device)
    out = torch.empty((B, y1 - y0, x1 - x0, C), dtype=torch.float32, device=maps.device)
    L.check(lib.vdr_op_crop_hwc(maps.data_ptr(), out.data_ptr(), B, H, W, C, y0, x0, y1 - y0, x1 - x0,
                                torch.cuda.current_stream(maps.device).cuda_stream))
    return out


# ---- tfds_dense_descriptor.py:242-284 -----------------------------------------------------------------------
def generate_features(model, img_3d, mask_3d, flip=None, max_batch=16):
    """Feature map of every slice, cropped to the nodule region.

    img_3d  (H, W, S) CT/PET volume in [0, 1] ('medsam') or (H, W, S, 3) ('dinov2'); mask_3d (H, W, S) bool.
    Returns (features_list, mask_list) exactly like the reference: per slice a (h', w', D) float32 array and
    the (h'', w'') bool mask crop.  All S slices go through prepare -> encoder -> ROI crop in batches of
    `max_batch` on the GPU; one D2H per batch of the already-cropped maps.

    flip ('horizontal' | 'vertical' | None): the result is that of the reference's
    `generate_features(model, *flip_image(img_3d, mask_3d, flip))` (tfds_dense_descriptor.py:463-467): the mask is
    flipped first, every box (volume crop, ROI of the feature maps, ROI of the masks -- asymmetric margins and all) is
    derived from the FLIPPED mask, and the pixels are read from the mirrored window of the unflipped volume with the
    reversal folded into the resize gather (no flipped copy of the volume is made)."""
    if flip not in (None, "horizontal", "vertical"):
        raise ValueError(f"flip must be None, 'horizontal' or 'vertical', got {flip!r}")
    mask_np = mask_3d.cpu().numpy() if isinstance(mask_3d, torch.Tensor) else np.asarray(mask_3d)
    if flip == "horizontal":
        mask_np = mask_np[:, ::-1]
    elif flip == "vertical":
        mask_np = mask_np[::-1]
    bigger_mask = np.sum(mask_np, axis=-1) > 0
    xmin, ymin, xmax, ymax = extract_coords(bigger_mask, margin=2)
    crop_size = max(xmax - xmin, ymax - ymin) * 2
    xmid, ymid = int(xmin + (xmax - xmin) / 2), int(ymin + (ymax - ymin) / 2)
    box = (xmid - crop_size, ymid - crop_size, xmid + crop_size, ymid + crop_size)

    vol = torch.as_tensor(img_3d)
    y0, y1, x0, x1 = crop_box(vol.shape[0:2], *box)   # in the coordinates of the flipped volume
    vy0, vy1, vx0, vx1 = y0, y1, x0, x1              # the same window in the unflipped volume: mirrored
    if flip == "horizontal":
        vx0, vx1 = vol.shape[1] - x1, vol.shape[1] - x0
    elif flip == "vertical":
        vy0, vy1 = vol.shape[0] - y1, vol.shape[0] - y0
    vol = vol[vy0:vy1, vx0:vx1]                  # a view; prepare_slices reads it strided (and reversed, for a flip)
    mask_c = mask_np[y0:y1, x0:x1]
    bigger_c = bigger_mask[y0:y1, x0:x1]
    vol = vol.to(model.device)
    if vol.dtype not in (torch.float32, torch.float64):
        vol = vol.to(torch.float32)
    S = vol.shape[2]
    medsam = model.model_name == "medsam"
    features_list, mask_list = [], []
    rb = mb = None
    pending = []
    for s0 in range(0, S, max_batch):
        s1 = min(S, s0 + max_batch)
        # bf16 pixels: the patch GEMM rounds them to bf16 in any case (same round-to-nearest-even, same features bit
        # for bit); written as bf16 here they are half the bytes, and for p = 16 (MedSAM) the GEMM gathers them straight
        # from the images (no im2col pass)
        x = prep.prepare_slices(vol[:, :, s0:s1], side=model.cfg.img, flip=flip, out_dtype=torch.bfloat16, device=model.device)
        if medsam:
            maps = model.engine.forward(x, L.OUT_ENCODER, torch.float32)          # [b, g, g, C] channel-last
        else:
            g = model.cfg.img // model.cfg.patch
            maps = model.engine.forward(x, L.OUT_PATCH_EMBED, torch.float32).reshape(s1 - s0, g, g, model.cfg.dim)
        if rb is None:  # the boxes depend on the (cropped) union mask only: once per volume, not once per slice
            rb = roi_box(maps.shape[1:3], bigger_c)
            mb = roi_box(mask_c.shape[0:2], bigger_c)
        crops = crop_maps(maps.contiguous(), rb)
        host = torch.empty(crops.shape, dtype=crops.dtype, pin_memory=True)
        host.copy_(crops, non_blocking=True)   # D2H queued behind the crop; the next batch is enqueued without a host sync
        pending.append((host, s0, s1))
    torch.cuda.current_stream(model.device).synchronize()  # ONE synchronisation per volume
    for host, s0, s1 in pending:
        arr = host.numpy()
        for i in range(s1 - s0):
            features_list.append(arr[i])
            mask_list.append(crop_image(mask_c[:, :, s0 + i] > 0, *mb))
    return features_list, mask_list


# ---- tfds_dense_descriptor.py:142-165 -----------------------------------------------------------------------
from .h5store import read_features, save_features  # noqa: E402,F401  (torch-free module: also runs where only h5py is)


# ---- tfds_dense_descriptor.py:452-491 ------------------------------------------------------------------------
AUGMENTATIONS = [(flip, angle) for flip in (None, "horizontal", "vertical") for angle in range(0, 180, 45)]


def feature_metadata(n_features_per_aug, patient_id, label, dataset_name, modality, spatial_res, augmentations=None):
    """The per-patient table the reference stores next to the HDF5 features (`{patient}_{modality}.parquet`):
    one row per feature map, columns feature_id, slice, angle, flip, patient_id, label, dataset, modality,
    augmentation, spatial_res — what train_models.py:147-157 / prepare_df read.

    n_features_per_aug: number of slices produced for each (flip, angle) in `augmentations` (default: the
    reference's 3 flips x 4 angles, in its loop order).  `augmentation` is True on every row: the reference
    computes `not (df['flip'] is None and angle == 0)` with `df['flip'] is None` evaluated on the Series object
    (always False), and consumers were written against that output."""
    import pandas as pd

    augmentations = AUGMENTATIONS if augmentations is None else list(augmentations)
    if isinstance(n_features_per_aug, int):
        n_features_per_aug = [n_features_per_aug] * len(augmentations)
    rows = {"slice": [], "angle": [], "flip": []}
    for (flip, angle), n in zip(augmentations, n_features_per_aug):
        rows["angle"] += [angle] * n
        rows["flip"] += [flip] * n
        rows["slice"] += list(range(n))
    df = pd.DataFrame(rows)
    df.reset_index(drop=False, inplace=True)
    df = df.rename(columns={"index": "feature_id"})
    df["patient_id"] = patient_id
    df["label"] = label
    df["dataset"] = dataset_name.replace("_dataset", "")
    df["modality"] = modality
    df["augmentation"] = True
    df["spatial_res"] = [spatial_res] * df.shape[0]
    return df


def rotate_image(image, mask, angle, axes=(0, 1)):
    """tfds_dense_descriptor.py:327-350 on the GPU (prep.rotate_image: SciPy's cubic-spline rotation restated in
    float64, bit-identical): image clipped to [0, 1], mask > 0.  numpy in -> numpy out (the reference's contract);
    device tensors in -> device tensors out, which generate_features consumes without another upload."""
    on_host = not (isinstance(image, torch.Tensor) and image.is_cuda)
    img, m = prep.rotate_image(image, mask, angle, axes=axes)
    if on_host:
        return img.cpu().numpy(), m.cpu().numpy()
    return img, m


def extract_patient_features(model, img_raw, mask_raw, patient_id, label, dataset_name, modality, spatial_res,
                             flips=(None, "horizontal", "vertical"), angles=(0, 45, 90, 135), max_batch=16):
    """The reference's per-patient augmentation loop (tfds_dense_descriptor.py:452-491): for every flip x angle,
    generate_features over the whole volume; returns (all_features, all_masks, metadata DataFrame) ready for
    save_features / save_metadata.  The volume and its mask are uploaded ONCE; flips (torch.flip), rotations
    (prep.rotate_image), resize, encoder and ROI crop all run on the GPU, and only the cropped feature maps and the
    rotated masks (needed for the boxes) come back."""
    all_features, all_masks, counts, augs = [], [], [], []
    img = torch.as_tensor(np.asarray(img_raw) if not isinstance(img_raw, torch.Tensor) else img_raw)
    if img.dtype not in (torch.float32, torch.float64):
        # scipy.ndimage.rotate on an INTEGER volume rounds its result back to the integer dtype; that store is not
        # restated on the GPU (prep.rotate_volume refuses integers), and the reference's volumes are floats at this
        # point (apply_window_ct / PET SUV): refuse rather than rotate something SciPy would round differently
        if any(a != 0 for a in angles):
            raise TypeError(f"extract_patient_features rotates float32 / float64 volumes only, got {img.dtype}; "
                            "convert explicitly (the reference passes windowed CT / PET floats)")
        img = img.to(torch.float64)
    img = img.to(model.device)
    mask = torch.as_tensor(np.asarray(mask_raw) if not isinstance(mask_raw, torch.Tensor) else mask_raw).to(model.device)
    mask = mask if mask.dtype == torch.bool else mask > 0

    for flip_type in flips:
        im_f, m_f = prep.flip_image(img, mask, flip_type)  # flip_image (:305-324)
        for angle in angles:
            im, m = (im_f, m_f) if angle == 0 else prep.rotate_image(im_f, m_f, angle)
            feats, fmasks = generate_features(model, im, m, max_batch=max_batch)
            all_features += feats
            all_masks += fmasks
            counts.append(len(feats))
            augs.append((flip_type, angle))
    df = feature_metadata(counts, patient_id, label, dataset_name, modality, spatial_res, augmentations=augs)
    return all_features, all_masks, df


def save_metadata(df, df_path):
    df.to_parquet(df_path)


# ---- train_models.py:30-44, :143-182 --------------------------------------------------------------------------
def resize_mask_nearest(mask, out_hw):
    """skimage.transform.resize(mask, out_hw, order=0) of a boolean mask (nearest sample of the pixel-centre
    mapping, = scipy.ndimage.zoom(order=0, grid_mode=True), what current skimage calls)."""
    mask = np.asarray(mask)
    H, W = mask.shape
    oh, ow = out_hw
    yi = np.clip(np.floor((np.arange(oh) + 0.5) * H / oh - 0.5 + 0.5).astype(np.int64), 0, H - 1)
    xi = np.clip(np.floor((np.arange(ow) + 0.5) * W / ow - 0.5 + 0.5).astype(np.int64), 0, W - 1)
    return mask[yi][:, xi].astype(bool)


def masked_voxel_sequence(slice_features, slice_masks, spatial_res, noise=(0.0, 0.0, 0.0), out_dtype=torch.float32,
                          device=None):
    """The Stage-C input of one patient and modality, as the 'transformer' branch of PETCTDataset3D._get_features
    builds it (train_models.py:143-182): per-slice (h, w, D) feature maps + nodule masks -> [n, D] sequence of the
    masked voxels, each with its 3-D sinusoidal position code / 4 added (positional_encoding_3d, :30-44).

    Host side (integer / small float64 logic, restated): nearest mask resize, kept-voxel indices, coordinates — with
    the reference's np.meshgrid 'xy' flat order, mean-centring and `noise` offsets.  Device side
    (vdr_op_voxel_sequence): gather + sin / cos + add, float64 like numpy, rounded once to `out_dtype`.
    slice_features may be numpy arrays (uploaded once) or one device tensor [S, h, w, D]."""
    lib = L.load()
    if isinstance(slice_features, torch.Tensor):
        feats = slice_features
    else:
        feats = torch.from_numpy(np.stack([np.asarray(f, dtype=np.float32) for f in slice_features], axis=0))
    dev = device if device is not None else (feats.device if feats.is_cuda else "cuda")
    S, h, w, D = feats.shape
    vol = feats.to(dev, torch.float32).permute(1, 2, 0, 3).contiguous().view(-1, D)       # (h, w, S) positions
    masks = np.stack([resize_mask_nearest(m, (h, w)) for m in slice_masks], axis=0)
    keep = np.transpose(masks, (1, 2, 0)).reshape(-1)
    h0, w0 = np.asarray(slice_masks[-1]).shape[0:2]
    x, y, z = np.meshgrid(np.arange(0, h), np.arange(0, w), np.arange(0, S))              # 'xy' indexing, as the reference
    x = (x.flatten() / w).flatten() * w0 * spatial_res[0]
    y = (y.flatten() / h).flatten() * h0 * spatial_res[1]
    z = (z.flatten()).flatten() * spatial_res[2]
    xyz = np.stack([(x - x.mean() + noise[0])[keep], (y - y.mean() + noise[1])[keep], (z - z.mean() + noise[2])[keep]])
    index = np.flatnonzero(keep).astype(np.int64)
    n = int(index.shape[0])
    out = torch.empty((n, D), dtype=out_dtype, device=vol.device)
    if n == 0:
        return out
    expo = np.array([10000 ** (6 * i / D) for i in range(D // 6)], dtype=np.float64)
    d_index = torch.from_numpy(index).to(vol.device)
    d_xyz = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float64)).to(vol.device)
    d_expo = torch.from_numpy(expo).to(vol.device)
    odt = {torch.float32: L.VDR_F32, torch.bfloat16: L.VDR_BF16, torch.float64: L.VDR_F64}[out_dtype]
    L.check(lib.vdr_op_voxel_sequence(vol.data_ptr(), d_index.data_ptr(), d_xyz.data_ptr(), d_expo.data_ptr(), n, D,
                                      out.data_ptr(), odt, torch.cuda.current_stream(vol.device).cuda_stream))
    return out
