"""GPU pre/post-processing against golden vectors made from the reference's own
numpy helpers and the skimage calls of prepare_image (tests/golden/make_golden_prep.py), through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import prep_oracle as po

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_prepare_image_matches_skimage_golden(golden_dir):
    """gray2rgb + resize(order 1, 'reflect', anti-aliasing) + CHW: <= 2 float32 ulp for gray inputs, 1e-5 for the
    3-channel path (skimage's own multichannel warp keeps float32 intermediates)."""
    from vdr import prep
    g = _load(golden_dir, "prep_resize.npz")
    for name in list(g["names"]) + ["up_gray64"]:
        x, want = g[name + "_in"], g[name + "_out"]
        side = want.shape[-1]
        vol = torch.from_numpy(x).unsqueeze(2)
        got = prep.prepare_slices(vol, side=side)[0].cpu().numpy()
        tol = 1e-5 if x.ndim == 3 else 3e-7
        err = float(np.abs(got.astype(np.float64) - want).max())
        assert got.shape == want.shape and err <= tol, (name, err)


def test_prepare_slices_volume_flips_bf16():
    """all slices of a strided (H, W, S) volume in one launch == slice by slice; flips folded into the gather ==
    flipping first (flip_image, tfds_dense_descriptor.py:305-324); bf16 output = rounded fp32 output."""
    from vdr import prep
    rng = np.random.default_rng(5)
    vol = rng.random((45, 38, 7)).astype(np.float32)
    ref = np.stack([po.prepare_image(vol[:, :, i], side=96) for i in range(7)])
    got = prep.prepare_slices(vol, side=96)
    assert float(np.abs(got.cpu().numpy() - ref).max()) <= 3e-7
    for flip, fl in (("horizontal", vol[:, ::-1]), ("vertical", vol[::-1])):
        a = prep.prepare_slices(vol, side=96, flip=flip).cpu().numpy()
        b = np.stack([po.prepare_image(np.ascontiguousarray(fl[:, :, i]), side=96) for i in range(7)])
        assert float(np.abs(a - b).max()) <= 3e-7, flip
    bf = prep.prepare_slices(vol, side=96, out_dtype=torch.bfloat16)
    assert torch.equal(bf, got.to(torch.bfloat16))
    # colour volume (H, W, S, 3) and a down-scaled one (anti-aliasing path)
    col = rng.random((70, 66, 3, 3)).astype(np.float32)
    ref = np.stack([po.prepare_image(col[:, :, i], side=40) for i in range(3)])
    got = prep.prepare_slices(col, side=40).cpu().numpy()
    assert float(np.abs(got - ref).max()) <= 2e-6
    # the reference's default sides
    assert prep.prepare_image(vol[:, :, 0]).shape == (1, 3, 1024, 1024)
    assert prep.prepare_image(col[:, :, 0]).shape == (1, 3, 896, 896)


def test_window_ct_and_hu_colormap_bit_exact(golden_dir):
    from vdr import prep
    g = _load(golden_dir, "prep_window.npz")
    for tag, (w, l) in {"w800_l40": (800, 40), "w1500_lm600": (1500, -600), "w350_l50": (350, 50)}.items():
        got = prep.apply_window_ct(g["ct_f32"], w, l).cpu().numpy()
        assert np.array_equal(got, g[f"{tag}_f32"]), tag                      # float32 in: numpy computes in float32
        got = prep.apply_window_ct(g["ct_i16"], w, l).cpu().numpy()
        assert np.array_equal(got, g[f"{tag}_i16"].astype(np.float32)), tag   # int16 in: float64, stored as fp32
    h = _load(golden_dir, "prep_hu.npz")
    for a, b in (("hu", "rgb"), ("hu_i16", "rgb_i16"), ("hu_f32", "rgb_f32")):
        got = prep.hu_to_rgb_vectorized(h[a]).cpu().numpy()
        assert got.dtype == np.uint8 and np.array_equal(got, h[b]), a
    # NaN HU: every one of the reference's nine masks is False, the pixel stays at np.zeros' (0, 0, 0)
    for dt in (np.float32, np.float64):
        v = np.array([np.nan, 500.0, -2000.0, np.nan], dtype=dt)
        got = prep.hu_to_rgb_vectorized(v).cpu().numpy()
        assert np.array_equal(got, np.array([[0, 0, 0], [255, 255, 255], [0, 0, 0], [0, 0, 0]], dtype=np.uint8))


def test_crop_maps_matches_reference_roi(golden_dir):
    """extract_roi / crop_image of a feature map on the device == the reference's numpy crops."""
    from vdr import pipeline
    g = _load(golden_dir, "prep_roi.npz")
    for i in range(int(g["n_cases"])):
        mask, feat = g[f"c{i}_mask"], g[f"c{i}_feat"]
        maps = torch.from_numpy(np.stack([feat, feat * 2])).cuda()
        got = pipeline.crop_maps(maps, pipeline.roi_box(feat.shape[0:2], mask)).cpu().numpy()
        assert np.array_equal(got[0], g[f"c{i}_roi_feat"]) and np.array_equal(got[1], g[f"c{i}_roi_feat"] * 2), i
        got = pipeline.crop_maps(maps, tuple(int(v) for v in g[f"c{i}_crop_args"])).cpu().numpy()
        assert np.array_equal(got[0], g[f"c{i}_crop"]), i


def test_generate_features_batched_pipeline_vs_oracle():
    """The per-slice hot loop (tfds_dense_descriptor.py:242-284) as one batched GPU pipeline on a small SAM
    geometry: same list lengths, same crop shapes, feature maps within the bf16 gate of the oracle pipeline
    (numpy prepare_image -> fp32 SAM oracle -> extract_roi)."""
    import vdr
    from vdr import pipeline
    from oracle import sam_oracle as so
    cfg = so.SamCfg(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, window=7, global_idx=(1,), out_chans=64)
    w = so.make_weights(cfg, seed=21)
    vc = vdr.VdrConfig(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, has_cls=False, window=7,
                       global_blocks=(1,), neck_chans=64)
    model = vdr.VitDescriptorModel(vc, w, "medsam", torch.device("cuda"))
    model.model_name = "medsam"
    rng = np.random.default_rng(8)
    H, W, S = 72, 80, 5
    img = rng.random((H, W, S)).astype(np.float32)
    mask = np.zeros((H, W, S), dtype=bool)
    mask[30:41, 36:50, 1:4] = True
    feats, masks = pipeline.generate_features(model, img, mask, max_batch=3)
    assert len(feats) == S and len(masks) == S
    # oracle pipeline, slice by slice, following the reference loop
    bigger = mask.sum(-1) > 0
    xmin, ymin, xmax, ymax = po.extract_coords(bigger, 2)
    cs = max(xmax - xmin, ymax - ymin) * 2
    xm, ym = int(xmin + (xmax - xmin) / 2), int(ymin + (ymax - ymin) / 2)
    box = (xm - cs, ym - cs, xm + cs, ym + cs)
    img_c, mask_c, big_c = po.crop_image(img, *box), po.crop_image(mask, *box), po.crop_image(bigger, *box)
    for i in range(S):
        x = torch.from_numpy(po.prepare_image(img_c[:, :, i], side=224))[None]
        f = so.sam_forward(cfg, w, x)["out"][0].permute(1, 2, 0).numpy()
        want = po.extract_roi(f, big_c)
        assert feats[i].shape == want.shape and feats[i].dtype == np.float32
        rel = np.linalg.norm(feats[i] - want) / np.linalg.norm(want)
        assert rel < 2e-2, (i, rel)
        assert np.array_equal(masks[i], po.extract_roi(mask_c[:, :, i] > 0, big_c))


@pytest.mark.parametrize("flip", ["horizontal", "vertical"])
def test_generate_features_flip_equals_explicitly_flipped_volume(flip):
    """generate_features(flip=...) == generate_features on flip_image(img, mask) (tfds_dense_descriptor.py:305-324,
    463-467): the nodule sits off-centre and near a border, so the asymmetric-margin boxes of the flipped mask differ
    from the mirrored boxes of the unflipped one, and the clamped crop window is not symmetric either."""
    import vdr
    from vdr import pipeline, prep
    from oracle import sam_oracle as so
    cfg = so.SamCfg(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, window=7, global_idx=(1,), out_chans=64)
    w = so.make_weights(cfg, seed=23)
    vc = vdr.VdrConfig(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, has_cls=False, window=7,
                       global_blocks=(1,), neck_chans=64)
    model = vdr.VitDescriptorModel(vc, w, "medsam", torch.device("cuda"))
    model.model_name = "medsam"
    rng = np.random.default_rng(10)
    H, W, S = 70, 90, 4
    img = rng.random((H, W, S)).astype(np.float32)
    mask = np.zeros((H, W, S), dtype=bool)
    mask[6:19, 60:83, 1:3] = True          # near the top-right corner: the 4 x box window is clamped on two sides
    mask[9, 58, 2] = True
    fi, fm = prep.flip_image(img, mask, flip)
    want_f, want_m = pipeline.generate_features(model, np.ascontiguousarray(fi), np.ascontiguousarray(fm))
    got_f, got_m = pipeline.generate_features(model, img, mask, flip=flip)
    assert len(got_f) == len(want_f) == S
    for i in range(S):
        assert got_f[i].shape == want_f[i].shape and np.array_equal(got_f[i], want_f[i]), i
        assert got_m[i].shape == want_m[i].shape and np.array_equal(got_m[i], want_m[i]), i
    # the boxes really come from the flipped mask: the mask crops are the mirror images of nothing the unflipped call
    # returns only by accident of symmetry, but the feature crops must differ from the unflipped ones
    plain_f, _ = pipeline.generate_features(model, img, mask)
    assert any(a.shape != b.shape or not np.array_equal(a, b) for a, b in zip(plain_f, got_f))
    with pytest.raises(ValueError):
        pipeline.generate_features(model, img, mask, flip="diagonal")


def test_extract_patient_features_augmentation_loop():
    """The per-patient loop (tfds_dense_descriptor.py:452-491) on a small SAM geometry: 3 flips x 2 angles, list
    lengths, metadata rows in loop order, and the flipped pass equal to running the flipped volume explicitly."""
    import vdr
    from vdr import pipeline
    from oracle import sam_oracle as so
    cfg = so.SamCfg(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, window=7, global_idx=(1,), out_chans=64)
    w = so.make_weights(cfg, seed=22)
    vc = vdr.VdrConfig(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, has_cls=False, window=7,
                       global_blocks=(1,), neck_chans=64)
    model = vdr.VitDescriptorModel(vc, w, "medsam", torch.device("cuda"))
    model.model_name = "medsam"
    rng = np.random.default_rng(9)
    H, W, S = 64, 60, 4
    img = rng.random((H, W, S)).astype(np.float32)
    mask = np.zeros((H, W, S), dtype=bool)
    mask[20:33, 25:41, 1:3] = True
    feats, masks, df = pipeline.extract_patient_features(model, img, mask, "P1", 0, "stanford_dataset", "pet",
                                                         np.array([0.8, 0.8, 0.8]), angles=(0, 90))
    assert len(feats) == len(masks) == len(df) == 3 * 2 * S
    assert list(df["flip"][::2 * S]) == [None, "horizontal", "vertical"] and list(df["angle"][::S][:2]) == [0, 90]
    assert list(df["feature_id"]) == list(range(len(df))) and set(df["dataset"]) == {"stanford"}
    # horizontal flip, angle 0 == generate_features on the explicitly flipped volume
    f2, m2 = pipeline.generate_features(model, np.ascontiguousarray(img[:, ::-1]), np.ascontiguousarray(mask[:, ::-1]))
    for i in range(S):
        assert np.array_equal(feats[2 * S + i], f2[i]) and np.array_equal(masks[2 * S + i], m2[i])


def test_rotate_volume_bit_identical_to_scipy_and_goldens(golden_dir):
    """rotate_image (tfds_dense_descriptor.py:327-350) on the GPU: bitwise equal to the committed fixtures of the
    reference's SciPy calls, to the oracle, and to scipy.ndimage.rotate itself (float64 / float32 / bool volumes,
    non-square planes, a 4-D colour volume, an arbitrary angle, a single plane)."""
    from scipy.ndimage import rotate
    from vdr import prep, pipeline
    from oracle import rotate_oracle as ro
    g = _load(golden_dir, "rotate_cases.npz")
    for name in g["names"]:
        img, mask = g[f"{name}_img"], g[f"{name}_mask"]
        for ang in (45, 90, 135):
            ri, rm = pipeline.rotate_image(img, mask, ang)  # numpy in -> numpy out
            assert ri.dtype == img.dtype and rm.dtype == np.bool_
            assert np.array_equal(ri, g[f"{name}_img_{ang}"]), (name, ang)
            assert np.array_equal(rm, g[f"{name}_mask_{ang}"]), (name, ang)
    rng = np.random.default_rng(11)
    for shape, dt in (((96, 96, 17), np.float64), ((70, 101, 5), np.float64), ((50, 47, 3, 3), np.float64),
                      ((64, 80, 9), np.float32), ((33, 35, 1), np.float64)):
        v = rng.random(shape).astype(dt)
        for ang in (45, 135, 30):
            want = rotate(v, ang, axes=(0, 1), reshape=False, mode="nearest")
            got = prep.rotate_volume(v, ang).cpu().numpy()
            assert got.dtype == want.dtype and np.array_equal(got, want), (shape, dt, ang)
            assert np.array_equal(got, ro.rotate_planes(v, ang))
    m = np.zeros((90, 84, 6), bool)
    m[30:61, 22:50, 1:5] = True
    m[8:14, 60:75, 0] = True
    for ang in (45, 90, 135):
        want = rotate(m, ang, axes=(0, 1), reshape=False, mode="nearest")
        got = prep.rotate_volume(m, ang).cpu().numpy()
        assert got.dtype == np.bool_ and np.array_equal(got, want), ang
    # angle 0: copies; device tensors in -> device tensors out
    img = torch.rand(20, 20, 2, dtype=torch.float64, device="cuda")
    msk = torch.zeros(20, 20, 2, dtype=torch.bool, device="cuda")
    ri, rm = pipeline.rotate_image(img, msk, 0)
    assert ri.is_cuda and torch.equal(ri, img) and ri.data_ptr() != img.data_ptr() and torch.equal(rm, msk)
    ri, rm = pipeline.rotate_image(img, msk, 90)
    assert ri.is_cuda and rm.is_cuda and rm.dtype == torch.bool
    # other angles SciPy accepts (negative, > 180, fractional), the smallest plane, and the refusal of integer volumes
    v = rng.random((31, 29, 2))
    for ang in (-30, 180, 270, 12.5):
        assert np.array_equal(prep.rotate_volume(v, ang).cpu().numpy(), rotate(v, ang, axes=(0, 1), reshape=False, mode="nearest")), ang
    tiny = rng.random((2, 2, 3))
    assert np.array_equal(prep.rotate_volume(tiny, 45).cpu().numpy(), rotate(tiny, 45, axes=(0, 1), reshape=False, mode="nearest"))
    with pytest.raises(TypeError):
        prep.rotate_volume(np.zeros((8, 8, 2), np.int16), 45)


def test_rotate_volume_ct_sized_slice_stack():
    """A CT-sized stack (512 x 512 x 24 float64 in [0, 1] + its nodule mask): image and mask bitwise equal to
    SciPy; clip01 == np.clip of the reference."""
    from scipy.ndimage import rotate
    from vdr import prep
    rng = np.random.default_rng(12)
    vol = rng.random((512, 512, 24))
    yy, xx = np.mgrid[0:512, 0:512]
    mask = np.repeat((((yy - 300) ** 2 + (xx - 210) ** 2) < 40 ** 2)[:, :, None], 24, axis=2)
    want = np.clip(rotate(vol, 45, axes=(0, 1), reshape=False, mode="nearest"), 0, 1)
    got = prep.rotate_volume(vol, 45, clip01=True).cpu().numpy()
    assert np.array_equal(got, want)
    wm = rotate(mask, 45, axes=(0, 1), reshape=False, mode="nearest") > 0
    gm = prep.rotate_volume(mask, 45).cpu().numpy()
    assert np.array_equal(gm, wm) and 0 < gm.sum() < mask.sum()  # the truncating bool store thins the mask


def test_extract_patient_features_rotations_match_scipy_path():
    """The augmentation loop with a non-trivial angle: the GPU-rotated pass equals generate_features on the volume
    SciPy rotates on the CPU (what the reference does), bit for bit."""
    import vdr
    from scipy.ndimage import rotate
    from vdr import pipeline
    from oracle import sam_oracle as so
    cfg = so.SamCfg(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, window=7, global_idx=(1,), out_chans=64)
    w = so.make_weights(cfg, seed=23)
    vc = vdr.VdrConfig(img=224, patch=16, dim=128, heads=2, layers=2, mlp_hidden=256, has_cls=False, window=7,
                       global_blocks=(1,), neck_chans=64)
    model = vdr.VitDescriptorModel(vc, w, "medsam", torch.device("cuda"))
    model.model_name = "medsam"
    rng = np.random.default_rng(10)
    H, W, S = 64, 60, 4
    img = rng.random((H, W, S))
    mask = np.zeros((H, W, S), dtype=bool)
    mask[18:38, 22:44, 1:3] = True
    feats, masks, df = pipeline.extract_patient_features(model, img, mask, "P2", 1, "stanford_dataset", "ct",
                                                         np.array([1.0, 1.0, 1.0]), flips=(None,), angles=(0, 45))
    assert len(feats) == 2 * S and list(df["angle"]) == [0] * S + [45] * S
    im45 = np.clip(rotate(img, 45, axes=(0, 1), reshape=False, mode="nearest"), 0, 1)
    m45 = rotate(mask, 45, axes=(0, 1), reshape=False, mode="nearest") > 0
    f2, m2 = pipeline.generate_features(model, im45, m45)
    for i in range(S):
        assert np.array_equal(feats[S + i], f2[i]) and np.array_equal(masks[S + i], m2[i])


def test_masked_voxel_sequence_matches_reference_goldens(golden_dir):
    """Stage-C input builder on the GPU (vdr_op_voxel_sequence) against the fixtures of the reference's numpy /
    skimage calls: same kept voxels; float64 output within 4 ulp-level 2e-15 of numpy (device sin / cos), float32
    output = the float64 result rounded once (<= 1 float32 ulp), and the sequence feeds the classifier."""
    from vdr import pipeline
    g = _load(golden_dir, "sequence_cases.npz")
    for n in g["names"]:
        feats, masks, want = list(g[f"{n}_feats"]), list(g[f"{n}_masks"]), g[f"{n}_seq"]
        s64 = pipeline.masked_voxel_sequence(feats, masks, g[f"{n}_res"], g[f"{n}_noise"], out_dtype=torch.float64)
        assert s64.shape == want.shape and s64.is_cuda
        err = float(np.abs(s64.cpu().numpy() - want).max())
        assert err <= 2e-15 * max(1.0, float(np.abs(want).max())), (n, err)
        s32 = pipeline.masked_voxel_sequence(feats, masks, g[f"{n}_res"], g[f"{n}_noise"])
        w32 = want.astype(np.float32)
        assert s32.dtype == torch.float32
        assert np.abs(s32.cpu().numpy() - w32).max() <= np.spacing(np.abs(w32).max()), n
        dev_in = pipeline.masked_voxel_sequence(torch.from_numpy(np.stack(feats)).cuda(), masks, g[f"{n}_res"], g[f"{n}_noise"])
        assert torch.equal(dev_in, s32)
    # a D % 3 == 2 width: the z block starts at (2 D) // 3
    from oracle import prep_oracle as po
    rng = np.random.default_rng(3)
    feats = [rng.standard_normal((6, 7, 50)).astype(np.float32) for _ in range(2)]
    masks = [rng.random((13, 15)) > 0.4 for _ in range(2)]
    want, keep = po.masked_voxel_sequence(feats, masks, (0.9, 1.1, 2.0), (0.1, -0.2, 0.3))
    got = pipeline.masked_voxel_sequence(feats, masks, (0.9, 1.1, 2.0), (0.1, -0.2, 0.3), out_dtype=torch.float64)
    assert got.shape == want.shape and np.abs(got.cpu().numpy() - want).max() <= 1e-14
    # nothing kept -> empty sequence
    empty = pipeline.masked_voxel_sequence(feats, [np.zeros((13, 15), bool)] * 2, (1, 1, 1))
    assert empty.shape == (0, 50)
