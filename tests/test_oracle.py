"""CPU: the oracle against the golden vectors.

postln_*.npz were produced by the REFERENCE's models_archs.TransformerNoduleClassifier
(tests/golden/make_golden.py); vit_hf_* / dinov2_hf_* by the in-container transformers
classes (architecture cross-check only).  Tolerance: fp32, max-abs <= 2e-5.
"""
import os

import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo

TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("tag", ["tiny", "refconf", "cfg1"])
def test_postln_matches_reference_class(golden_dir, tag):
    g = _load(golden_dir, f"postln_{tag}.npz")
    cfg = vo.postln_cfg(int(g["dim"]), int(g["heads"]), int(g["layers"]), int(g["ffn"]))
    w = vo.make_weights(cfg, seed=int(g["wseed"]), scale=float(g["wscale"]))
    x = vo.make_tokens(int(g["batch"]), int(g["seq"]), int(g["dim"]), seed=int(g["xseed"]))
    # the seeded generators reproduce what the fixture was made from
    np.testing.assert_array_equal(x[0, :2, :8].numpy(), g["x_probe"])
    np.testing.assert_array_equal(w["blocks.0.attn.qkv.weight"][:2, :8].numpy(), g["w_probe"])
    o = vo.forward_tokens(cfg, w, x)
    assert o["cls"].shape == (int(g["batch"]), int(g["dim"]))
    assert np.abs(o["cls"].numpy() - g["cls"]).max() <= TOL
    logits = vo.mlp_head(o["cls"], torch.from_numpy(g["head.classifier.dense1.weight"]),
                         torch.from_numpy(g["head.classifier.dense1.bias"]),
                         torch.from_numpy(g["head.classifier.dense2.weight"]),
                         torch.from_numpy(g["head.classifier.dense2.bias"]))
    assert np.abs(logits.numpy() - g["logits"]).max() <= TOL


@pytest.mark.parametrize("name", ["vit_hf_tiny", "vit_hf_p16", "dinov2_hf_tiny"])
def test_preln_matches_transformers_crosscheck(golden_dir, name):
    g = _load(golden_dir, name + ".npz")
    sw = name.startswith("dinov2")
    cfg = vo.VitCfg(int(g["img"]), int(g["patch"]), 3, int(g["dim"]), int(g["heads"]), int(g["layers"]),
                    int(g["ffn"]), act="swiglu" if sw else "gelu", layerscale=sw, ln_eps=1e-6)
    w = vo.make_weights(cfg, seed=int(g["wseed"]), scale=float(g["wscale"]))
    x = vo.make_images(cfg, int(g["batch"]), seed=int(g["xseed"]))
    o = vo.forward_images(cfg, w, x)
    assert np.abs(o["tokens"].numpy() - g["tokens"]).max() <= 5e-5
    np.testing.assert_array_equal(o["cls"].numpy(), o["tokens"][:, 0].numpy())
    np.testing.assert_array_equal(o["dense"].numpy(), o["tokens"][:, 1:].numpy())


def test_patch_embed_is_conv2d():
    cfg = vo.VitCfg(56, 14, 3, 32, 1, 1, 64)
    w = vo.make_weights(cfg, seed=5, scale=0.1)
    x = vo.make_images(cfg, 2, seed=1)
    ref = torch.nn.functional.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=14)
    ref = ref.flatten(2).transpose(1, 2)
    got = vo.patch_embed(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], 14)
    assert (got - ref).abs().max() <= 1e-5


def test_sdpa_matches_torch():
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 3, 37, 64) for _ in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert (vo.sdpa(q, k, v) - ref).abs().max() <= 1e-5


def test_emulated_bf16_close_to_fp32():
    cfg = vo.VitCfg(32, 8, 3, 64, 1, 2, 128)
    w = vo.make_weights(cfg, seed=2)
    x = vo.make_images(cfg, 2, seed=2)
    a = vo.forward_images(cfg, w, x)["cls"]
    b = vo.forward_images(cfg, w, x, emulate_bf16=True)["cls"]
    rel = ((a - b).norm() / a.norm()).item()
    assert rel < 2e-2


def test_flops_formula_matches_survey():
    # ViT-B/16 35.13 GFLOP/img, ViT-Ti/16 2.51, ViT-L/14@336 381.9, ViT-g/14 598.8
    assert abs(vo.flops_per_image(vo.CONFIGS["vit_base16_224"]) / 1e9 - 35.13) < 0.02
    assert abs(vo.flops_per_image(vo.CONFIGS["vit_tiny16_224"]) / 1e9 - 2.51) < 0.01
    assert abs(vo.flops_per_image(vo.CONFIGS["vit_large14_336"]) / 1e9 - 381.9) < 0.2
    assert abs(vo.flops_per_image(vo.CONFIGS["dinov2_giant14_224"]) / 1e9 - 598.8) < 0.3


@pytest.mark.parametrize("tag", ["tiny", "w7"])
def test_sam_encoder_matches_transformers_crosscheck(golden_dir, tag):
    """SAM / MedSAM image encoder restatement (windowed attention with zero padding after norm1,
    decomposed relative position bias, global blocks, conv neck) vs transformers.SamVisionModel."""
    from oracle import sam_oracle as so
    g = _load(golden_dir, f"sam_hf_{tag}.npz")
    cfg = so.SamCfg(int(g["img"]), int(g["patch"]), 3, int(g["dim"]), int(g["heads"]), int(g["layers"]), int(g["ffn"]),
                    int(g["window"]), tuple(int(i) for i in g["global_idx"]), int(g["out_chans"]), 1e-6)
    w = so.make_weights(cfg, seed=int(g["wseed"]), scale=float(g["wscale"]))
    x = so.make_images(cfg, int(g["batch"]), seed=int(g["xseed"]))
    o = so.sam_forward(cfg, w, x)
    assert o["out"].shape == g["out"].shape
    assert np.abs(o["out"].numpy() - g["out"]).max() <= 1e-4


def test_sam_flops_formula():
    from oracle import sam_oracle as so
    # SAM ViT-B @ 1024^2 is ~0.94 TFLOP per slice
    assert abs(so.flops_per_image(so.SAM_VIT_B) / 1e12 - 0.94) < 0.08


def test_mx_fp8_quantiser_matches_golden_table(golden_dir):
    """oracle/mx_oracle.py vs the committed torch.float8_e4m3fn fixture (SURVEY §8c golden (iv))."""
    from oracle import mx_oracle as mx

    g = np.load(os.path.join(golden_dir, "e4m3fn_table.npz"))
    mine = mx.e4m3_decode_table()
    assert np.array_equal(np.isnan(mine), np.isnan(g["decode"]))
    assert np.array_equal(np.nan_to_num(mine), np.nan_to_num(g["decode"]))
    x = torch.from_numpy(g["x"])
    q, e = mx.mx_quantize(x)
    assert np.array_equal(q.view(torch.uint8).numpy(), g["payload"])
    assert np.array_equal(e.numpy(), g["exponent"])
    assert np.array_equal(mx.mx_dequantize(q, e).numpy(), g["dequant"])
    # properties: never saturates, block maximum lands in (224, 448], zero blocks stay zero, half-ulp error bound
    qa = q.to(torch.float32).abs().reshape(x.shape[0], -1, 32).amax(-1)
    assert float(qa.max()) <= 448.0
    nz = x.reshape(x.shape[0], -1, 32).abs().amax(-1) > 0
    assert bool((qa[nz] > 224.0).all()) and bool((qa[~nz] == 0).all())
    d = mx.mx_dequantize(q, e)
    blockmax = x.reshape(x.shape[0], -1, 32).abs().amax(-1, keepdim=True).expand(-1, -1, 32).reshape(x.shape)
    assert bool(((d - x).abs() <= blockmax * 2.0 ** -4 * 1.0001 + 1e-30).all())


def test_prep_oracle_matches_reference_goldens(golden_dir):
    """oracle/prep_oracle.py vs vectors produced by the reference's own visualization_utils functions and the
    skimage calls of prepare_image (tests/golden/make_golden_prep.py)."""
    from oracle import prep_oracle as po

    g = np.load(os.path.join(golden_dir, "prep_resize.npz"))
    for name in list(g["names"]) + ["up_gray64"]:
        x, want = g[name + "_in"], g[name + "_out"]
        got = po.prepare_image(x, side=want.shape[-1])
        assert float(np.abs(got.astype(np.float64) - want).max()) <= (1e-5 if x.ndim == 3 else 1.2e-7), name
    h = np.load(os.path.join(golden_dir, "prep_hu.npz"))
    for a, b in (("hu", "rgb"), ("hu_i16", "rgb_i16"), ("hu_f32", "rgb_f32")):
        assert np.array_equal(po.hu_to_rgb(h[a]), h[b]), a
    r = np.load(os.path.join(golden_dir, "prep_roi.npz"))
    for i in range(int(r["n_cases"])):
        m, f = r[f"c{i}_mask"], r[f"c{i}_feat"]
        for mg in (1, 2):
            assert tuple(r[f"c{i}_coords_m{mg}"]) == po.extract_coords(m, mg)
        assert np.array_equal(po.extract_roi(f, m), r[f"c{i}_roi_feat"])
        assert np.array_equal(po.extract_roi(m, m), r[f"c{i}_roi_mask"])
        assert np.array_equal(po.crop_image(f, *[int(v) for v in r[f"c{i}_crop_args"]]), r[f"c{i}_crop"])
    wv = np.load(os.path.join(golden_dir, "prep_window.npz"))
    for tag, (w, l) in {"w800_l40": (800, 40), "w1500_lm600": (1500, -600), "w350_l50": (350, 50)}.items():
        for nm in ("i16", "f32"):
            assert np.array_equal(po.apply_window_ct(wv["ct_" + nm], w, l), wv[f"{tag}_{nm}"])


def test_rotate_oracle_is_bit_identical_to_scipy_and_goldens(golden_dir):
    """oracle/rotate_oracle.py restates scipy.ndimage.rotate(order 3, mode 'nearest', axes (0, 1), reshape False)
    operation for operation: bitwise equal to SciPy itself on float64 / float32 / bool volumes (square, non-square,
    4-D), and to the committed fixtures of the reference's rotate_image calls (make_golden_rotate.py)."""
    from scipy.ndimage import rotate
    from oracle import rotate_oracle as ro
    rng = np.random.default_rng(7)
    for shape, dt in (((40, 40, 3), np.float64), ((37, 52, 2), np.float64), ((30, 34, 2, 3), np.float64),
                      ((33, 41, 2), np.float32)):
        v = rng.random(shape).astype(dt)
        for ang in (45, 90, 135, 30):
            ref = rotate(v, ang, axes=(0, 1), reshape=False, mode="nearest")
            got = ro.rotate_planes(v, ang)
            assert got.dtype == ref.dtype and np.array_equal(ref, got), (shape, dt, ang)
    m = np.zeros((64, 56, 4), bool)
    m[20:40, 15:35, 1:3] = True
    m[5:9, 40:50, 0] = True
    for ang in (45, 90, 135):
        assert np.array_equal(rotate(m, ang, axes=(0, 1), reshape=False, mode="nearest"), ro.rotate_planes(m, ang))
    g = np.load(os.path.join(golden_dir, "rotate_cases.npz"))
    for name in g["names"]:
        img, mask = g[f"{name}_img"], g[f"{name}_mask"]
        for ang in (45, 90, 135):
            ri, rm = ro.rotate_image(img, mask, ang)
            assert np.array_equal(ri, g[f"{name}_img_{ang}"]) and np.array_equal(rm, g[f"{name}_mask_{ang}"]), (name, ang)
    ri, rm = ro.rotate_image(img, mask, 0)
    assert np.array_equal(ri, img) and np.array_equal(rm, mask)


@pytest.mark.parametrize("tag", ["tiny", "refdim"])
def test_bimodal_oracle_matches_reference_class(golden_dir, tag):
    """oracle/bimodal_oracle.py vs the outputs of the reference's own TransformerNoduleBimodalClassifier
    (models_archs.py:38-124; tests/golden/make_golden_bimodal.py), all three modality modes, <= 2e-5."""
    from oracle import bimodal_oracle as bo
    g = np.load(os.path.join(golden_dir, f"bimodal_{tag}.npz"), allow_pickle=False)
    dim, lc, lp = int(g["dim"]), int(g["layers_ct"]), int(g["layers_pet"])
    fc, fp = int(float(g["ratio_ct"]) * dim), int(float(g["ratio_pet"]) * dim)
    sd = bo.make_state_dict(dim, fc, fp, lc, lp, int(g["classes"]), seed=int(g["seed"]))
    assert np.array_equal(sd["cross_attention_ct.multihead_attn.in_proj_weight"][:2, :8].numpy(), g["w_probe"])
    x_ct, x_pet = torch.from_numpy(g["x_ct"]), torch.from_numpy(g["x_pet"])
    for mode, (a, b) in (("both", (x_ct, x_pet)), ("ct", (x_ct, None)), ("pet", (None, x_pet))):
        out = bo.forward(sd, dim, fc, fp, int(g["heads_ct"]), int(g["heads_pet"]), lc, lp, a, b)
        for name, o in zip(("logits_petct", "cls_petct", "logits_ct", "logits_pet"), out):
            want = torch.from_numpy(g[f"{mode}_{name}"])
            assert o.shape == want.shape and (o - want).abs().max().item() < 2e-5, (mode, name)


def test_voxel_sequence_oracle_matches_goldens_and_zoom(golden_dir):
    """Stage-C input builder (train_models.py:30-44, :143-182): the oracle reproduces the fixtures made with the
    reference's numpy / skimage calls exactly (kept voxels, float64 sequences); its nearest mask resize equals
    scipy.ndimage.zoom(order=0, grid_mode=True) — what current skimage calls — also where samples fall on ties."""
    from scipy import ndimage as ndi
    from oracle import prep_oracle as po
    g = np.load(os.path.join(golden_dir, "sequence_cases.npz"))
    for n in g["names"]:
        seq, keep = po.masked_voxel_sequence(list(g[f"{n}_feats"]), list(g[f"{n}_masks"]), g[f"{n}_res"], g[f"{n}_noise"])
        assert np.array_equal(keep, g[f"{n}_keep"]) and seq.dtype == np.float64
        assert np.array_equal(seq, g[f"{n}_seq"]), n
    rng = np.random.default_rng(0)
    for H, W, h, w in ((40, 40, 12, 12), (64, 64, 10, 10), (64, 64, 16, 16), (33, 51, 9, 14), (11, 9, 20, 18), (30, 50, 15, 25)):
        m = rng.random((H, W)) > 0.5
        assert np.array_equal(po.resize_mask_nearest(m, (h, w)), ndi.zoom(m, (h / H, w / W), order=0, mode="mirror", grid_mode=True))
    pe = po.positional_encoding_3d([1.0], [2.0], [3.0], 50)   # D % 3 == 2: the z block starts at (2 D) // 3 = 33
    assert pe[0, 16] == np.sin(2.0) and pe[0, 33] == np.sin(3.0) and pe[0, 32] == 0.0
