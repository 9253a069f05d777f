"""CPU: host-side logic — state_dict translation, expected weight lists, shard arithmetic, and the
world_size-2 gloo all-gather that reassembles the feature matrix."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vit_oracle as vo


def test_expected_weight_shapes_match_oracle_for_every_named_config():
    import vdr
    from vdr.weights import expected_weight_shapes
    for name, ocfg in vo.CONFIGS.items():
        vcfg = vdr.ARCHS[name]
        assert expected_weight_shapes(vcfg) == vo.weight_shapes(ocfg), name
        assert vcfg.n_tokens == ocfg.n_tokens
    # model_name 'dinov2' is what the reference runs of that backbone: model.patch_embed only (tfds_dense_descriptor.py:128)
    assert sorted(expected_weight_shapes(vdr.ARCHS["dinov2"])) == ["patch_embed.proj.bias", "patch_embed.proj.weight"]
    assert vdr.ARCHS["dinov2"].n_patches == 4096


def test_torch_encoder_state_dict_translation():
    """vdr.weights.from_torch_encoder_state_dict (the product's key translation for models_archs.py:127-139 state_dicts)
    checked against torch itself: a live nn.TransformerEncoder (post-LN, GELU, the reference's construction) with a cls
    token and an input LayerNorm is run on CPU, its state_dict goes through the PRODUCT's translation, and the oracle's
    forward on the translated weights must reproduce torch's output -- a swapped q/k/v block, a transposed matrix or a
    norm1 / norm2 mix-up cannot pass."""
    from vdr.weights import from_torch_encoder_state_dict
    torch.manual_seed(3)
    D, H, L, F, B, S = 64, 1, 2, 128, 3, 7
    layer = torch.nn.TransformerEncoderLayer(d_model=D, dim_feedforward=F, nhead=H, activation="gelu", batch_first=True, dropout=0.0)
    enc = torch.nn.TransformerEncoder(layer, num_layers=L, enable_nested_tensor=False).eval()
    norm = torch.nn.LayerNorm(D)
    with torch.no_grad():
        for p in list(enc.parameters()) + list(norm.parameters()):
            p.copy_(torch.randn_like(p) * (0.2 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 and p.shape[0] == D else 0.0))
    cls = torch.randn(1, 1, D)
    sd = {"transformer_encoder." + k: v for k, v in enc.state_dict().items()}
    sd["cls_token"] = cls
    sd["norm.weight"], sd["norm.bias"] = norm.weight.detach(), norm.bias.detach()
    mine = from_torch_encoder_state_dict(sd, L)
    assert sorted(mine) == sorted(vo.weight_shapes(vo.postln_cfg(D, H, L, F)))
    x = torch.randn(B, S, D)
    with torch.no_grad():
        want = enc(norm(torch.cat([cls.repeat(B, 1, 1), x], dim=1)))[:, 0, :]   # models_archs.py:141-147
    got = vo.forward_tokens(vo.postln_cfg(D, H, L, F), mine, x)["cls"]
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())


def test_shard_bounds_cover_rows_in_order():
    from vdr.dist import shard_bounds
    for total in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, D, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vdr.dist import extract_features_sharded, shard_bounds
        full = torch.arange(total * D, dtype=torch.float32).reshape(total, D) * 0.5 - 3.0
        lo, hi = shard_bounds(total, rank, world)
        calls = []

        def fake_extract(local):  # stands in for the per-rank HIP forward: identity on "features"
            calls.append(local.shape[0])
            return local.clone()

        out = extract_features_sharded(fake_extract, full[lo:hi], total)
        ok = torch.equal(out, full) and out.is_contiguous() and calls == [hi - lo]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])  # equal shards (single all_gather_into_tensor) and a ragged tail
def test_world2_gloo_allgather_reassembles_rows_bitwise(total):
    import vdr  # noqa: F401  (path set by conftest)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (3, 8)])  # equal shards, ragged tails, a world that does not divide N
def test_launch_ranks_gloo_allgather_bitwise(tmp_path, world, total):
    """The launcher bench.py uses for `--gpus N` (vdr.dist.launch_ranks -> torch.distributed.run children) with the
    gloo backend: every rank's gathered matrix equals the unsharded one, rows in dataset order."""
    import json
    from vdr.dist import launch_ranks
    out = str(tmp_path / "res")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    rc = launch_ranks(worker, world, ["--backend", "gloo", "--total", total, "--out", out], timeout=300)
    assert rc == 0
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    assert [r["rank"] for r in res] == list(range(world))
    assert all(r["ok"] and r["world"] == world and r["group_size"] == world and r["backend"] == "gloo" for r in res)


def test_bench_plain_multi_gpu_command_line_starts_the_ranks():
    """`python bench.py --gpus 2` started plainly must itself start 2 ranks (here, without a GPU, each rank stops at
    the 'needs an MI355X' check and the launcher reports failure) instead of running one rank and printing n_gpus 1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box this test stays a CPU test
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    # torch.distributed.run stops the surviving rank as soon as the first one fails: one message is guaranteed, and its
    # failure report names both local ranks
    assert "bench.py needs an MI355X" in r.stderr, r.stderr[-2000:]
    assert "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus"' not in r.stdout


def test_pipeline_roi_host_logic_matches_reference_goldens(golden_dir):
    """vdr.pipeline's host-side box maths (extract_coords / extract_roi / crop_image) vs the reference's own
    functions (tests/golden/prep_roi.npz)."""
    import os
    import numpy as np
    from vdr import pipeline

    r = np.load(os.path.join(golden_dir, "prep_roi.npz"))
    for i in range(int(r["n_cases"])):
        m, f = r[f"c{i}_mask"], r[f"c{i}_feat"]
        for mg in (1, 2):
            assert tuple(r[f"c{i}_coords_m{mg}"]) == pipeline.extract_coords(m, mg)
        assert np.array_equal(pipeline.extract_roi(f, m), r[f"c{i}_roi_feat"])
        assert np.array_equal(pipeline.extract_roi(m, m), r[f"c{i}_roi_mask"])
        assert np.array_equal(pipeline.crop_image(f, *[int(v) for v in r[f"c{i}_crop_args"]]), r[f"c{i}_crop"])
    with __import__("pytest").raises(ValueError):
        pipeline.extract_coords(np.zeros((4, 4), dtype=bool), 1)


def test_feature_metadata_table_layout(tmp_path):
    """The per-patient parquet the reference writes next to the HDF5 features (tfds_dense_descriptor.py:452-491):
    column names / order, feature_id numbering across augmentations, the 3 x 4 augmentation loop order, and a
    parquet round trip (pandas + pyarrow are in the image)."""
    import numpy as np
    import pandas as pd
    from vdr import pipeline

    res = np.array([0.8, 0.8, 0.8])
    df = pipeline.feature_metadata(5, "LUNG-001", 1, "santa_maria_dataset", "ct", res)
    assert list(df.columns) == ["feature_id", "slice", "angle", "flip", "patient_id", "label", "dataset", "modality",
                                "augmentation", "spatial_res"]
    assert len(df) == 5 * 12 and list(df["feature_id"]) == list(range(60))
    assert list(df["slice"][:7]) == [0, 1, 2, 3, 4, 0, 1]
    assert list(df["angle"][::5]) == [0, 45, 90, 135] * 3
    assert [f for f in df["flip"][::20]] == [None, "horizontal", "vertical"]
    assert set(df["dataset"]) == {"santa_maria"} and bool(df["augmentation"].all())
    path = tmp_path / "LUNG-001_ct.parquet"
    pipeline.save_metadata(df, str(path))
    back = pd.read_parquet(path)
    assert list(back.columns) == list(df.columns) and len(back) == 60
    assert np.allclose(np.asarray(back["spatial_res"][0], dtype=float), res)


def test_device_gelu_polynomial_matches_erf_gelu():
    """The GEMM epilogue's erf-GELU (csrc/vdr_dev.h: max(x, 0) - a 2^Q(a), Q a degree-6 polynomial) restated in numpy
    float32 from the coefficients in the device header, against the exact 0.5 x (1 + erf(x / sqrt 2)) in float64: the
    relative error stays far below the bf16 rounding (2e-3) that follows it, the tail below the test's 5e-7."""
    import re
    from scipy.special import erf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "vit-deep-radiomics_amd", "csrc", "vdr_dev.h")).read()
    body = src[src.index("VDR_DEV float gelu_erf(float x) {"):]
    body = body[:body.index("}")]
    coef = [np.float32(v) for v in re.findall(r"(-?\d\.\d+e[+-]\d+)f", body)]
    assert len(coef) == 7 and "5.7f" in body
    x = np.concatenate([np.linspace(-12, 12, 400001), np.array([-100.0, 100.0, 0.0])]).astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(5.7))
    q = coef[0]
    for c in coef[1:]:
        q = (q * a + c).astype(np.float32)
    got = (np.maximum(x, 0) - a * np.exp2(q.astype(np.float64)).astype(np.float32)).astype(np.float64)
    ref = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    err = np.abs(got - ref)
    assert (err <= 1e-4 * np.abs(ref) + 3e-7).all(), float((err / (np.abs(ref) + 3e-3)).max())
    assert err.max() < 8e-6


def test_source_id_names_the_kernel_sources(tmp_path, monkeypatch):
    """vdr.source_id() is what ties a committed PMC profile to the kernels it was taken with (bench.py drops
    `roofline.traffic` when the ids differ): 16 hex digits, the same on every call, different as soon as one byte of a
    file under csrc/ differs."""
    import shutil
    import vdr
    from vdr import _lib
    sid = vdr.source_id()
    assert len(sid) == 16 and int(sid, 16) >= 0 and vdr.source_id() == sid
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(_lib.__file__)))
    clone = tmp_path / "pkg"
    (clone / "vdr").mkdir(parents=True)
    shutil.copytree(os.path.join(pkg, "csrc"), clone / "csrc", ignore=shutil.ignore_patterns("build*"))
    monkeypatch.setattr(_lib, "__file__", str(clone / "vdr" / "_lib.py"))
    assert _lib.source_id() == sid
    with open(clone / "csrc" / "vdr_dev.h", "a") as f:
        f.write("\n")
    assert _lib.source_id() != sid
