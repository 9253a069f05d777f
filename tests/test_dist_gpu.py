"""GPU: the RCCL path of the batch-shard design (umap_cls_token.py:133-139 stacks the [N, D]
matrix in dataset order).  Ranks are started with vdr.dist.launch_ranks, the launcher `bench.py --gpus N` uses;
each rank runs the HIP forward on its shard, the shards are all-gathered over RCCL, and the result must be BITWISE
the 1-rank forward (pure gather, no reduction)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")


def _run(tmp_path, world, total, backend="nccl", extra=()):
    from vdr.dist import launch_ranks
    out = str(tmp_path / "res")
    rc = launch_ranks(WORKER, world, ["--backend", backend, "--total", total, "--out", out, *extra], timeout=600)
    res = [json.load(open(f"{out}.{r}")) for r in range(world) if os.path.exists(f"{out}.{r}")]
    assert rc == 0, res
    assert all(r["ok"] and r["world"] == world and r["group_size"] == world and r["backend"] == backend for r in res), res
    return res


def test_rccl_single_rank_process_group(tmp_path):
    """One rank under the launcher: RCCL initialises, all_gather_into_tensor runs, rows equal the plain forward."""
    _run(tmp_path, 1, 8)


@pytest.mark.parametrize("total", [8, 7])  # equal shards (one in-place collective) and a ragged tail
def test_rccl_two_rank_allgather_bitwise_equal_to_one_rank(tmp_path, total):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one rank per GPU); the same code runs under gloo in tests/test_host_logic.py")
    _run(tmp_path, 2, total)


@pytest.mark.parametrize("world,total", [(2, 8), (3, 8)])
def test_shared_gpu_gloo_rehearsal_of_the_overlapped_step(tmp_path, world, total):
    """2 / 3 rank processes share the box's one GPU and run the HIP forward; gloo carries the device rows (host reads
    of VRAM, no stream ordering), so OverlappedGather takes no side stream and synchronises the compute stream before
    each post.  Covers slices / bounds / both transfer paths / micro-batches on DEVICE tensors with real peers; the
    side-stream wiring itself is test_overlapped_gather_side_stream_wiring_loopback."""
    res = _run(tmp_path, world, total, backend="gloo", extra=["--device", "cuda"])
    assert all(all(v[0] for v in r["overlap"].values()) for r in res), res


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("chunks", [1, 2, 4])
def test_overlapped_gather_side_stream_wiring_loopback(world, chunks):
    """OverlappedGather.run through the REAL event -> side stream -> post -> finish() code on the device, one process
    playing every rank (vdr.dist.LoopbackTransport: the peers' rows arrive as device-to-device copies enqueued on the
    side stream, this rank's rows leave as a copy into `sent`).  Every pretended rank must end with the matrix of the
    whole-batch forward, bit for bit, and must have "sent" exactly the rows its forward wrote -- a post that ran ahead of
    its micro-batch's kernels would send the -77 fill.  Consumer contract: umap_cls_token.py:133-139."""
    import vdr
    from oracle import vit_oracle as vo
    from vdr.dist import LoopbackTransport, OverlappedGather
    total = 13  # ragged over 2 and 3 ranks
    cfg = vo.VitCfg(64, 16, 3, 128, 2, 2, 512)
    eng = vdr.Engine(vdr.VdrConfig(img=64, patch=16, dim=128, heads=2, layers=2, mlp_hidden=512))
    eng.load_weights(vo.make_weights(cfg, seed=1, scale=0.05))
    x = vo.make_images(cfg, total, seed=0).cuda()
    for out_mode in (vdr.OUT_CLS, vdr.OUT_DENSE):
        whole = eng.forward(x, out_mode)
        whole = whole.reshape(total, -1).contiguous()
        torch.cuda.synchronize()
        for rank in range(world):
            feats = torch.full_like(whole, -77.0)
            tr = LoopbackTransport(world, rank, whole)
            og = OverlappedGather(feats, total, chunks=chunks, mode="mesh", transport=tr)
            assert og.side is not None and og.chunks == min(chunks, max(b - a for a, b in og.bounds))
            lo, hi = og.my_rows()
            main = torch.cuda.current_stream().cuda_stream

            def fwd(x0, x1, rows):
                eng.forward_into(x[lo + x0: lo + x1].contiguous(), rows, out_mode)

            og.run(fwd)
            # no synchronisation here: finish() must have ordered the compute stream behind the side stream
            assert torch.equal(feats, whole), (world, rank, chunks, out_mode)
            assert torch.equal(tr.sent[lo:hi], whole[lo:hi]), "a micro-batch was sent before its forward had written it"
            assert len(tr.posts) == og.chunks and all(s != main for _, s in tr.posts), "posts must run on the side stream"
            assert og.last_gather_ms() is not None

