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


def _run(tmp_path, world, total):
    from vdr.dist import launch_ranks
    out = str(tmp_path / "res")
    rc = launch_ranks(WORKER, world, ["--backend", "nccl", "--total", total, "--out", out], timeout=600)
    assert rc == 0
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    assert all(r["ok"] and r["world"] == world and r["group_size"] == world and r["backend"] == "nccl" for r in res), res
    return res


def test_rccl_single_rank_process_group(tmp_path):
    """One rank under the launcher: RCCL initialises, all_gather_into_tensor runs, rows equal the plain forward."""
    _run(tmp_path, 1, 8)


@pytest.mark.parametrize("total", [8, 7])  # equal shards (one in-place collective) and a ragged tail
def test_rccl_two_rank_allgather_bitwise_equal_to_one_rank(tmp_path, total):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one rank per GPU); the same code runs under gloo in tests/test_host_logic.py")
    _run(tmp_path, 2, total)
