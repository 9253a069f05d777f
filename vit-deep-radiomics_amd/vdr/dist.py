"""Batch-shard data parallelism over the 8 GPUs of one node.

One process per GPU (torchrun); every image is independent end to end, so the only exchange is ONE
all-gather of the feature matrix into its final row order (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Pure gather, no reduction: the N-rank result is bitwise equal to
the 1-rank result.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; the first total % world ranks get one extra row."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(local: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
    """Gather per-rank row blocks [rows_r, ...] into [total_rows, ...] in rank (== dataset) order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    bounds = [shard_bounds(total_rows, r, world) for r in range(world)]
    sizes = [hi - lo for lo, hi in bounds]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    out = torch.empty((total_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)  # one in-place collective
    else:
        # ragged tail: pad every shard to the largest, one collective, then drop the padding rows
        mx = max(sizes)
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: sizes[rank]] = local
        tmp = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(tmp, pad, group=group)
        for r, (lo, hi) in enumerate(bounds):
            out[lo:hi] = tmp[r * mx: r * mx + (hi - lo)]
    return out


def extract_features_sharded(extract_fn, images_local: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
    """extract_fn(images_local) -> [rows_local, D] on this rank's device; returns the full [N, D]
    matrix on every rank (the array umap_cls_token.py:139 stacks / embedding_classifier.py:102 reads)."""
    return all_gather_rows(extract_fn(images_local), total_rows, group)
