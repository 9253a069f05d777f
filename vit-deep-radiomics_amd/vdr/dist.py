"""Batch-shard data parallelism over the 8 GPUs of one node.

One process per GPU (torchrun); every image is independent end to end, so the only exchange is ONE
all-gather of the feature matrix into its final row order (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Pure gather, no reduction: the N-rank result is bitwise equal to
the 1-rank result.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(script: str, nproc: int, args=(), env=None, timeout=None) -> int:
    """Start `nproc` fresh rank processes of `script` on this node (one per GPU) and wait for them:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script args`.
    The ranks are CHILD processes created with subprocess (fork + exec of a new interpreter); the caller must not
    have touched the GPU yet and is never replaced itself -- a process that has initialised HIP is not re-exec'd.
    stdout / stderr of the ranks pass through.  Returns the launcher's exit code (0 = every rank exited 0)."""
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)  # a stale rendezvous in the caller's environment must not leak into the children
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *[str(a) for a in args]]
    return subprocess.run(cmd, env=e, timeout=timeout).returncode


def init_from_env(backend: str = "nccl", device=None):
    """Join the process group torch.distributed.run (or launch_ranks) prepared: RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_* from the environment.  backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors.
    Returns (rank, world, local_rank)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=device if device is not None else torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


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
