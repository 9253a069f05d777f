#!/usr/bin/env python3
"""Does the gloo backend move CUDA tensors (point-to-point and all_gather_into_tensor)?  Two ranks on ONE GPU: if it does,
vdr.dist.OverlappedGather's stream / event logic can be exercised with real device tensors on a one-GPU box."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def work(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    res = {}
    try:
        x = torch.full((4, 3), float(rank), device=dev)
        y = torch.empty_like(x)
        ops = [dist.P2POp(dist.isend, x, 1 - rank), dist.P2POp(dist.irecv, y, 1 - rank)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        res["p2p"] = bool((y == float(1 - rank)).all())
    except Exception as e:  # noqa: BLE001
        res["p2p"] = f"{type(e).__name__}: {str(e)[:120]}"
    try:
        out = torch.empty(8, 3, device=dev)
        dist.all_gather_into_tensor(out, torch.full((4, 3), float(rank), device=dev))
        torch.cuda.synchronize()
        res["all_gather_into_tensor"] = bool((out[:4] == 0).all() and (out[4:] == 1).all())
    except Exception as e:  # noqa: BLE001
        res["all_gather_into_tensor"] = f"{type(e).__name__}: {str(e)[:120]}"
    print(rank, res, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(work, args=(2, port), nprocs=2, join=True)
