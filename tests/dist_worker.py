"""Rank process for the multi-rank tests, started by vdr.dist.launch_ranks (the launcher bench.py uses).

    --backend gloo : CPU tensors; the per-rank "forward" is an identity on a known matrix (no GPU needed)
    --backend gloo --device cuda : every rank shares cuda:0 and runs the HIP forward; gloo moves the device rows through
                     the host with no stream ordering, so OverlappedGather synchronises before each post (no side stream)
    --backend nccl : one rank per GPU (RCCL); every rank runs the HIP forward on its batch shard, the shards are
                     all-gathered and compared BITWISE with that rank's own forward of the whole batch

Each rank writes {"rank", "world", "ok", ...} as JSON to <out>.<rank>; the test reads them back.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "vit-deep-radiomics_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--total", type=int, default=8)
    ap.add_argument("--dim", type=int, default=5)
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", default="cpu", choices=["cpu", "cuda"])
    a = ap.parse_args()
    from vdr.dist import all_gather_rows, extract_features_sharded, init_from_env, shard_bounds

    rank, world, local_rank = init_from_env(a.backend)
    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "group_size": dist.get_world_size()}
    try:
        lo, hi = shard_bounds(a.total, rank, world)
        if a.backend == "gloo" and a.device == "cpu":
            full = torch.arange(a.total * a.dim, dtype=torch.float32).reshape(a.total, a.dim) * 0.5 - 3.0
            calls = []

            def fake_extract(local):  # stands in for the per-rank HIP forward
                calls.append(local.shape[0])
                return local.clone()

            out = extract_features_sharded(fake_extract, full[lo:hi], a.total)
            res["ok"] = bool(torch.equal(out, full) and out.is_contiguous() and calls == [hi - lo])
            # the step that hides its gather (vdr.dist.OverlappedGather): same matrix through both transfer paths, whole
            # and in micro-batches (more micro-batches than some ranks have rows included), bf16 rows as config 4 gathers
            from vdr.dist import OverlappedGather
            res["overlap"] = {}
            for dt in (torch.float32, torch.bfloat16):
                want = full.to(dt)
                for mode in ("mesh", "collective", "auto"):
                    for chunks in (1, 3, 0):
                        feats = torch.full((a.total, a.dim), -77.0, dtype=dt)
                        og = OverlappedGather(feats, a.total, chunks=chunks, mode=mode)
                        seen = []

                        def fwd(x0, x1, rows, seen=seen, want=want):
                            seen.append((x0, x1))
                            rows.copy_(want[lo + x0: lo + x1])

                        got = og.run(fwd)
                        covered = sorted(seen) and sum(b - a_ for a_, b in seen) == hi - lo
                        good = bool(torch.equal(got, want) and got.data_ptr() == feats.data_ptr() and covered)
                        res["overlap"][f"{str(dt)[6:]}/{mode}/{chunks}"] = [good, og.mode, og.chunks]
                        res["ok"] = res["ok"] and good
            # ragged shards on both sides of the mesh threshold: the plan must come out the same on every rank (it is
            # made from the largest shard), or the ranks post different numbers of transfers and hang
            import vdr.dist as vd
            sizes = sorted({shard_bounds(a.total, r, world)[1] - shard_bounds(a.total, r, world)[0] for r in range(world)})
            if len(sizes) == 2:
                keep = vd.MESH_THRESHOLD_BYTES
                vd.MESH_THRESHOLD_BYTES = (sizes[0] * a.dim * 4 + sizes[1] * a.dim * 4) // 2
                try:
                    feats = torch.full((a.total, a.dim), -77.0)
                    og = OverlappedGather(feats, a.total, chunks=0, mode="auto")
                    got = og.run(lambda x0, x1, rows: rows.copy_(full[lo + x0: lo + x1]))
                    good = bool(torch.equal(got, full))
                    res["straddle"] = [good, og.mode, og.chunks]
                    res["ok"] = res["ok"] and good and og.mode == "mesh" and og.chunks == min(4, sizes[1])
                finally:
                    vd.MESH_THRESHOLD_BYTES = keep
        else:
            import vdr
            from oracle import vit_oracle as vo
            shared = a.backend == "gloo"  # every rank on cuda:0 (one-GPU box): gloo carries the device rows
            torch.cuda.set_device(0 if shared else local_rank)
            cfg = vo.VitCfg(64, 16, 3, 128, 2, 2, 512)
            eng = vdr.Engine(vdr.VdrConfig(img=64, patch=16, dim=128, heads=2, layers=2, mlp_hidden=512))
            eng.load_weights(vo.make_weights(cfg, seed=1, scale=0.05))
            x = vo.make_images(cfg, a.total, seed=0).cuda()
            whole = eng.forward(x, vdr.OUT_CLS)                       # the 1-rank result, on every rank
            mine = eng.forward(x[lo:hi].contiguous(), vdr.OUT_CLS)    # this rank's shard
            got = all_gather_rows(mine, a.total)
            # an explicit collective even at world == 1 (all_gather_rows short-cuts there): RCCL itself is exercised
            probe = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), device=mine.device) if (hi - lo) * world == a.total else None
            if probe is not None:
                dist.all_gather_into_tensor(probe, mine.contiguous())
            torch.cuda.synchronize()
            res["checks"] = {"gathered == whole": bool(torch.equal(got, whole)), "contiguous fp32": bool(got.is_contiguous() and got.dtype == torch.float32),
                             "explicit collective == whole": bool(probe is None or torch.equal(probe, whole))}
            res["ok"] = all(res["checks"].values())
            res["rows"] = [lo, hi]
            # the overlapped step of bench.py --gpus N: micro-batches gathered on a side stream, both transfer paths
            from vdr.dist import OverlappedGather
            res["overlap"] = {}
            for mode in ("mesh", "collective"):
                for chunks in (1, 2):
                    feats = torch.full_like(whole, -77.0)
                    og = OverlappedGather(feats, a.total, chunks=chunks, mode=mode)
                    assert (og.side is None) == (shared or world == 1), "side stream only behind a stream-ordered transport"
                    og.run(lambda x0, x1, rows: eng.forward_into(x[lo + x0: lo + x1].contiguous(), rows, vdr.OUT_CLS))
                    torch.cuda.synchronize()
                    good = bool(torch.equal(feats, whole))
                    res["overlap"][f"{mode}/{chunks}"] = [good, og.mode, og.chunks, og.last_gather_ms()]
                    res["ok"] = res["ok"] and good
    except Exception as e:  # a failed rank must not park its peers in a barrier: report, skip the barrier, leave
        res["ok"] = False
        res["error"] = f"{type(e).__name__}: {e}"[:400]
        with open(f"{a.out}.{rank}", "w") as f:
            json.dump(res, f)
        os._exit(1)
    with open(f"{a.out}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
