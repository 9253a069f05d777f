#!/usr/bin/env python3
"""Headline benchmark: images/s of the ViT-B/16 224^2 bf16 CLS-feature forward path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 (no RANK / WORLD_SIZE in the environment) it launches the N ranks itself
(vdr.dist.launch_ranks: fresh child processes of torch.distributed.run, created BEFORE this process makes any GPU
call; this process only waits for them and returns their exit code).

One "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
[256,3,224,224] bf16 per GPU -> vdr_forward (patch-embed, 12 pre-LN blocks, final LN + CLS slice)
-> [256,768] fp32 CLS features; for N > 1 each rank processes its own 256 images (weak scaling,
BASELINE config 3 = 8 x 256) and ONE all-gather (RCCL over xGMI) reassembles the [N*256,768]
feature matrix inside the step.  Rank 0 prints ONE JSON line.

Last block: with a CLS output the library runs the last block's out-projection / norm2 / MLP on the CLS rows only (after
the last attention every operation is row-wise and x[:, 0] is all `model(x) -> (logits, cls)` returns; the features are
the bits of the full computation, checked in every run: full_last_block.features_bitwise_equal).  `value` is that
default; `full_last_block.value` (same run, same K steps) computes every token like the reference; the FLOP counts of
both are on the line and whole_forward_frac prices the FLOPs actually executed.  --full-last-block times the latter.

roofline: the dominant kernel class (by summed time) of the timed region, timed with HIP events
recorded by libvdr on the stream the kernels are launched on (vdr_profile_*): achieved = algorithmic
FLOPs of those launches / their summed duration; peak = 2.5 PFLOP/s dense bf16 MFMA.
cpu_baseline: the CPU fp32 oracle (oracle/vit_oracle.py, a port — the reference's Python never
travels) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vit-deep-radiomics_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0   # dense MX-fp8, same table
PEAK_HBM_GBS = 8000.0

GEMM_CLASSES = ("gemm_patch", "gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2", "attention")


def usable_cores():
    """Host cores this process may actually use: cgroup CPU quota if one is set, else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_baseline(cfg_name, seconds_target=12.0, max_threads=64):
    """Oracle forward on the host cores; bounded sample (batch 8 per run, ~10-25 s of CPU work)."""
    from oracle import vit_oracle as vo
    cfg = vo.CONFIGS[cfg_name]
    cores = min(usable_cores(), max_threads)
    torch.set_num_threads(cores)
    w = vo.make_weights(cfg, seed=1)
    b = 8
    x = vo.make_images(cfg, b, seed=0)
    t0 = time.perf_counter()
    vo.forward_images(cfg, w, x)  # warm-up
    warm = time.perf_counter() - t0
    runs, t0 = 0, time.perf_counter()
    while True:
        vo.forward_images(cfg, w, x)
        runs += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or runs >= 50 or dt + warm > 2.5 * seconds_target:
            break
    return {"value": round(runs * b / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{runs} runs x batch {b} of {cfg_name} fp32 (oracle/vit_oracle.forward_images), {dt:.1f} s wall, "
                      f"{os.cpu_count()} logical CPUs visible"}


def cpu_baseline_sam(seconds_target=12.0, max_threads=64):
    """The reference's default backbone on the host cores: one 1024^2 slice at a time through the SAM ViT-B image encoder
    restated in oracle/sam_oracle.py (12 blocks, window 14 + 4 global blocks, conv neck), fp32 -- what
    tfds_dense_descriptor.py:271-281 does per slice on a machine without a GPU.  Bounded sample."""
    from oracle import sam_oracle as so
    cfg = so.SAM_VIT_B
    cores = min(usable_cores(), max_threads)
    torch.set_num_threads(cores)
    w = so.make_weights(cfg, seed=1)
    x = so.make_images(cfg, 1, seed=0)
    t0 = time.perf_counter()
    so.sam_forward(cfg, w, x)  # warm-up
    warm = time.perf_counter() - t0
    runs, t0 = 0, time.perf_counter()
    while True:
        so.sam_forward(cfg, w, x)
        runs += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or runs >= 50 or dt + warm > 2.5 * seconds_target:
            break
    return {"value": round(runs / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{runs} slices of 1024^2 through oracle/sam_oracle.sam_forward (SAM ViT-B encoder, fp32), one per call, {dt:.1f} s wall, "
                      f"{os.cpu_count()} logical CPUs visible"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--model", type=str, default="vit_base16_224")
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--fp8", type=int, nargs="?", const=1, default=0, choices=[0, 1],
                    help="1: qkv / fc1 / fc2 as MX-fp8 on the block-scaled MFMA (BASELINE config 5; not the headline dtype)")
    ap.add_argument("--fp8-cls-bf16", action="store_true",
                    help="with --fp8: the MLP of the CLS rows on the bf16 weights (vdr_config.fp8_cls_bf16)")
    ap.add_argument("--resid-fp32", action="store_true",
                    help="vdr_config.resid_fp32 = 1: fp32 master copy of the residual stream (bf16 path; parity option, costs traffic)")
    ap.add_argument("--ln-fin-fused", action="store_true",
                    help="vdr_config.ln_fin_fused = 1: the LayerNorm fold's row statistics finalised inside the residual GEMMs instead of by their own launches (A/B: same bits, same time)")
    ap.add_argument("--out", choices=["cls", "dense"], default="cls",
                    help="cls: [N, D] CLS features (headline); dense: [N, n*D] per-patch descriptors (BASELINE config 4)")
    ap.add_argument("--input", choices=["bf16", "fp32"], default="bf16",
                    help="dtype of the resident input images (the reference feeds fp32; SURVEY §8d asks for both)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-last-block", action="store_true",
                    help="CLS output: run the out-projection / MLP of the LAST block on every token, as the reference does "
                         "before it keeps x[:, 0] (default: on the CLS rows only -- the same features bit for bit, "
                         "tests/test_model_gpu.py::test_cls_rows_only_last_block_bitwise).  Without this flag the default "
                         "run ALSO times this form after the timed region and reports it as `full_last_block`")
    ap.add_argument("--two-stream", action="store_true",
                    help="after the timed region, also measure the same step with micro-batches of B/2 on two internal HIP "
                         "streams (reported as `two_stream`; never the headline value; off by default so that a rocprof "
                         "summary of the default command holds full-batch launches only)")
    ap.add_argument("--gather-chunks", type=int, default=0,
                    help="N > 1: micro-batches per step whose gathers overlap the next micro-batch's kernels (0 = by message size: "
                         "1 for the CLS matrix, 4 for dense descriptors)")
    ap.add_argument("--gather-mode", choices=["auto", "mesh", "collective"], default="auto",
                    help="N > 1: mesh = point-to-point sends into the peers' row slices (batch_isend_irecv over the xGMI mesh), "
                         "collective = all_gather_into_tensor; auto = mesh above 8 MB per rank")
    ap.add_argument("--dense-dtype", choices=["bf16", "fp32"], default="bf16",
                    help="--out dense: dtype of the gathered per-patch descriptors (BASELINE config 4 states bf16)")
    ap.add_argument("--clean-timing", action="store_true",
                    help="keep the per-kernel HIP events out of the timed region (roofline from a second pass)")
    a = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # torchrun / torch.distributed.run
    if a.gpus > 1 and not launched:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has touched the GPU (importing torch
        # and vdr does not initialise HIP), the ranks are new child processes, and this process is not replaced.
        from vdr.dist import launch_ranks
        raise SystemExit(launch_ranks(os.path.abspath(__file__), a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libvdr has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                         f"(one rank per GPU; --gpus {a.gpus} needs {a.gpus} devices)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if launched:
        from vdr.dist import init_from_env
        init_from_env("nccl", dev)  # nccl == RCCL on ROCm

    # a checkout without the built library (artefacts are git-ignored): build it once, rank 0 first, exactly as
    # __graft_entry__.build() does; the product itself never builds or falls back (vdr.load() raises without the .so)
    lib_path = os.path.join(ROOT, "vit-deep-radiomics_amd", "vdr", "libvdr.so")
    if not os.path.exists(lib_path):
        if rank == 0:
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "vit-deep-radiomics_amd", "csrc"), "-j8"], check=True,
                           stdout=subprocess.DEVNULL)
        if launched:
            dist.barrier()
    import vdr
    from vdr.dist import all_gather_rows
    from oracle import vit_oracle as vo  # weights/images generators + cpu_baseline only

    sam = a.model == "medsam"
    if sam:
        from oracle import sam_oracle as so
        ocfg = so.SAM_VIT_B
        weights = so.make_weights(ocfg, seed=1)
    else:
        ocfg = vo.CONFIGS[a.model]
        weights = vo.make_weights(ocfg, seed=1)
    model = vdr.load_model(a.model, weights=weights, device=dev, micro_batch=a.micro_batch, streams=a.streams, fp8=a.fp8,
                           full_last_block=a.full_last_block, fp8_cls_bf16=a.fp8_cls_bf16, resid_fp32=a.resid_fp32,
                           ln_fin_fused=a.ln_fin_fused)
    eng = model.engine
    B, D = a.batch, ocfg.dim
    g = torch.Generator().manual_seed(1000 + rank)
    images = torch.rand(B, 3, ocfg.img, ocfg.img, generator=g)  # synthetic [0,1)
    images = (images if (sam or a.input == "fp32") else images.to(torch.bfloat16)).to(dev)  # the reference feeds fp32
    total = B * world
    dense = a.out == "dense" and not sam
    if sam:  # dense descriptor maps [N, 64, 64, 256] fp32, flattened to rows for the gather
        D = ocfg.grid * ocfg.grid * ocfg.out_chans
    elif dense:  # per-patch token descriptors [N, n, D] fp32, one row per image for the gather
        D = (ocfg.img // ocfg.patch) ** 2 * ocfg.dim
    # final row-ordered [N, D] matrix: fp32 CLS features (what umap_cls_token.py:139 / embedding_classifier.py:102 read);
    # dense per-patch descriptors in bf16 (BASELINE config 4) unless --dense-dtype fp32
    fdt = torch.bfloat16 if (dense and a.dense_dtype == "bf16") else torch.float32
    feats = torch.empty((total, D), dtype=fdt, device=dev)
    out_mode = vdr.OUT_ENCODER if sam else (vdr.OUT_DENSE if dense else vdr.OUT_CLS)
    # Every rank's forward writes STRAIGHT into its row slice of the matrix.  N > 1: micro-batch by micro-batch, each
    # finished micro-batch going to the peers on a side stream while the next one computes (vdr.dist.OverlappedGather)
    og = None
    if world > 1:
        from vdr.dist import OverlappedGather
        og = OverlappedGather(feats, total, chunks=a.gather_chunks, mode=a.gather_mode)
    mine = feats if world == 1 else feats[rank * B:(rank + 1) * B]

    def step():
        if og is None:
            eng.forward_into(images, mine, out_mode)
        else:
            og.run(lambda x0, x1, rows: eng.forward_into(images[x0:x1], rows, out_mode))

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    # one fully bracketed (untimed) step: which class dominates
    eng.profile(True)
    eng.profile_read()
    step()
    sync()
    prof_one = eng.profile_read()
    dom = max((k for k in prof_one if k in GEMM_CLASSES), key=lambda k: prof_one[k]["ms"])
    # timed region: ONLY the dominant kernel class is bracketed with HIP events (its launches are
    # timed live on the stream they run on; 2 events per launch keep the cost negligible)
    eng.profile(not a.clean_timing, classes=[dom])
    eng.profile_read()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof_dom = eng.profile_read()
    eng.profile(False)
    # second, untimed pass over the same K steps with EVERY class bracketed: the per-class table is an average over K
    # steps, like the dominant class (one step alone reads im2col at 0.063 ms where rocprof's average says 0.036)
    eng.profile(True)
    eng.profile_read()
    for _ in range(a.steps):
        step()
    sync()
    prof_all = eng.profile_read()
    eng.profile(False)
    if a.clean_timing:
        prof_dom = {dom: prof_all[dom]}
    for v in prof_all.values():  # per step
        v["ms"] /= a.steps
        v["flops"] /= a.steps
        v["bytes"] /= a.steps
        v["launches"] = v["launches"] / a.steps

    # informational: the same K steps with the batch split over two internal HIP streams (vdr_config.streams = 2,
    # micro_batch = B/2).  Kernels of the two halves overlap (one GEMM's store-heavy epilogue and ramp-down under the
    # other's main loop), so the step gets shorter while every single launch gets LONGER -- which is why it is not the
    # default configuration of the measurement: the per-kernel roofline above is taken with one kernel on the chip.
    two = None
    if a.two_stream and not sam and world == 1 and B >= 2 and not a.streams and not a.micro_batch:
        try:
            m2 = vdr.load_model(a.model, weights=weights, device=dev, micro_batch=(B + 1) // 2, streams=2, fp8=a.fp8)
            mode = vdr.OUT_DENSE if dense else vdr.OUT_CLS
            for _ in range(max(a.warmup, 2)):
                m2.engine.forward_into(images, mine, mode)
            sync()
            t2 = time.perf_counter()
            for _ in range(a.steps):
                m2.engine.forward_into(images, mine, mode)
            sync()
            dt2 = time.perf_counter() - t2
            two = {"value": round(B * a.steps / dt2, 1), "ms_per_step": round(dt2 / a.steps * 1e3, 3), "micro_batch": (B + 1) // 2,
                   "streams": 2}
            del m2
        except Exception as e:  # never let the side measurement break the benchmark line
            two = {"error": str(e)[:200]}

    # informational, same K steps: every token of the last block computed, as the reference does before it keeps
    # x[:, 0].  The default (timed above) runs the last block's out-projection / MLP on the CLS rows only -- the same
    # features bit for bit -- and that is an algorithmic saving of this path, so both numbers are on the line.
    full = None
    if not a.full_last_block and not sam and not dense and world == 1:
        try:
            mf = vdr.load_model(a.model, weights=weights, device=dev, micro_batch=a.micro_batch, streams=a.streams, fp8=a.fp8,
                                full_last_block=True, fp8_cls_bf16=a.fp8_cls_bf16, resid_fp32=a.resid_fp32,
                                ln_fin_fused=a.ln_fin_fused)
            ref = torch.empty_like(mine)
            for _ in range(max(a.warmup, 2)):
                mf.engine.forward_into(images, ref, vdr.OUT_CLS)
            sync()
            tf0 = time.perf_counter()
            for _ in range(a.steps):
                mf.engine.forward_into(images, ref, vdr.OUT_CLS)
            sync()
            dtf = time.perf_counter() - tf0
            full = {"value": round(B * a.steps / dtf, 1), "ms_per_step": round(dtf / a.steps * 1e3, 3),
                    "features_bitwise_equal": bool(torch.equal(ref, mine))}
            del mf, ref
        except Exception as e:  # never let the side measurement break the benchmark line
            full = {"error": str(e)[:200]}

    # N > 1, informational: what the transfers of a step occupy on the side stream, and the same K steps WITHOUT any
    # transfer (the N = 1-equivalent rate of this rank: what the overlap is measured against)
    gather = None
    if og is not None:
        step()
        sync()
        gms = og.last_gather_ms()
        sync()
        tc0 = time.perf_counter()
        for _ in range(a.steps):
            eng.forward_into(images, mine, out_mode)
        sync()
        dtc = time.perf_counter() - tc0
        gather = {"mode": og.mode, "chunks": og.chunks, "bytes_per_rank": int(B * D * feats.element_size()),
                  "side_stream_ms_per_step": None if gms is None else round(gms, 3),
                  "compute_only_ms_per_step": round(dtc / a.steps * 1e3, 3),
                  "n1_equivalent_images_per_s_per_gpu": round(B * a.steps / dtc, 1)}
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    assert torch.isfinite(feats.float()).all()

    if rank == 0:
        flops_ref = so.flops_per_image(ocfg) if sam else vo.flops_per_image(ocfg)  # every token of every block
        # what this forward executes (sum of the launches' algorithmic FLOPs, booked by the library per launch): equal
        # to flops_ref except for the CLS-rows-only tail of the last block
        # (the SAM encoder keeps its algorithmic count: the launches also cover the zero-padded rows of the border windows)
        flops_img = flops_ref if sam else sum(v["flops"] for v in prof_all.values()) / B
        if not (0.85 * flops_ref <= flops_img <= 1.001 * flops_ref):
            raise SystemExit(f"executed FLOPs per image {flops_img:.4g} vs the model's {flops_ref:.4g}: accounting is off")
        ms = dt / a.steps * 1e3
        ips = total * a.steps / dt
        kern = {}
        for k, v in prof_all.items():
            e = {"ms_per_step": round(v["ms"], 4), "launches_per_step": v["launches"]}
            if v["flops"] > 0:
                e["TFLOP/s"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
            e["GB/s_algorithmic"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
            kern[k] = e
        dv = prof_dom[dom]
        ach = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch of that kernel from the committed rocprofv3 --pmc passes
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; profiles/r04_pmc_traffic.*): PMC counters cannot be
        # collected from inside this process, so the number is the profiled one for the same launch shape -- and only
        # while the kernels are the profiled ones: the file carries vdr.source_id() (sha256 over csrc/) of its run
        traffic, traffic_note = None, "profiles/r04_pmc_traffic.txt"
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
            key = {"gemm_fc1": "gemm_fc1 (EPI_BIAS_GELU, 8-phase)", "gemm_qkv": "gemm_qkv (EPI_BIAS, 8-phase)", "attention": "attention",
                   "gemm_fc2": "gemm_fc2 (EPI_BIAS_RESID, K > N)", "gemm_proj": "gemm_proj (EPI_BIAS_RESID, K = N)"}.get(dom)
            if pm.get("_source_id") != vdr.source_id():
                traffic_note = f"dropped: profiles/r04_pmc_traffic.json was taken with kernel sources {pm.get('_source_id')}, this run has {vdr.source_id()}"
            elif key in pm and a.model == "vit_base16_224" and B == 256:
                traffic = round((pm[key]["read_mb_corrected"] + pm[key]["write_mb"]) * 1e6)
        except Exception as e:
            traffic, traffic_note = None, f"unavailable: {type(e).__name__}"
        # fp8 run: the qkv / fc1 / fc2 GEMMs are priced against the dense fp8 peak, everything else against bf16
        peak = PEAK_FP8_TFLOPS if a.fp8 and dom in ("gemm_qkv", "gemm_fc1", "gemm_fc2") else PEAK_BF16_TFLOPS
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC passes: " + traffic_note + ")",
                "avg_launch_ms": round(dv["ms"] / dv["launches"], 4),
                "flops_per_launch": dv["flops"] / dv["launches"],
                "whole_forward_frac": round(flops_img * B * a.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
                "flops_per_image_executed": flops_img, "flops_per_image_every_token": flops_ref,
                "kernel_time_sum_ms_per_step": round(sum(v["ms"] for v in prof_all.values()), 3),
                "kernels_table": f"average over {a.steps} steps (second, fully bracketed pass)"}
        # north_star's own target is stated on the attention block (qkv projection + scaled-dot-product + out-projection):
        # algorithmic FLOPs of those three classes over their summed launch time, against the same dense bf16 peak
        blk = [prof_all[k] for k in ("gemm_qkv", "attention", "gemm_proj") if k in prof_all]
        if len(blk) == 3 and not a.fp8:
            bms = sum(v["ms"] for v in blk)
            roof["attention_block"] = {"ms_per_step": round(bms, 3), "TFLOP/s": round(sum(v["flops"] for v in blk) / (bms * 1e-3) / 1e12, 1),
                                       "frac": round(sum(v["flops"] for v in blk) / (bms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                       "target_frac": 0.60}
        out = {"metric": "images/sec, ViT-B/16 224^2 bf16 CLS-feature extraction" if a.model == "vit_base16_224"
               and not a.fp8 and not dense else ("slices/sec, MedSAM ViT-B 1024^2 dense descriptor (64,64,256)" + (" fp8 weights" if a.fp8 else "") if sam
                     else f"images/sec, {a.model}{' fp8 weights' if a.fp8 else ''} {'dense-descriptor' if dense else 'CLS-feature'} extraction"),
               "value": round(ips, 1), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "rccl_ranks": dist.get_world_size() if launched else 1,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "fp8 (MX e4m3 qkv/fc1/fc2, bf16 elsewhere)" if a.fp8 else "bf16", "data": "synthetic",
               "config": {"workload": (f"medsam (SAM ViT-B image encoder) {ocfg.img}^2 fp32 in / bf16 compute, batch {B}/GPU -> "
                                       f"[{total},64,64,256] fp32" if sam else
                                       f"{a.model} {ocfg.img}^2 {'MX-fp8 weights' if a.fp8 else 'bf16'}, batch {B}/GPU, {'dense per-patch descriptors' if dense else 'CLS-token extraction'} -> [{total},{D}] {'bf16' if fdt == torch.bfloat16 else 'fp32'}")
                                      + (", all-gather of feature matrix" if world > 1 else ""),
                          "global_batch": total, "parallelism": f"batch-shard dp{world}",
                          "weights": "random-init (seed 1)", "input_dtype": "fp32" if (sam or a.input == "fp32") else "bf16", "micro_batch": a.micro_batch, "streams": a.streams,
                          "fp8_cls_bf16": bool(a.fp8_cls_bf16), "resid_fp32": bool(a.resid_fp32), "ln_fin_fused": bool(a.ln_fin_fused),
                          "last_block": "every token" if (a.full_last_block or sam or dense) else
                                        "attention on every token; out-projection / norm2 / MLP on the CLS rows only (bitwise the same features)"},
               "feature_GBps": round(total * D * feats.element_size() * a.steps / dt / 1e9, 4),
               "gather": gather,
               "TFLOPs_per_s": round(flops_img * total * a.steps / dt / 1e12, 1),
               "roofline": roof, "kernels": kern, "two_stream": two, "full_last_block": full}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_sam() if sam else cpu_baseline(a.model)
        print(json.dumps(out), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
