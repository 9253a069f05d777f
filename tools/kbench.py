#!/usr/bin/env python3
"""Per-kernel timings at the BASELINE config-2 shapes (ViT-B/16, B=256: M = 50432).
Interleaved rounds in one process (guide §5.4 rule 24), random data (rule 25)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch  # noqa: E402

import vdr  # noqa: E402
from vdr import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=197)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--variants", type=str, default="0,2,3")
    ap.add_argument("--out", type=str, default="")
    ap.add_argument("--packed", type=int, default=1, help="1: weights in the packed (pair-interleaved) layout, as in the forward")
    ap.add_argument("--gemm-only", action="store_true")
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--lds-pad", type=str, default="", help="tuning builds: VDR_GEMM_LDS_PAD values to compare, e.g. 0,40000 (40000: one ring4 workgroup per CU)")
    a = ap.parse_args()
    M, D = a.batch * a.seq, a.dim
    H = D // 64
    res = {}
    dev = "cuda"
    torch.manual_seed(0)
    shapes = {"qkv": (3 * D, D, vdr.EPI_BIAS), "proj": (D, D, vdr.EPI_BIAS_RESID), "fc1": (4 * D, D, vdr.EPI_BIAS_GELU),
              "fc2": (D, 4 * D, vdr.EPI_BIAS_RESID)}
    # interleaved rounds in ONE process (guide rule 24): every round times each (shape, variant) once, so clock /
    # thermal drift hits all variants alike; report the median and the minimum over the rounds
    cases = []
    for name, (N, K, epi) in shapes.items():
        x = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).bfloat16() if epi == vdr.EPI_BIAS_RESID else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        Wp = ops.pack_linear_weight(W) if a.packed else W
        for v in [int(s) for s in a.variants.split(",")]:
            if a.lds_pad:
                for pad in a.lds_pad.split(","):
                    def run(x=x, Wp=Wp, b=b, r=r, epi=epi, v=v, out=out, pad=pad):
                        os.environ["VDR_GEMM_LDS_PAD"] = pad
                        ops.linear(x, Wp, b, resid=r, epilogue=epi, variant=v, out=out, packed=bool(a.packed))
                    cases.append((name + "/pad" + pad, N, K, v, run))
                continue
            cases.append((name, N, K, v, (lambda x=x, Wp=Wp, b=b, r=r, epi=epi, v=v, out=out:
                                          ops.linear(x, Wp, b, resid=r, epilogue=epi, variant=v, out=out, packed=bool(a.packed)))))
    for c in cases:
        c[4]()
    torch.cuda.synchronize()
    times = [[] for _ in cases]
    # The variants of a shape share their tensors: whichever runs second finds them in the Infinity Cache.  Every case
    # therefore runs once untimed right before its timed pair (same warm state for all), and the order alternates.
    for rnd in range(a.rounds):
        evs = []
        order = list(range(len(cases)))
        if rnd & 1:
            order.reverse()
        for i in order:
            c = cases[i]
            c[4]()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            c[4]()
            c[4]()
            e1.record()
            evs.append((i, e0, e1))
        torch.cuda.synchronize()
        for i, e0, e1 in evs:
            times[i].append(e0.elapsed_time(e1) / 2)
    for (name, N, K, v, _), ts in zip(cases, times):
        ts = sorted(ts)
        med, mn = ts[len(ts) // 2], ts[0]
        tf = 2.0 * M * N * K / (med * 1e-3) / 1e12
        res[f"gemm_{name}_v{v}"] = {"ms": med, "min_ms": mn, "TF": tf}
        print(f"gemm {name:14s} M{M} N{N} K{K} variant {v:4d}: {med:8.4f} ms (min {mn:.4f})  {tf:7.1f} TFLOP/s", flush=True)
    if a.gemm_only:
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
        return
    qkv = torch.randn(M, 3 * D, device=dev).bfloat16()
    for v in (3, 2, 1, 12, 22):
        med, mn = timeit(lambda: ops.attention(qkv, a.batch, a.seq, H, variant=v))
        tf = 4.0 * a.seq * a.seq * 64 * H * a.batch / (med * 1e-3) / 1e12
        res[f"attention_v{v}"] = {"ms": med, "min_ms": mn, "TF": tf}
        print(f"attention B{a.batch} N{a.seq} H{H} variant {v}: {med:8.3f} ms (min {mn:.3f})  {tf:7.1f} TFLOP/s", flush=True)
    x = torch.randn(M, D, device=dev).bfloat16()
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    med, mn = timeit(lambda: ops.layernorm(x, g, b, 1e-6))
    gbs = 4.0 * M * D / (med * 1e-3) / 1e9
    res["layernorm"] = {"ms": med, "min_ms": mn, "GBs": gbs}
    print(f"layernorm M{M} D{D}: {med:8.3f} ms  {gbs:8.1f} GB/s", flush=True)
    img = torch.rand(a.batch, 3, 224, 224, device=dev)
    Wp = (torch.randn(D, 3, 16, 16, device=dev) * 0.05).bfloat16()
    bp = torch.randn(D, device=dev)
    for dt in (torch.float32, torch.bfloat16):
        xi = img.to(dt)
        med, mn = timeit(lambda: ops.patch_embed(xi, Wp, bp, 16))
        by = a.batch * 3 * 224 * 224 * xi.element_size() + a.batch * 196 * D * 2 + D * 768 * 2
        res[f"patch_embed_{str(dt)[6:]}"] = {"ms": med, "GBs": by / (med * 1e-3) / 1e9}
        print(f"patch_embed in={dt}: {med:8.3f} ms  algorithmic {by / (med * 1e-3) / 1e9:8.1f} GB/s", flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
