#!/usr/bin/env python3
"""Stream GEMM (variant 30) against ring4 (variant 26, packed weights): bitwise comparison + interleaved timing."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch  # noqa: E402

import vdr  # noqa: E402
from vdr import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--skip-timing", action="store_true")
    ap.add_argument("--variant", type=int, default=30, help="30 = stream kernel (plain weights); other numbers: packed weights")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    ok = True
    # exact integer cases incl. ragged M, several tiles per workgroup
    for (M, N, K) in [(128, 256, 768), (333, 768, 768), (1000, 512, 1024), (130, 256, 1536), (128 * 300 + 5, 256, 832), (70000, 768, 768)]:
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-3, 4, (N,), generator=g).float()
        Mr = (M + 127) // 128 * 128
        xpad = torch.full((Mr, K), float("nan"), dtype=torch.bfloat16, device=dev)  # rows past M: readable, never used
        xpad[:M] = x.bfloat16().to(dev)
        xd, Wd, bd = xpad[:M], W.bfloat16().to(dev), b.to(dev)
        ref = ops.linear(xd, Wd, bd, epilogue=vdr.EPI_BIAS, variant=22)
        pad = torch.full((M + 256, N), 7.0, device=dev, dtype=torch.bfloat16)
        out = pad[:M]
        if a.variant == 30:
            ops.linear(xd, Wd, bd, epilogue=vdr.EPI_BIAS, variant=30, out=out, x_rows=Mr)
        else:
            ops.linear(xd, Wd, bd, epilogue=vdr.EPI_BIAS, variant=a.variant, out=out)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        guard = bool((pad[M:] == 7.0).all())
        print(f"int  M{M} N{N} K{K}: equal={same} guard_untouched={guard}", flush=True)
        if not same:
            d = (out.float() - ref.float()).abs()
            idx = torch.nonzero(d > 0)
            print("   mismatches:", idx.shape[0], "first", idx[:5].tolist(), flush=True)
        ok &= same and guard
    M = 50432
    shapes = {"qkv": (2304, 768, vdr.EPI_BIAS), "fc1": (3072, 768, vdr.EPI_BIAS_GELU), "wide_k": (768, 3072, vdr.EPI_BIAS)}
    cases = []
    for name, (N, K, epi) in shapes.items():
        x = torch.randn(M, K, device=dev, generator=None).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        Wp = ops.pack_linear_weight(W)
        o26 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        o30 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.linear(x, Wp, b, epilogue=epi, variant=26, out=o26, packed=True)
        for rep in range(3):
            o30.zero_()
            if a.variant == 30:
                ops.linear(x, W, b, epilogue=epi, variant=30, out=o30)
            else:
                ops.linear(x, Wp, b, epilogue=epi, variant=a.variant, out=o30, packed=True)
            torch.cuda.synchronize()
            same = torch.equal(o26, o30)
            if not same:
                d = (o26.float() - o30.float()).abs()
                bad = torch.nonzero(d > 0)
                rows = torch.unique(bad[:, 0])
                print(f"rand {name}: MISMATCH rep {rep}: {bad.shape[0]} elements in {rows.shape[0]} rows; first {bad[:4].tolist()} max {d.max().item():.4g}", flush=True)
            ok &= same
        print(f"rand {name} M{M} N{N} K{K}: equal={same}", flush=True)
        cases.append((name, N, K, 26, lambda x=x, Wp=Wp, b=b, epi=epi, o=o26: ops.linear(x, Wp, b, epilogue=epi, variant=26, out=o, packed=True)))
        if a.variant == 30:
            cases.append((name, N, K, 30, lambda x=x, W=W, b=b, epi=epi, o=o30: ops.linear(x, W, b, epilogue=epi, variant=30, out=o)))
        else:
            cases.append((name, N, K, a.variant, lambda x=x, Wp=Wp, b=b, epi=epi, o=o30, v=a.variant: ops.linear(x, Wp, b, epilogue=epi, variant=v, out=o, packed=True)))
    print("ALL EQUAL" if ok else "FAILED", flush=True)
    if a.skip_timing:
        return 0 if ok else 1
    times = [[] for _ in cases]
    for rnd in range(a.rounds):
        order = list(range(len(cases)))
        if rnd & 1:
            order.reverse()
        evs = []
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
        med = ts[len(ts) // 2]
        print(f"time {name:8s} N{N} K{K} variant {v}: {med * 1e3:8.1f} us (min {ts[0] * 1e3:.1f})  {2.0 * M * N * K / (med * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
