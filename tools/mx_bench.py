#!/usr/bin/env python3
"""bf16 vs MX-fp8 GEMM at the ViT-B / ViT-g shapes (one GPU).  python tools/mx_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-deep-radiomics_amd"))
import torch
from vdr import ops, _lib as L


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


shapes = [("vitb qkv", 50432, 2304, 768), ("vitb fc1", 50432, 3072, 768), ("vitb fc2", 50432, 768, 3072),
          ("vitg qkv", 8224, 4608, 1536), ("vitg w12", 8224, 8192, 1536), ("vitg w3", 8224, 1536, 4096),
          ("vitg proj", 8224, 1536, 1536)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    xq, wq = ops.mx_quantize(x), ops.mx_quantize(w)
    fl = 2.0 * M * N * K
    res = []
    for v in (19, 21):
        t = timeit(lambda: ops.linear(x, w, bias=b, variant=v))
        res.append(f"bf16 v{v} {t:.3f} ms {fl / t / 1e9:.0f} TF")
    for v in (0, 1, 2):
        t = timeit(lambda: ops.linear_mx(xq, wq, bias=b, variant=v))
        res.append(f"mx v{v} {t:.3f} ms {fl / t / 1e9:.0f} TF")
    if "fc1" in name or "w12" in name:
        epi = L.EPI_SWIGLU if "w12" in name else L.EPI_BIAS_GELU
        for v in (0, 1):
            t = timeit(lambda: ops.linear_mx(xq, wq, bias=b, epilogue=epi, variant=v))
            res.append(f"mx v{v} act->bf16 {t:.3f}")
            t = timeit(lambda: ops.linear_mx(xq, wq, bias=b, epilogue=epi, variant=v, mx_out=True))
            res.append(f"mx v{v} act->mx {t:.3f}")
        t = timeit(lambda: ops.linear(x, w, bias=b, epilogue=epi, variant=26))
        res.append(f"bf16 v19 act {t:.3f}")
    t = timeit(lambda: ops.mx_quantize(x))
    res.append(f"quant {t:.3f} ms")
    print(f"{name:10s} M={M} N={N} K={K}: " + " | ".join(res), flush=True)
