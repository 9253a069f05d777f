#!/usr/bin/env python3
"""Run torch.nn.functional.linear (hipBLASLt) on the four ViT-B GEMM shapes, for a `rocprofv3 --kernel-trace --stats` pass:
the vendor kernel names encode macro tile, MFMA shape and load path -- a reference point for csrc/gemm_kernels.h, nothing
the product calls."""
import torch

M = 50432
shapes = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    for _ in range(12):
        y = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    evs[0].record()
    for _ in range(20):
        y = torch.nn.functional.linear(x, w, b)
    evs[1].record()
    torch.cuda.synchronize()
    print(f"{name}: {evs[0].elapsed_time(evs[1]) / 20 * 1000:.1f} us", flush=True)
