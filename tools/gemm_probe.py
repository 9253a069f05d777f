#!/usr/bin/env python3
"""Small driver for rocprofv3: a few GEMM launches per (shape, variant) so counters can be read per dispatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch
import vdr
from vdr import ops
M = 50432
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,5").split(",")]
shapes = {"fc1": (3072, 768, vdr.EPI_BIAS_GELU), "proj": (768, 768, vdr.EPI_BIAS_RESID), "fc2": (768, 3072, vdr.EPI_BIAS_RESID)}
torch.manual_seed(0)
for name, (N, K, epi) in shapes.items():
    x = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16() if epi == vdr.EPI_BIAS_RESID else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for v in variants:
        for _ in range(3):
            ops.linear(x, W, b, resid=r, epilogue=epi, variant=v, out=out)
    torch.cuda.synchronize()
print("probe done")
