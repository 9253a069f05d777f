#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch
from vdr import ops
B, N, H = 256, 197, 12
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
for v in (0, 1):
    for _ in range(3):
        ops.attention(qkv, B, N, H, variant=v)
torch.cuda.synchronize()
print("done")
