#!/usr/bin/env python3
"""What the reference's own style of execution does on this GPU: a plain PyTorch-ROCm eager ViT-B/16 forward
(stock nn.Conv2d / nn.LayerNorm / nn.MultiheadAttention-style SDPA / nn.Linear / F.gelu, i.e. rocBLAS/hipBLASLt +
torch's attention kernels), fp32 as the reference runs it (tfds_dense_descriptor.py:123, no autocast) and bf16.
Same shapes as bench.py's headline line (batch 256, 224^2, CLS out).  Not part of the product or of the tests;
it exists so README can quote a same-box number next to ours.   python tools/torch_baseline.py
"""
import json
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F


class Block(nn.Module):
    def __init__(self, d, h, f):
        super().__init__()
        self.n1, self.n2 = nn.LayerNorm(d, eps=1e-6), nn.LayerNorm(d, eps=1e-6)
        self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.fc1, self.fc2 = nn.Linear(d, f), nn.Linear(f, d)
        self.h = h

    def forward(self, x):
        B, N, D = x.shape
        q, k, v = self.qkv(self.n1(x)).reshape(B, N, 3, self.h, D // self.h).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)
        x = x + self.proj(a)
        return x + self.fc2(F.gelu(self.fc1(self.n2(x))))


class ViT(nn.Module):
    def __init__(self, img=224, p=16, d=768, h=12, layers=12, f=3072):
        super().__init__()
        self.pe = nn.Conv2d(3, d, p, p)
        self.cls = nn.Parameter(torch.zeros(1, 1, d))
        self.pos = nn.Parameter(torch.zeros(1, (img // p) ** 2 + 1, d))
        self.blocks = nn.ModuleList([Block(d, h, f) for _ in range(layers)])
        self.norm = nn.LayerNorm(d, eps=1e-6)

    def forward(self, x):
        x = self.pe(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], 1) + self.pos
        for b in self.blocks:
            x = b(x)
        return self.norm(x)[:, 0]


@torch.no_grad()
def run(dtype, batch=256, steps=10, warmup=3):
    torch.manual_seed(0)
    m = ViT().cuda().to(dtype).eval()
    x = torch.rand(batch, 3, 224, 224, device="cuda", dtype=dtype)
    for _ in range(warmup):
        m(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        y = m(x)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    return {"dtype": str(dtype).replace("torch.", ""), "ms_per_step": round(ms, 3), "images_per_s": round(batch / ms * 1e3, 1),
            "out": list(y.shape)}


@torch.no_grad()
def run_sam(dtype, batch=1, steps=10, warmup=3):
    """The reference's default backbone is the SAM ViT-B image encoder run eagerly in fp32, one slice per call
    (tfds_dense_descriptor.py:123).  segment_anything is not installed here; transformers' SamVisionModel is the
    same architecture (random init, no download)."""
    from transformers import SamVisionConfig, SamVisionModel
    torch.manual_seed(0)
    m = SamVisionModel(SamVisionConfig()).cuda().to(dtype).eval()
    x = torch.rand(batch, 3, 1024, 1024, device="cuda", dtype=dtype)
    for _ in range(warmup):
        m(pixel_values=x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        y = m(pixel_values=x).last_hidden_state
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    return {"dtype": str(dtype).replace("torch.", ""), "batch": batch, "ms_per_step": round(ms, 3),
            "slices_per_s": round(batch / ms * 1e3, 1), "out": list(y.shape)}


if __name__ == "__main__":
    if not torch.cuda.is_available():
        sys.exit("needs a GPU")
    for dt in (torch.float32, torch.bfloat16):
        print(json.dumps({"what": "PyTorch-ROCm eager ViT-B/16 224^2 batch 256 CLS", **run(dt)}), flush=True)
    for dt, b in ((torch.float32, 1), (torch.bfloat16, 1), (torch.bfloat16, 4)):
        try:
            print(json.dumps({"what": "PyTorch-ROCm eager SAM ViT-B 1024^2 image encoder (transformers SamVisionModel)",
                              **run_sam(dt, b)}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"what": "SAM baseline failed", "error": repr(e)[:200]}), flush=True)
