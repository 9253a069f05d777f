#!/usr/bin/env python3
"""Patch-embed as the reference runs it: `model.patch_embed(x)` of DINOv2 ViT-S/14 at 896^2, fp32 in -> fp32
[B, 4096, 384] out (/root/reference/src/tfds_dense_descriptor.py:128-133; the reference uses B = 1), plus the headline
geometry (ViT-B/16 224^2, B = 256, bf16 in -> bf16 tokens).  Reports achieved ALGORITHMIC GB/s (input image + weight +
output, each moved once) against the 8 TB/s HBM peak: north_star asks for "achieved HBM GB/s on patch-embed"."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch  # noqa: E402

import vdr  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def main():
    out = []
    g = torch.Generator().manual_seed(0)
    # the reference's dinov2 mode
    D, p, img = 384, 14, 896
    sd = {"patch_embed.proj.weight": torch.randn(D, 3, p, p, generator=g) * 0.05, "patch_embed.proj.bias": torch.randn(D, generator=g) * 0.1}
    model = vdr.load_model("dinov2", weights=sd)
    for B in (1, 2, 4, 8, 16):
        x = torch.rand(B, 3, img, img, generator=g).cuda()
        y = torch.empty(B, 4096, D, dtype=torch.float32, device="cuda")
        ms = timeit(lambda: model.engine.forward_into(x, y, vdr.OUT_PATCH_EMBED))
        by = B * 3 * img * img * 4 + D * 3 * p * p * 2 + B * 4096 * D * 4
        model.engine.profile(True)
        model.engine.profile_read()
        for _ in range(10):
            model.engine.forward_into(x, y, vdr.OUT_PATCH_EMBED)
        torch.cuda.synchronize()
        pr = model.engine.profile_read()
        model.engine.profile(False)
        rec = {"mode": "dinov2 ViT-S/14 896^2 fp32 -> fp32 (reference mode)", "batch": B, "ms": round(ms, 4), "algorithmic_MB": round(by / 1e6, 2),
               "GB/s": round(by / ms / 1e6, 1), "frac_of_8TBs": round(by / ms / 1e6 / 8000, 4), "slices/s": round(B / ms * 1e3, 1),
               "kernels_ms": {k: round(v["ms"] / 10, 4) for k, v in pr.items()}}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    # headline geometry: ViT-B/16 224^2, batch 256, bf16 in -> bf16 tokens (what the full forward's first two kernels do)
    D, p, img, B = 768, 16, 224, 256
    W = (torch.randn(D, 3, p, p, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(D, generator=g).cuda()
    from vdr import ops
    for dt in (torch.bfloat16, torch.float32):
        x = torch.rand(B, 3, img, img, generator=g).to(dt).cuda()
        ms = timeit(lambda: ops.patch_embed(x, W, b, p))
        by = B * 3 * img * img * x.element_size() + D * 3 * p * p * 2 + B * 196 * D * 2
        rec = {"mode": f"ViT-B/16 224^2 {str(dt)[6:]} -> bf16 tokens", "batch": B, "ms": round(ms, 4), "algorithmic_MB": round(by / 1e6, 2),
               "GB/s": round(by / ms / 1e6, 1), "frac_of_8TBs": round(by / ms / 1e6 / 8000, 4)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
