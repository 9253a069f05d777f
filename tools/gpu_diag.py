#!/usr/bin/env python3
"""First-light diagnostics on the GPU box: every kernel on a few shapes, error statistics printed
(no asserts, every failure caught) so that one gpurun call reports on all kernels at once."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch  # noqa: E402

import vdr  # noqa: E402
from vdr import ops  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402


def stat(name, got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    rel = (err.norm() / ref.norm()).item()
    print(f"  {name:40s} max|err| {err.max().item():9.3e}  relL2 {rel:9.3e}  finite {bool(torch.isfinite(got).all())}", flush=True)
    if not (rel < 3e-2):
        FAILED.append(name)
    return rel


FAILED = []


def guard(fn):
    try:
        fn()
    except Exception:
        FAILED.append(fn.__name__)
        traceback.print_exc()
        sys.stdout.flush()


def t_ln():
    for D in (192, 768, 1024):
        x = torch.randn(1003, D).bfloat16()
        g, b = 1 + 0.1 * torch.randn(D), 0.1 * torch.randn(D)
        y = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6)
        stat(f"layernorm D={D}", y, vo.layer_norm(x.float(), g, b, 1e-6))


def t_gemm_int():
    for (M, N, K) in [(128, 128, 64), (256, 256, 128), (197, 192, 192), (333, 776, 320)]:
        x = torch.randint(-2, 3, (M, K)).float()
        W = torch.randint(-2, 3, (N, K)).float()
        b = torch.randint(-3, 4, (N,)).float()
        ref = x @ W.t() + b
        for v in (0, 22, 23, 24, 26, 27, 28):
            y = ops.linear(x.bfloat16().cuda(), W.bfloat16().cuda(), b.cuda(), variant=v).float().cpu()
            nbad = int((y != ref).sum())
            print(f"  int gemm {M}x{N}x{K} variant {v}: mismatches {nbad}/{y.numel()}", flush=True)
            if nbad:
                FAILED.append(f"int gemm v{v}")
            if nbad and v == 0:
                idx = torch.nonzero(y != ref)[:8].tolist()
                print("    first bad:", [(i, y[tuple(i)].item(), ref[tuple(i)].item()) for i in idx])


def t_gemm_split():
    # many tiles per CU (several rounds of resident workgroups)
    for (M, N, K) in [(30000, 768, 128), (50432, 768, 64), (40000, 1024, 64)]:
        x = torch.randint(-2, 3, (M, K)).float()
        W = torch.randint(-2, 3, (N, K)).float()
        b = torch.randint(-3, 4, (N,)).float()
        ref = x @ W.t() + b
        y = ops.linear(x.bfloat16().cuda(), W.bfloat16().cuda(), b.cuda(), variant=26).float().cpu()
        nbad = int((y != ref).sum())
        print(f"  int gemm (split) {M}x{N}x{K}: mismatches {nbad}/{y.numel()}", flush=True)
        if nbad:
            FAILED.append("int gemm split")
            rows = torch.nonzero((y != ref).any(dim=1)).flatten()
            print("    bad rows:", rows[:5].tolist(), "...", rows[-5:].tolist(), "count", rows.numel())


def t_gemm():
    for (M, N, K) in [(591, 768, 768), (399, 2304, 768), (300, 3072, 768), (260, 768, 3072)]:
        x = torch.randn(M, K).bfloat16()
        W = (torch.randn(N, K) * 0.05).bfloat16()
        b = torch.randn(N) * 0.1
        r = torch.randn(M, N).bfloat16()
        lin = x.float() @ W.float().t() + b
        stat(f"linear {M}x{N}x{K}", ops.linear(x.cuda(), W.cuda(), b.cuda()), lin)
        stat(f"linear+gelu {M}x{N}x{K}", ops.linear(x.cuda(), W.cuda(), b.cuda(), epilogue=vdr.EPI_BIAS_GELU), vo.gelu_erf(lin))
        stat(f"linear+resid {M}x{N}x{K}", ops.linear(x.cuda(), W.cuda(), b.cuda(), resid=r.cuda(), epilogue=vdr.EPI_BIAS_RESID), lin + r.float())


def t_attn():
    for (B, N, H, v) in [(2, 197, 3, 3), (2, 197, 3, 2), (64, 197, 12, 0), (300, 150, 2, 2), (5, 100, 3, 2), (1, 64, 1, 0), (2, 33, 2, 0), (1, 257, 2, 0), (1, 577, 2, 0), (1, 197, 2, 1), (1, 400, 1, 1)]:
        qkv = torch.randn(B * N, 3 * H * 64).bfloat16()
        q, k, vv = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = torch.nn.functional.scaled_dot_product_attention(q, k, vv).transpose(1, 2).reshape(B * N, H * 64)
        stat(f"attention B{B} N{N} H{H} var{v}", ops.attention(qkv.cuda(), B, N, H, variant=v), ref)


def t_pe():
    for (img, p, D, dt) in [(224, 16, 192, torch.float32), (56, 14, 128, torch.bfloat16)]:
        x = torch.rand(2, 3, img, img).to(dt)
        W = (torch.randn(D, 3, p, p) * 0.05).bfloat16()
        b = torch.randn(D) * 0.1
        n = (img // p) ** 2
        ref = torch.nn.functional.conv2d(x.bfloat16().float(), W.float(), b, stride=p).flatten(2).transpose(1, 2).reshape(2 * n, D)
        stat(f"patch_embed {img}/{p} D{D}", ops.patch_embed(x.cuda(), W.cuda(), b.cuda(), p), ref)


def t_model():
    cfg = vo.VitCfg(64, 16, 3, 128, 2, 3, 512)
    w = vo.make_weights(cfg, seed=3, scale=0.05)
    x = vo.make_images(cfg, 5, seed=4)
    ref = vo.forward_images(cfg, w, x)
    emu = vo.forward_images(cfg, w, x, emulate_bf16=True)
    e = vdr.Engine(vdr.VdrConfig(img=64, patch=16, dim=128, heads=2, layers=3, mlp_hidden=512))
    e.load_weights(w)
    for mode, key in ((vdr.OUT_PATCH_EMBED, "patch_embed"), (vdr.OUT_TOKENS, "tokens"), (vdr.OUT_CLS, "cls")):
        got = e.forward(x.cuda(), mode)
        stat(f"model {key} vs fp32", got, ref[key])
        stat(f"model {key} vs bf16-emulated", got, emu[key])


if __name__ == "__main__":
    torch.manual_seed(0)
    print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    for f in (t_ln, t_gemm_int, t_gemm_split, t_gemm, t_attn, t_pe, t_model):
        print(f.__name__, flush=True)
        guard(f)
    torch.cuda.synchronize()
    print("diag done; failed:", FAILED, flush=True)
    sys.exit(1 if FAILED else 0)
