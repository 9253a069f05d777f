#!/usr/bin/env python3
"""CPU analysis of the MX-fp8 path under injected outlier channels (the stress case of
tests/test_model_gpu.py::test_outlier_channels_vit_g_geometry), with the MX-emulating oracle (oracle/vit_oracle.py,
emulate_bf16="mx": the kernels' quantisation points in plain torch fp32).  No GPU, no kernel: what it shows is a property
of the number format at these points.  Output kept in profiles/r04_fp8_outlier_analysis.txt.

  1. one quantisation point at a time  -> which operand costs the CLS rows their cosine
  2. CLS rows vs patch rows quantised  -> whose arithmetic it is
  3. the split ("hi + lo") remedy for the massive channels, activation and weight side, and its ideal limit (those
     channels exact)                  -> why it was not built into the kernels
  4. magnitudes along the CLS row      -> the mechanism
  5. depth: L = 4 / 12 / 24 / 40, plain and injected, bf16 and MX-fp8
"""
import functools
import importlib.util
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mx_oracle as mo  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402

spec = importlib.util.spec_from_file_location("tmg", os.path.join(ROOT, "tests", "test_model_gpu.py"))
tmg = importlib.util.module_from_spec(spec)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec.loader.exec_module(tmg)
torch.set_num_threads(min(8, os.cpu_count() or 1))
big = vo.CONFIGS["dinov2_giant14_224"]


def make(L):
    cfg = vo.VitCfg(big.img, big.patch, 3, big.dim, big.heads, L, big.mlp_hidden, act=big.act, layerscale=True)
    w0 = vo.make_weights(cfg, seed=31)
    w, ch = tmg._inject_outlier_channels(w0, cfg.layers, cfg.dim, 32, "mlp.w12")
    return cfg, w0, w, ch, vo.make_images(cfg, 2, seed=33)


def cosines(cfg, tok, ref):
    cos = F.cosine_similarity(tok.double().reshape(-1, cfg.dim), ref.double().reshape(-1, cfg.dim), dim=-1)
    m = torch.ones(cos.numel(), dtype=torch.bool)
    m[::ref.shape[1]] = False
    return cos[~m].min().item(), cos[m].min().item()


def main():
    cfg, w0, w, ch, x = make(4)
    ref = vo.forward_images(cfg, w, x)["tokens"]

    def report(name, tok):
        c, p = cosines(cfg, tok, ref)
        print(f"  {name:58s} min cos CLS rows {c:.6f}   patch rows {p:.6f}")

    print("ViT-g/14 geometry, 4 blocks, outlier channels injected (gains x30/60/100 on three channels, +300 / -120 in two residual channels of every patch token)")
    print("1. one quantisation point at a time (everything else bf16):")
    orig_attention, orig_mlp, oqs = vo.attention, vo.mlp, vo._q_split

    def attention_sel(x_, wqkv, bqkv, wproj, bproj, heads, emulate=False, qw=False):
        B, N = x_.shape[:2]
        D = wqkv.shape[0] // 3
        wq_ = mo.mx_round(vo._r(wqkv, True)) if qw else vo._r(wqkv, True)
        qkv = vo._r(x_ @ wq_.t() + bqkv, True)
        q, k, v = qkv.reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        o = vo.sdpa(q, k, v, True).transpose(1, 2).reshape(B, N, D)
        return vo._r(o, True) @ vo._r(wproj, True).t() + bproj

    def mlp_sel(x_, w_, prefix, act, emulate=False, q12w=False, qu=False, q3w=False):
        w12, w3 = w_[prefix + "w12.weight"], w_[prefix + "w3.weight"]
        w12q = mo.mx_round(vo._r(w12, True)) if q12w else vo._r(w12, True)
        w3q = mo.mx_round(vo._r(w3, True)) if q3w else vo._r(w3, True)
        a, b = (x_ @ w12q.t() + w_[prefix + "w12.bias"]).chunk(2, dim=-1)
        u = F.silu(a) * b
        u = mo.mx_round(u) if qu else vo._r(u, True)
        return u @ w3q.t() + w_[prefix + "w3.bias"]

    def run(name, qa1=False, qw1=False, qa2=False, q12w=False, qu=False, q3w=False, rows="all"):
        vo.attention = functools.partial(attention_sel, qw=qw1)
        vo.mlp = functools.partial(mlp_sel, q12w=q12w, qu=qu, q3w=q3w)
        calls = {"n": 0}

        def qs(y, e):
            calls["n"] += 1
            site1 = calls["n"] % 2 == 1
            if not ((site1 and qa1) or ((not site1) and qa2)):
                return vo._r(y, True)
            yq, out = mo.mx_round(y), vo._r(y, True).clone()
            if rows == "all":
                return yq
            if rows == "cls":
                out[:, 0] = yq[:, 0]
            else:
                out[:, 1:] = yq[:, 1:]
            return out

        vo._q_split = qs
        try:
            report(name, vo.forward_images(cfg, w, x, emulate_bf16="mx")["tokens"])
        finally:
            vo.attention, vo.mlp, vo._q_split = orig_attention, orig_mlp, oqs

    run("nothing (bf16 at every point)")
    run("LayerNorm-1 output (qkv operand)", qa1=True)
    run("qkv weight", qw1=True)
    run("LayerNorm-2 output (w12 operand)", qa2=True)
    run("w12 weight", q12w=True)
    run("u = silu(a) b (w3 operand)", qu=True)
    run("w3 weight", q3w=True)
    run("all six", True, True, True, True, True, True)
    print("2. whose rows:")
    run("LayerNorm-2 output quantised on the CLS rows only", qa2=True, rows="cls")
    run("LayerNorm-2 output quantised on the patch rows only", qa2=True, rows="patch")

    print("3. the split remedy for the two massive channels (+300 / -120): y = mx(y) + y_lo, W = mx(W) + W_lo in 64 extra K columns")
    vo.MX_SPLIT_CHANNELS = []
    report("MX-fp8 as shipped", vo.forward_images(cfg, w, x, emulate_bf16="mx")["tokens"])
    vo.MX_SPLIT_CHANNELS = [ch[3], ch[4]]
    report("split: the two massive channels", vo.forward_images(cfg, w, x, emulate_bf16="mx")["tokens"])
    vo.MX_SPLIT_CHANNELS = list(ch)
    report("split: all five injected channels", vo.forward_images(cfg, w, x, emulate_bf16="mx")["tokens"])
    vo.MX_SPLIT_CHANNELS = []
    oq, owq, owqs = vo._q, vo._wq, vo._wq_split
    CH = torch.tensor([ch[3], ch[4]])

    def qs_exact(y, e):
        h = oq(y, e).clone()
        h[..., CH] = y[..., CH]
        return h

    def wq_exact(w_, e):
        wq = owq(w_, e).clone()
        if w_.shape[1] == cfg.dim:
            wq[:, CH] = vo._r(w_, True)[:, CH]
        return wq

    vo._q_split, vo._wq, vo._wq_split = qs_exact, wq_exact, wq_exact
    try:
        report("ideal limit: those channels and their weight columns EXACT", vo.forward_images(cfg, w, x, emulate_bf16="mx")["tokens"])
    finally:
        vo._q_split, vo._wq, vo._wq_split = oqs, owq, owqs

    print("4. magnitudes along the CLS row (fp32 forward): what its final value is made of")
    for name, ww in (("plain", w0), ("injected", w)):
        xx = vo.assemble_tokens(cfg, ww, vo.patch_embed(x, ww["patch_embed.proj.weight"], ww["patch_embed.proj.bias"], cfg.patch))
        for i in range(cfg.layers):
            p = f"blocks.{i}."
            h = vo.layer_norm(xx, ww[p + "norm1.weight"], ww[p + "norm1.bias"], cfg.ln_eps)
            a = ww[p + "ls1.gamma"] * vo.attention(h, ww[p + "attn.qkv.weight"], ww[p + "attn.qkv.bias"], ww[p + "attn.proj.weight"],
                                                   ww[p + "attn.proj.bias"], cfg.heads)
            xx = xx + a
            h2 = vo.layer_norm(xx, ww[p + "norm2.weight"], ww[p + "norm2.bias"], cfg.ln_eps)
            m = ww[p + "ls2.gamma"] * vo.mlp(h2, ww, p + "mlp.", cfg.act)
            print(f"  {name:8s} block {i}: CLS row |x| {xx[0, 0].norm():6.2f}  |attention update| {a[0, 0].norm():6.2f}  |MLP update| {m[0, 0].norm():6.2f}   "
                  f"patch row |x| {xx[0, 5].norm():6.1f}  |attention| {a[0, 5].norm():5.2f}  |MLP| {m[0, 5].norm():5.2f}")
            xx = xx + m

    print("5. depth (2 images; min row cosine / rel-L2 against the fp32 oracle)")
    for L in (4, 12, 24, 40):
        cfgL, w0L, wL, _, xL = make(L)
        for name, ww in (("plain", w0L), ("injected", wL)):
            refL = vo.forward_images(cfgL, ww, xL)["tokens"]
            row = f"  L = {L:2d} {name:8s}"
            for mode, label in ((True, "bf16"), ("mx", "MX-fp8")):
                tok = vo.forward_images(cfgL, ww, xL, emulate_bf16=mode)["tokens"]
                c, pch = cosines(cfgL, tok, refL)
                rel = lambda a_, b_: ((a_.double() - b_.double()).norm() / b_.double().norm()).item()
                row += f" | {label}: CLS cos {c:.6f} relL2 {rel(tok[:, 0], refL[:, 0]):.2e}, patch cos {pch:.6f} relL2 {rel(tok[:, 1:], refL[:, 1:]):.2e}"
            print(row, flush=True)


if __name__ == "__main__":
    main()
