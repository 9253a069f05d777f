"""GPU parity of every HIP kernel, called through the C ABI (vdr_op_*), against the CPU oracle /
the matching torch fp32 op on the same bf16-rounded inputs (SURVEY.md §8c "O2").

Tolerances (stated per test): kernels accumulate in fp32 and store bf16, so against an fp32
reference computed from the SAME bf16 inputs the error budget is one bf16 rounding of the output
(2^-8 relative) plus fp32 summation-order noise.
"""
import math

import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -8


@pytest.fixture(scope="module")
def ops():
    import vdr  # noqa: F401
    from vdr import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return _ops


def _bf(x):
    return x.to(torch.bfloat16)


def _assert_close(got, ref, rtol, atol, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    if bad.any():
        i = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} outside tol; first at {i}: got "
                             f"{got[tuple(i)].item():.6g} ref {ref[tuple(i)].item():.6g}; max err {err.max().item():.4g}")


# ---- LayerNorm ------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [192, 256, 384, 768, 1024, 1536])
@pytest.mark.parametrize("in_dt,out_dt", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                          (torch.float32, torch.bfloat16), (torch.float32, torch.float32)])
def test_layernorm(ops, D, in_dt, out_dt):
    g = torch.Generator().manual_seed(D)
    rows = 1003  # not a multiple of the 4 rows per workgroup
    x = (torch.randn(rows, D, generator=g) * 1.7 + 0.3).to(in_dt)
    gamma = 1 + 0.1 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    for eps in (1e-6, 1e-5):
        y = ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), eps, out_dt)
        ref = vo.layer_norm(x.float(), gamma, beta, eps)
        if out_dt == torch.float32:
            _assert_close(y, ref, 2e-5, 2e-5, f"LN D={D}")
        else:
            _assert_close(y, ref, BF16_EPS, 1e-3, f"LN D={D}")


# ---- Linear (GEMM + epilogues) ----------------------------------------------------------------------
def test_linear_exact_integers_catches_layout_bugs(ops):
    """Small-integer operands: every product and partial sum is exact in fp32 and every output is an
    integer <= 256 in magnitude, exactly representable in bf16 -> bit-exact equality.  W is
    asymmetric and x is the identity-like selector, so a transposed or permuted fragment cannot pass."""
    from vdr import EPI_BIAS
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(128, 128, 64), (256, 256, 128), (197, 192, 192), (333, 776, 320)]:
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-3, 4, (N,), generator=g).float()
        ref = x @ W.t() + b
        assert ref.abs().max() <= 256
        y = ops.linear(_bf(x).cuda(), _bf(W).cuda(), b.cuda(), epilogue=EPI_BIAS)
        assert torch.equal(y.float().cpu(), ref), f"integer GEMM mismatch at {(M, N, K)}"
    # identity activation: y == W^T rows
    K = 128
    x = torch.eye(K)
    W = torch.arange(96 * K).reshape(96, K).float() % 251 - 125
    y = ops.linear(_bf(x).cuda(), _bf(W).cuda(), None, epilogue=EPI_BIAS)
    assert torch.equal(y.float().cpu(), _bf(W).float().t())


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("variant", [22, 23, 24, 25, 26, 27, 28, 29])
def test_linear_every_tile_variant_exact(ops, variant, packed):
    """Every tile configuration the library ships (ring3 = LDS ring of 32-deep units, ring3k = K split across two wave
    groups, ring4 = activations staged as 64-deep pieces) on integer data, with the weight in the PyTorch layout and in
    the packed (pair-interleaved) layout: bit-exact, with bias, residual epilogue, ragged edges in M and N, and K from
    one 64-deep piece (the ring never wraps) to many."""
    from vdr import EPI_BIAS, EPI_BIAS_RESID
    g = torch.Generator().manual_seed(variant)
    for (M, N, K) in [(333, 776, 320), (64, 64, 64), (700, 1032, 128), (130, 264, 192), (257, 512, 768)]:
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-3, 4, (N,), generator=g).float()
        r = torch.randint(-4, 5, (M, N), generator=g).float()
        ref = x @ W.t() + b
        Wd = _bf(W).cuda()
        if packed:
            Wd = ops.pack_linear_weight(Wd)
        y = ops.linear(_bf(x).cuda(), Wd, b.cuda(), epilogue=EPI_BIAS, variant=variant, packed=packed)
        # integer sums are exact in fp32; the only rounding is the final one to bf16 (a no-op while |y| <= 256)
        assert torch.equal(y.float().cpu(), _bf(ref).float()), f"variant {variant} bias {(M, N, K)}"
        y = ops.linear(_bf(x).cuda(), Wd, b.cuda(), resid=_bf(r).cuda(), epilogue=EPI_BIAS_RESID, variant=variant, packed=packed)
        assert torch.equal(y.float().cpu(), _bf(ref + r).float()), f"variant {variant} resid {(M, N, K)}"


def test_linear_rejects_unknown_and_ablation_variants(ops):
    """The variant argument of the C ABI selects a shipped tile configuration; diagnostic encodings (>= 100, tuning
    builds) and retired numbers are an error, never a silent garbage result."""
    import vdr
    x = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
    tuning = bool(vdr.load().vdr_tuning_build())  # a tools/ build accepts the ablation encodings
    for bad in (5, 21, 31, 32) + (() if tuning else (122, 822)):  # (31 exists, but not for a 64 x 64 problem)
        with pytest.raises(vdr.VdrError):
            ops.linear(x, x, None, variant=bad)


def test_linear_full_size_packed_equals_plain_bitwise(ops):
    """At the headline shapes (M = 50432) on random data: packed-weight ring4 (the forward's default) == plain-layout
    ring3, bit for bit -- same products, same summation order, only the operand staging differs."""
    from vdr import EPI_BIAS, EPI_BIAS_GELU
    g = torch.Generator().manual_seed(11)
    M = 50432
    for (N, K, epi) in [(768, 768, EPI_BIAS), (3072, 768, EPI_BIAS_GELU), (768, 3072, EPI_BIAS)]:
        x = _bf(torch.randn(M, K, generator=g)).cuda()
        W = _bf(torch.randn(N, K, generator=g) * 0.05).cuda()
        b = torch.randn(N, generator=g).cuda()
        a = ops.linear(x, W, b, epilogue=epi, variant=22)
        Wp = ops.pack_linear_weight(W)
        for v in (22, 26, 27):
            assert torch.equal(a, ops.linear(x, Wp, b, epilogue=epi, variant=v, packed=True)), (N, K, v)


def test_linear_8phase_variant_exact_and_bitwise(ops):
    """Tile variant 31 (csrc/gemm_8p.hip: 256 x 256 x 64 tiles, one persistent 8-wave workgroup per CU, two M halves of the
    waves ping-ponging load and MFMA segments, output stored from the accumulator layout in whole lines): (i) integer data
    -> bit-exact against the fp32 reference, EPI_BIAS and the exact structure of every tile position (several tiles per
    workgroup, a partial last round, K of 4 and of many K-tiles); (ii) random data at the headline shapes -> the bits of
    ring3 / ring4 (same products, same summation order, same epilogue formula), erf-GELU included; (iii) launches it
    does not take (too few tiles, ragged M / N, a residual epilogue) are refused, not mis-run."""
    from vdr import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID
    g = torch.Generator().manual_seed(31)
    for (M, N, K) in [(256 * 64, 2048, 256), (256 * 128, 1024, 384), (256 * 512, 256, 256), (256 * 85, 2304, 1024)]:
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-3, 4, (N,), generator=g).float()
        y = ops.linear(_bf(x).cuda(), _bf(W).cuda(), b.cuda(), epilogue=EPI_BIAS, variant=31)
        ref = x.cuda() @ W.cuda().t() + b.cuda()
        assert torch.equal(y.float(), _bf(ref.cpu()).cuda().float()), (M, N, K)
        y0 = ops.linear(_bf(x).cuda(), _bf(W).cuda(), None, epilogue=EPI_BIAS, variant=31)  # no bias: the launcher's zeros
        assert torch.equal(y0.float(), _bf((x.cuda() @ W.cuda().t()).cpu()).cuda().float()), (M, N, K, "no bias")
    M = 50432
    for (N, K, epi) in [(2304, 768, EPI_BIAS), (3072, 768, EPI_BIAS_GELU), (1536, 1536, EPI_BIAS), (3072, 768, EPI_BIAS)]:
        x = _bf(torch.randn(M, K, generator=g)).cuda()
        W = _bf(torch.randn(N, K, generator=g) * 0.05).cuda()
        b = torch.randn(N, generator=g).cuda()
        a = ops.linear(x, W, b, epilogue=epi, variant=22)
        y = torch.full_like(a, 7.0)
        for _ in range(3):  # three launches in a row: the kernel leaves no state behind
            ops.linear(x, W, b, epilogue=epi, variant=31, out=y)
            assert torch.equal(a, y), (N, K, epi)
        assert torch.equal(a, ops.linear(x, ops.pack_linear_weight(W), b, epilogue=epi, variant=26, packed=True)), (N, K, epi)
    xs = _bf(torch.randn(4096, 768, generator=g)).cuda()
    Ws = _bf(torch.randn(768, 768, generator=g)).cuda()
    for bad in (dict(x=xs, W=Ws),                                           # 48 tiles: not worth a persistent launch
                dict(x=_bf(torch.randn(256 * 67, 768, generator=g)).cuda(), W=_bf(torch.randn(2304, 768, generator=g)).cuda()),  # 603 tiles = 2.4 rounds: the last one 35 % full
                dict(x=_bf(torch.randn(50432 + 8, 768, generator=g)).cuda(), W=_bf(torch.randn(2304, 768, generator=g)).cuda()),   # M % 256
                dict(x=_bf(torch.randn(50432, 768, generator=g)).cuda(), W=_bf(torch.randn(2304 + 64, 768, generator=g)).cuda())):  # N % 256
        with pytest.raises(RuntimeError):
            ops.linear(bad["x"], bad["W"], None, variant=31)
    with pytest.raises(RuntimeError):
        big = _bf(torch.randn(50432, 768, generator=g)).cuda()
        ops.linear(big, _bf(torch.randn(768, 768, generator=g)).cuda(), None, resid=big, epilogue=EPI_BIAS_RESID, variant=31)


def test_linear_many_tiles_exact(ops):
    """Shapes with more than two tiles per CU (several rounds of workgroups, a partial last round); integer data ->
    bit-exact against the fp32 reference, every row."""
    from vdr import EPI_BIAS
    g = torch.Generator().manual_seed(3)
    for (M, N, K) in [(30000, 768, 128), (50432, 768, 64)]:
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-3, 4, (N,), generator=g).float()
        for v, packed in ((0, False), (26, True), (27, True)):
            Wd = _bf(W).cuda()
            y = ops.linear(_bf(x).cuda(), ops.pack_linear_weight(Wd) if packed else Wd, b.cuda(), epilogue=EPI_BIAS, variant=v, packed=packed)
            assert torch.equal(y.float().cpu(), x @ W.t() + b), (M, N, K, v)


def test_linear_persistent_residual_epilogue_exact(ops):
    """More tiles than the chip holds workgroups (two per CU) and the residual epilogue: ring4 then runs its persistent
    form -- resident workgroups striding over the tile list, the argument block re-read per tile.  Integer data: exact,
    every row, ragged last tile row and a partial last stride; and bitwise equal to the one-tile-per-workgroup kernels
    (ring3, which has no persistent form)."""
    from vdr import EPI_BIAS_RESID
    g = torch.Generator().manual_seed(29)
    for (M, N, K) in [(30011, 768, 128), (50432, 768, 192), (9000, 3000, 64)]:
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-3, 4, (N,), generator=g).float()
        r = torch.randint(-4, 5, (M, N), generator=g).float()
        Wd = _bf(W).cuda()
        y = ops.linear(_bf(x).cuda(), ops.pack_linear_weight(Wd), b.cuda(), resid=_bf(r).cuda(), epilogue=EPI_BIAS_RESID, variant=26, packed=True)
        assert torch.equal(y.float().cpu(), x @ W.t() + b + r), (M, N, K)
        y3 = ops.linear(_bf(x).cuda(), Wd, b.cuda(), resid=_bf(r).cuda(), epilogue=EPI_BIAS_RESID, variant=22)
        assert torch.equal(y, y3)


SHAPES = [(197 * 3, 768, 768), (197 * 2 + 5, 2304, 768), (300, 3072, 768), (260, 768, 3072), (197, 192, 192),
          (1000, 576, 192), (64, 1024, 1024), (257 * 2, 1536, 1536)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_bias(ops, M, N, K):
    from vdr import EPI_BIAS
    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.05)
    b = torch.randn(N, generator=g) * 0.1
    ref = x.float() @ W.float().t() + b
    y = ops.linear(x.cuda(), W.cuda(), b.cuda(), epilogue=EPI_BIAS)
    _assert_close(y, ref, BF16_EPS, 2e-3 * math.sqrt(K / 768), f"linear {M}x{N}x{K}")
    y0 = ops.linear(x.cuda(), W.cuda(), None, epilogue=EPI_BIAS)
    _assert_close(y0, ref - b, BF16_EPS, 2e-3 * math.sqrt(K / 768), "linear nobias")


@pytest.mark.parametrize("M,N,K", [(197 * 2, 3072, 768), (130, 768, 192)])
def test_linear_gelu(ops, M, N, K):
    from vdr import EPI_BIAS_GELU
    g = torch.Generator().manual_seed(7)
    x = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.08)
    b = torch.randn(N, generator=g) * 0.1
    ref = vo.gelu_erf(x.float() @ W.float().t() + b)
    assert torch.allclose(ref, torch.nn.functional.gelu(x.float() @ W.float().t() + b), atol=1e-5, rtol=1e-5)
    y = ops.linear(x.cuda(), W.cuda(), b.cuda(), epilogue=EPI_BIAS_GELU)
    _assert_close(y, ref, BF16_EPS, 2e-3, "linear+gelu")


def test_gelu_tail_accuracy(ops):
    """erf-GELU on a sweep of pre-activations incl. the negative tail: x = 0, W = 0, so the
    pre-activation of column c is exactly the fp32 bias vals[c]."""
    from vdr import EPI_BIAS_GELU
    K = 64
    vals = torch.linspace(-8, 8, 64 * 8)
    # per-column bias sweep: column c gets pre-activation vals[c]
    N = vals.numel()
    Wz = torch.zeros(N, K)
    y = ops.linear(_bf(torch.zeros(4, K)).cuda(), _bf(Wz).cuda(), vals.cuda(), epilogue=EPI_BIAS_GELU)
    ref = torch.nn.functional.gelu(vals.double()).float().expand(4, N)
    _assert_close(y, ref, BF16_EPS, 5e-7, "gelu sweep")


@pytest.mark.parametrize("M,N,K,ls", [(197 * 2, 768, 768, False), (333, 768, 3072, True), (70, 192, 768, False)])
def test_linear_residual(ops, M, N, K, ls):
    from vdr import EPI_BIAS_RESID
    g = torch.Generator().manual_seed(11)
    x = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.05)
    b = torch.randn(N, generator=g) * 0.1
    r = _bf(torch.randn(M, N, generator=g))
    gamma = (1 + 0.2 * torch.randn(N, generator=g)) if ls else None
    lin = x.float() @ W.float().t() + b
    ref = r.float() + (gamma * lin if ls else lin)
    y = ops.linear(x.cuda(), W.cuda(), b.cuda(), resid=r.cuda(), gamma=gamma.cuda() if ls else None,
                   epilogue=EPI_BIAS_RESID)
    _assert_close(y, ref, BF16_EPS, 3e-3 * math.sqrt(K / 768), "linear+resid")
    # in place (y aliases resid), the way the forward uses it
    rr = r.cuda().clone()
    ops.linear(x.cuda(), W.cuda(), b.cuda(), resid=rr, gamma=gamma.cuda() if ls else None, epilogue=EPI_BIAS_RESID, out=rr)
    assert torch.equal(rr.cpu(), y.cpu())


def test_linear_swiglu(ops):
    from vdr import EPI_SWIGLU
    g = torch.Generator().manual_seed(13)
    M, D, F = 200, 384, 512
    x = _bf(torch.randn(M, D, generator=g))
    w12 = _bf(torch.randn(2 * F, D, generator=g) * 0.05)
    b12 = torch.randn(2 * F, generator=g) * 0.1
    y12 = x.float() @ w12.float().t() + b12
    a, b = y12.chunk(2, dim=-1)
    ref = torch.nn.functional.silu(a) * b
    wp, bp = ops.pack_w12(w12, b12)
    y = ops.linear(x.cuda(), wp.cuda(), bp.cuda(), epilogue=EPI_SWIGLU)
    assert y.shape == (M, F)
    _assert_close(y, ref, BF16_EPS, 2e-3, "swiglu")


# ---- attention ---------------------------------------------------------------------------------------
def _attn_ref(qkv, B, N, H):
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    return o.transpose(1, 2).reshape(B * N, H * 64)


@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (1, 197, 12), (3, 32, 2), (2, 33, 1), (1, 64, 2), (2, 100, 2),
                                   (1, 224, 1), (2, 257, 2), (1, 288, 1), (1, 5, 1), (2, 51, 4)])
def test_attention_single_chunk(ops, B, N, H):
    g = torch.Generator().manual_seed(B * 1000 + N)
    qkv = _bf(torch.randn(B * N, 3 * H * 64, generator=g))
    ref = _attn_ref(qkv, B, N, H)
    o = ops.attention(qkv.cuda(), B, N, H)
    # P is rounded to bf16 before P.V and the output is stored as bf16: 2^-8 relative on O(1) values
    _assert_close(o, ref, 2 * BF16_EPS, 6e-3, f"attention B{B} N{N} H{H}")


@pytest.mark.parametrize("B,N,H", [(43, 197, 12), (2, 129, 3), (3, 224, 2), (5, 160, 1), (1, 193, 7)])
def test_attention_persistent_kernels_match_one_shot_bitwise(ops, B, N, H):
    """Sequences of 129..224 tokens have three single-chunk kernels: one workgroup per (image, head) (variant 3), the
    persistent kernel (2) and the persistent kernel with its loader wave (4; the library's choice from 512 items on --
    (43, 197, 12) is 516).  Same arithmetic in the same order: bitwise equal, and within bf16 of the fp32 reference."""
    g = torch.Generator().manual_seed(B * 131 + N)
    qkv = _bf(torch.randn(B * N, 3 * H * 64, generator=g))
    ref = _attn_ref(qkv, B, N, H)
    outs = {v: ops.attention(qkv.cuda(), B, N, H, variant=v) for v in (3, 2, 4, 0)}
    for v in (2, 4, 0):
        assert torch.equal(outs[v], outs[3]), f"variant {v} B{B} N{N} H{H}"
    _assert_close(outs[0], ref, 2 * BF16_EPS, 6e-3, f"attention(persistent) B{B} N{N} H{H}")


@pytest.mark.parametrize("B,N,H", [(1, 577, 2), (2, 300, 1), (1, 1024, 1), (1, 197, 2), (1, 129, 1)])
def test_attention_online_softmax_chunks(ops, B, N, H):
    g = torch.Generator().manual_seed(N)
    qkv = _bf(torch.randn(B * N, 3 * H * 64, generator=g))
    ref = _attn_ref(qkv, B, N, H)
    o = ops.attention(qkv.cuda(), B, N, H, variant=1)
    _assert_close(o, ref, 2 * BF16_EPS, 6e-3, f"attention(online) B{B} N{N} H{H}")


def test_attention_rescale_branch_is_exercised(ops):
    """Force the running max to jump at a later key chunk (guide rule: a rare data-dependent branch
    needs an input that takes it): one key far along the sequence matches the queries strongly."""
    B, N, H = 1, 400, 1
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * N, 3 * 64, generator=g) * 0.3
    q = qkv[:, :64]
    qkv[390, 64:128] = q.mean(0) * 0 + 4.0 * torch.sign(q[7])  # key 390 (4th chunk) aligned with query 7
    qkv[7, :64] = 3.0 * torch.sign(q[7])
    qkv = _bf(qkv)
    ref = _attn_ref(qkv, B, N, H)
    o = ops.attention(qkv.cuda(), B, N, H, variant=1)
    _assert_close(o, ref, 2 * BF16_EPS, 6e-3, "attention rescale")


def test_attention_softmax_is_shift_invariant_and_rows_sum_to_one(ops):
    """Size-independent property at the BASELINE shape: with V = all-ones the output must be 1."""
    B, N, H = 4, 197, 12
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g)
    qkv[:, 2 * H * 64:] = 1.0
    o = ops.attention(_bf(qkv).cuda(), B, N, H)
    _assert_close(o, torch.ones(B * N, H * 64), 0.0, 4e-3, "rows sum to one")


# ---- patch embed ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("img,p,D,dt", [(224, 16, 192, torch.float32), (224, 16, 768, torch.bfloat16),
                                        (56, 14, 384, torch.float32), (84, 14, 128, torch.bfloat16),
                                        (32, 8, 64, torch.float32),
                                        # 37 patches per row: the LDS im2col's second workgroup of a row holds 5 of 32
                                        (518, 14, 64, torch.float32), (518, 14, 64, torch.bfloat16)])
def test_patch_embed(ops, img, p, D, dt):
    g = torch.Generator().manual_seed(img + D)
    B = 3
    x = torch.rand(B, 3, img, img, generator=g).to(dt)
    W = _bf(torch.randn(D, 3, p, p, generator=g) * 0.05)
    b = torch.randn(D, generator=g) * 0.1
    n = (img // p) ** 2
    ref = torch.nn.functional.conv2d(_bf(x).float(), W.float(), b, stride=p).flatten(2).transpose(1, 2).reshape(B * n, D)
    y = ops.patch_embed(x.cuda(), W.cuda(), b.cuda(), p)
    _assert_close(y, ref, BF16_EPS, 2e-3, f"patch_embed {img}/{p}")
    # with pos-embed and the CLS-row layout the full forward uses (row 0 of every image untouched)
    pos = torch.randn(n + 1, D, generator=g) * 0.02
    y2 = ops.patch_embed(x.cuda(), W.cuda(), b.cuda(), p, pos=pos.cuda(), row_stride=n + 1, row_offset=1)
    y2 = y2.float().cpu().reshape(B, n + 1, D)
    assert torch.equal(y2[:, 0], torch.zeros(B, D))
    _assert_close(y2[:, 1:].reshape(B * n, D), ref + pos[1:].repeat(B, 1), BF16_EPS, 2e-3, "patch_embed+pos")


@pytest.mark.parametrize("img,p,D,B", [(224, 16, 768, 5), (64, 8, 192, 3), (96, 32, 256, 7), (224, 16, 200, 2)])
def test_patch_embed_gathered_from_images_exact(ops, img, p, D, B):
    """bf16 images with p in {8, 16, 32} take the im2col-free path (the GEMM's operand loader gathers pixel runs from the
    NCHW images); integer pixels and weights make the result exact, so every (token, channel, ky, kx) -> operand mapping
    is pinned bit for bit -- and it must equal the im2col path (fp32 images of the same values) bit for bit as well.
    Ragged token counts (B * n not a multiple of the 128-row tile), D not a multiple of the tile width."""
    g = torch.Generator().manual_seed(img * 7 + p)
    x = torch.randint(-3, 4, (B, 3, img, img), generator=g).float()
    W = torch.randint(-2, 3, (D, 3, p, p), generator=g).float()
    b = torch.randint(-3, 4, (D,), generator=g).float()
    n = (img // p) ** 2
    ref = torch.nn.functional.conv2d(x, W, b, stride=p).flatten(2).transpose(1, 2).reshape(B * n, D)
    assert ref.abs().max() < 16384  # (bf16 output: compare after the same rounding)
    y = ops.patch_embed(x.bfloat16().cuda(), W.bfloat16().cuda(), b.cuda(), p)
    assert torch.equal(y.float().cpu(), _bf(ref).float()), f"gathered patch embed {img}/{p}"
    y_col = ops.patch_embed(x.cuda(), W.bfloat16().cuda(), b.cuda(), p)  # fp32 images: im2col + GEMM
    assert torch.equal(y, y_col)
    pos = torch.randint(-2, 3, (n + 1, D), generator=g).float()
    y2 = ops.patch_embed(x.bfloat16().cuda(), W.bfloat16().cuda(), b.cuda(), p, pos=pos.cuda(), row_stride=n + 1, row_offset=1)
    y2 = y2.float().cpu().reshape(B, n + 1, D)
    assert torch.equal(y2[:, 1:].reshape(B * n, D), _bf(ref + pos[1:].repeat(B, 1)).float())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_patch_embed_p14_exact(ops, dt):
    """p = 14 (DINOv2 / ViT-L/14 / ViT-g/14) goes through the LDS im2col (coalesced pixel pairs scattered into [patch][K]
    rows, K padded 588 -> 640 with zeros): integer data pins every pixel's place exactly, for both pixel types, with a
    ragged last workgroup per patch row (37 = 32 + 5) and more than one image."""
    g = torch.Generator().manual_seed(14)
    B, img, p, D = 2, 518, 14, 72
    x = torch.randint(-3, 4, (B, 3, img, img), generator=g).float()
    W = torch.randint(-2, 3, (D, 3, p, p), generator=g).float()
    b = torch.randint(-3, 4, (D,), generator=g).float()
    n = (img // p) ** 2
    ref = torch.nn.functional.conv2d(x, W, b, stride=p).flatten(2).transpose(1, 2).reshape(B * n, D)
    y = ops.patch_embed(x.to(dt).cuda(), W.bfloat16().cuda(), b.cuda(), p)
    assert torch.equal(y.float().cpu(), _bf(ref).float())


# ---- SAM / MedSAM attention with decomposed relative position bias ------------------------------------
def _relpos_attn_ref(qkv, rel_h, rel_w, B, S, H):
    from oracle import sam_oracle as so
    q, k, v = qkv.float().reshape(B, S * S, 3, H, 64).permute(2, 0, 3, 1, 4)
    attn = (q * 0.125) @ k.transpose(-1, -2)
    Rh, Rw = so.rel_table(S, rel_h), so.rel_table(S, rel_w)
    rq = q.reshape(B, H, S, S, 64)
    rh = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh)
    rw = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw)
    attn = (attn.view(B, H, S, S, S, S) + rh[..., :, None] + rw[..., None, :]).view(B, H, S * S, S * S)
    o = torch.softmax(attn, dim=-1) @ v
    return o.transpose(1, 2).reshape(B * S * S, H * 64)


@pytest.mark.parametrize("B,S,H", [(3, 4, 2), (5, 7, 1), (2, 10, 2), (4, 14, 3), (25, 14, 12)])
def test_attention_relpos_windows(ops, B, S, H):
    g = torch.Generator().manual_seed(S * 100 + B)
    qkv = _bf(torch.randn(B * S * S, 3 * H * 64, generator=g))
    rel_h = torch.randn(2 * S - 1, 64, generator=g) * 0.1
    rel_w = torch.randn(2 * S - 1, 64, generator=g) * 0.1
    ref = _relpos_attn_ref(qkv, rel_h, rel_w, B, S, H)
    out = ops.attention_relpos(qkv.cuda(), rel_h.cuda(), rel_w.cuda(), B, S, H)
    _assert_close(out, ref, 2 * BF16_EPS, 6e-3, f"relpos attention B{B} S{S} H{H}")


def test_attention_relpos_global_grid_64(ops):
    """The four global blocks of SAM ViT-B: 64 x 64 = 4096 tokens, online softmax over 128-key chunks
    (two grid rows per chunk) with the bias indexed from registers."""
    B, S, H = 1, 64, 2
    g = torch.Generator().manual_seed(64)
    qkv = _bf(torch.randn(B * S * S, 3 * H * 64, generator=g))
    rel_h = torch.randn(2 * S - 1, 64, generator=g) * 0.1
    rel_w = torch.randn(2 * S - 1, 64, generator=g) * 0.1
    ref = _relpos_attn_ref(qkv, rel_h, rel_w, B, S, H)
    out = ops.attention_relpos(qkv.cuda(), rel_h.cuda(), rel_w.cuda(), B, S, H)
    _assert_close(out, ref, 2 * BF16_EPS, 6e-3, "relpos attention global 64x64")


# ---- MX-fp8 (BASELINE config 5) -------------------------------------------------------------------
def _mx_ref(x):
    from oracle import mx_oracle as mx

    return mx.mx_round(x.float().cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K", [(1, 32), (70, 96), (300, 768), (513, 4096)])
def test_mx_quantize_bitexact_vs_oracle(rows, K):
    """quantise on the GPU, dequantise on the GPU: bitwise the oracle's quantise/dequantise (covers the
    scale layout, zero blocks, a wide dynamic range)."""
    from vdr import ops

    g = torch.Generator().manual_seed(rows * 7 + K)
    x = torch.randn(rows, K, generator=g) * torch.logspace(-5, 4, rows)[:, None]
    if rows > 2:
        x[1, :32] = 0.0
    xb = x.to(torch.bfloat16)
    t = ops.mx_quantize(xb.cuda())
    got = t.dequantize().cpu()
    want = _mx_ref(xb)
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (300, 192, 256), (1000, 512, 1536), (257, 1536, 4096)])
def test_linear_mx_integer_exact(M, N, K, variant):
    """small-integer operands with per-block power-of-two magnitudes: every product and partial sum is exact in
    fp32 and the result is exact in bf16 -> the MX GEMM must be BIT-exact (operand layout, scale routing,
    opsel, edge tiles)."""
    from vdr import ops

    g = torch.Generator().manual_seed(M + N + K + variant)
    # values in {-2..2} * 2^s with a per-(row, block) shift s in {0, 1, 2}: scales differ between blocks
    xs = torch.randint(0, 3, (M, K // 32, 1), generator=g)
    ws = torch.randint(0, 3, (N, K // 32, 1), generator=g)
    x = (torch.randint(-2, 3, (M, K // 32, 32), generator=g) * (2 ** xs)).reshape(M, K).float()
    w = (torch.randint(-2, 3, (N, K // 32, 32), generator=g) * (2 ** ws)).reshape(N, K).float()
    # keep |sum| small enough for bf16 to hold it exactly: thin out the operands for large K
    keep = torch.rand(M, K, generator=g) < min(1.0, 48.0 / K)
    x = x * keep
    ref = x.double() @ w.double().t()
    assert float(ref.abs().max()) < 2 ** 15
    xq = ops.mx_quantize(x.to(torch.bfloat16).cuda())
    wq = ops.mx_quantize(w.to(torch.bfloat16).cuda())
    assert torch.equal(xq.dequantize().cpu(), x) and torch.equal(wq.dequantize().cpu(), w)
    y = ops.linear_mx(xq, wq, variant=variant).float().cpu()
    want = ref.float().to(torch.bfloat16).float()
    assert torch.equal(y, want)


@pytest.mark.gpu
@pytest.mark.parametrize("epi", ["bias", "gelu", "resid", "swiglu"])
def test_linear_mx_epilogues_vs_dequantised_reference(epi):
    """random operands: the MX GEMM equals F.linear on the DEQUANTISED operands (fp32) up to one bf16 rounding
    of the output, for every fused epilogue."""
    from vdr import ops, _lib as L

    M, N, K = 777, 512, 768
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    xq, wq = ops.mx_quantize(x.cuda()), ops.mx_quantize(w.cuda())
    acc = (xq.dequantize().double() @ wq.dequantize().double().t()).cpu() + b.double()
    if epi == "bias":
        y = ops.linear_mx(xq, wq, bias=b.cuda())
        want = acc
    elif epi == "gelu":
        y = ops.linear_mx(xq, wq, bias=b.cuda(), epilogue=L.EPI_BIAS_GELU)
        want = torch.nn.functional.gelu(acc)
    elif epi == "resid":
        r = torch.randn(M, N, generator=g).to(torch.bfloat16)
        gam = 1.0 + 0.1 * torch.randn(N, generator=g)
        y = ops.linear_mx(xq, wq, bias=b.cuda(), resid=r.cuda(), gamma=gam.cuda(), epilogue=L.EPI_BIAS_RESID)
        want = r.double() + gam.double() * acc
    else:
        wp, bp = ops.pack_w12(w, b)
        wq2 = ops.mx_quantize(wp.cuda())
        acc2 = (xq.dequantize().double() @ wq2.dequantize().double().t()).cpu() + bp.double()
        # un-interleave the gate pairs of the packed layout: blocks of 64 = [32 x1 | 32 x2]
        a = acc2.reshape(M, N // 64, 2, 32)
        want = (torch.nn.functional.silu(a[:, :, 0]) * a[:, :, 1]).reshape(M, N // 2)
        y = ops.linear_mx(xq, wq2, bias=bp.cuda(), epilogue=L.EPI_SWIGLU)
    y = y.float().cpu()
    want = want.float()
    err = (y - want).abs()
    tol = 2.0 ** -8 * want.abs() + 1e-3
    assert bool((err <= tol).all()), float((err / (want.abs() + 1e-3)).max())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,D", [(5, 64), (197, 768), (300, 1536)])
def test_layernorm_mx_vs_oracle(rows, D):
    """LayerNorm with MX output: dequantised result within half an e4m3 ulp (2^-4 relative to the block
    maximum) of the fp32 LayerNorm, and bitwise the oracle quantiser applied to the kernel's own fp32 result
    on all but rounding-boundary elements."""
    from vdr import ops

    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 3 + 0.5).to(torch.bfloat16)
    gam = 1.0 + 0.1 * torch.randn(D, generator=g)
    bet = 0.1 * torch.randn(D, generator=g)
    t = ops.layernorm_mx(x.cuda(), gam.cuda(), bet.cuda(), 1e-6)
    got = t.dequantize().cpu()
    ref = torch.nn.functional.layer_norm(x.float(), (D,), gam, bet, 1e-6)
    blockmax = ref.reshape(rows, D // 32, 32).abs().amax(-1, keepdim=True).expand(-1, -1, 32).reshape(rows, D)
    assert bool(((got - ref).abs() <= blockmax * (2.0 ** -4) * 1.02 + 1e-6).all())
    want = _mx_ref(ref)
    assert float((got != want).float().mean()) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("epi", ["gelu", "swiglu"])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_linear_mx_output_requantised(epi, variant):
    """fc1 epilogue writing MX-fp8 directly (the fc2 operand): dequantised output = oracle quantiser applied to
    the exact epilogue result, up to elements whose fp32 value sits on a rounding boundary."""
    from vdr import ops, _lib as L

    M, N, K = 333, 512, 768
    g = torch.Generator().manual_seed(5 + variant)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    xq = ops.mx_quantize(x.cuda())
    if epi == "gelu":
        wq = ops.mx_quantize(w.cuda())
        acc = (xq.dequantize().double() @ wq.dequantize().double().t()).cpu() + b.double()
        want = torch.nn.functional.gelu(acc).float()
        t = ops.linear_mx(xq, wq, bias=b.cuda(), epilogue=L.EPI_BIAS_GELU, variant=variant, mx_out=True)
    else:
        wp, bp = ops.pack_w12(w, b)
        wq = ops.mx_quantize(wp.cuda())
        acc = (xq.dequantize().double() @ wq.dequantize().double().t()).cpu() + bp.double()
        a = acc.reshape(M, N // 64, 2, 32)
        want = (torch.nn.functional.silu(a[:, :, 0]) * a[:, :, 1]).reshape(M, N // 2).float()
        t = ops.linear_mx(xq, wq, bias=bp.cuda(), epilogue=L.EPI_SWIGLU, variant=variant, mx_out=True)
    got = t.dequantize().cpu()
    assert got.shape == want.shape
    ref = _mx_ref(want)
    blockmax = want.reshape(M, -1, 32).abs().amax(-1, keepdim=True).expand(-1, -1, 32).reshape(want.shape)
    assert bool(((got - want).abs() <= blockmax * (2.0 ** -4) * 1.02 + 1e-6).all())
    assert float((got != ref).float().mean()) < 5e-3
    # and it feeds the next GEMM: same result as quantising the bf16 output with the plain quantiser, within noise
    w2 = (torch.randn(256, want.shape[1], generator=g) * 0.05).to(torch.bfloat16)
    w2q = ops.mx_quantize(w2.cuda())
    y = ops.linear_mx(t, w2q).float().cpu()
    yref = (got.double() @ w2q.dequantize().double().cpu().t()).float()
    assert bool(((y - yref).abs() <= 2.0 ** -8 * yref.abs() + 1e-3).all())
