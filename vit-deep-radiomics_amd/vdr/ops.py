"""Single-operator entry points of libvdr.so (vdr_op_*), used by the per-kernel parity tests and
the kernel benchmark.  Every tensor must already live on the HIP device; nothing falls back."""
from __future__ import annotations

import torch

from . import _lib as L


def _s(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out_dtype=torch.bfloat16):
    lib = L.load()
    assert x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    L.check(lib.vdr_op_layernorm(x.data_ptr(), 1 if x.dtype == torch.bfloat16 else 0, y.data_ptr(),
                                 1 if out_dtype == torch.bfloat16 else 0, gamma.data_ptr(), beta.data_ptr(), rows, D,
                                 float(eps), _s(x)))
    return y


def pack_linear_weight(W: torch.Tensor) -> torch.Tensor:
    """W [N, K] bf16 (PyTorch layout) -> the library's packed GEMM weight layout (what vdr_finalize builds for a
    model's weights): [N/2][K/32][2][32], returned as an [N, K]-shaped opaque tensor."""
    lib = L.load()
    assert W.is_cuda and W.dtype == torch.bfloat16 and W.is_contiguous() and W.dim() == 2
    Wp = torch.empty_like(W)
    L.check(lib.vdr_op_pack_linear_weight(W.data_ptr(), W.shape[0], W.shape[1], Wp.data_ptr(), _s(W)))
    return Wp


def linear(x, W, bias=None, resid=None, gamma=None, epilogue=L.EPI_BIAS, variant=0, out=None, packed=False):
    """x [M,K] bf16, W [N,K] bf16 (for EPI_SWIGLU W/bias must already be gate-pair packed: see pack_w12).
    packed=True: W is the result of pack_linear_weight (same values, whole-line operand loads)."""
    lib = L.load()
    assert x.is_cuda and x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and x.is_contiguous() and W.is_contiguous()
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty((M, N // 2 if epilogue == L.EPI_SWIGLU else N), dtype=torch.bfloat16, device=x.device)
    fn = lib.vdr_op_linear_packed if packed else lib.vdr_op_linear
    L.check(fn(x.data_ptr(), W.data_ptr(), _p(bias), _p(resid), _p(gamma), out.data_ptr(), M, N, K, epilogue, variant, _s(x)))
    return out


class MxTensor:
    """e4m3 payload [rows, K] (uint8) + e8m0 block scales in the device layout of csrc/mx.hip."""

    def __init__(self, q: torch.Tensor, scales: torch.Tensor):
        self.q, self.scales = q, scales

    @property
    def shape(self):
        return self.q.shape

    @staticmethod
    def empty(rows: int, K: int, device):
        lib = L.load()
        return MxTensor(torch.empty((rows, K), dtype=torch.uint8, device=device),
                        torch.zeros(int(lib.vdr_mx_scale_bytes(rows, K)), dtype=torch.uint8, device=device))

    def dequantize(self) -> torch.Tensor:
        lib = L.load()
        rows, K = self.q.shape
        y = torch.empty((rows, K), dtype=torch.float32, device=self.q.device)
        L.check(lib.vdr_op_mx_dequantize(self.q.data_ptr(), self.scales.data_ptr(), rows, K, y.data_ptr(), _s(self.q)))
        return y


def mx_quantize(x: torch.Tensor) -> MxTensor:
    """bf16 [rows, K] -> MX-fp8 (block 32 along K)."""
    lib = L.load()
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[1] % 32 == 0
    rows, K = x.shape
    t = MxTensor.empty(rows, K, x.device)
    L.check(lib.vdr_op_mx_quantize(x.data_ptr(), rows, K, t.q.data_ptr(), t.scales.data_ptr(), _s(x)))
    return t


def layernorm_mx(x: torch.Tensor, gamma, beta, eps: float) -> MxTensor:
    lib = L.load()
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
    rows, D = x.shape
    t = MxTensor.empty(rows, D, x.device)
    L.check(lib.vdr_op_layernorm_mx(x.data_ptr(), gamma.float().contiguous().data_ptr(), beta.float().contiguous().data_ptr(),
                                    float(eps), rows, D, t.q.data_ptr(), t.scales.data_ptr(), _s(x)))
    return t


def linear_mx(x: MxTensor, W: MxTensor, bias=None, resid=None, gamma=None, epilogue=L.EPI_BIAS, variant=0, mx_out=False):
    """MX x [M,K] . MX W [N,K]^T on the block-scaled fp8 MFMA; returns bf16 [M,N] or (mx_out) an MxTensor."""
    lib = L.load()
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    No = N // 2 if epilogue == L.EPI_SWIGLU else N
    if mx_out:
        out = MxTensor.empty(M, No, x.q.device)
        yp, ysp = out.q.data_ptr(), out.scales.data_ptr()
    else:
        out = torch.empty((M, No), dtype=torch.bfloat16, device=x.q.device)
        yp, ysp = out.data_ptr(), None
    L.check(lib.vdr_op_linear_mx(x.q.data_ptr(), x.scales.data_ptr(), W.q.data_ptr(), W.scales.data_ptr(), _p(bias),
                                 _p(resid), _p(gamma), yp, ysp, M, N, K, epilogue, variant, _s(x.q)))
    return out


def pack_w12(w12: torch.Tensor, b12: torch.Tensor):
    """Interleave SwiGLU x1/x2 rows in blocks of 32 (the layout vdr_set_weight builds for mlp.w12)."""
    F2 = w12.shape[0]
    F = F2 // 2
    assert F % 32 == 0
    idx = torch.arange(F2)
    blk, t = idx // 64, idx % 64
    src = torch.where(t < 32, blk * 32 + t, F + blk * 32 + (t - 32))
    return w12[src].contiguous(), b12[src].contiguous()


def attention(qkv: torch.Tensor, batch: int, seq: int, heads: int, variant=0):
    lib = L.load()
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    assert qkv.shape == (batch * seq, 3 * heads * 64)
    out = torch.empty((batch * seq, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    L.check(lib.vdr_op_attention(qkv.data_ptr(), out.data_ptr(), batch, seq, heads, variant, _s(qkv)))
    return out


def patch_embed(images, weight, bias, p, pos=None, row_stride=None, row_offset=0, out=None):
    """images [B,C,H,H] fp32/bf16; weight [D,C,p,p] (any float dtype); returns bf16 [B*row_stride, D]."""
    lib = L.load()
    B, Cc, H, _ = images.shape
    D = weight.shape[0]
    g = H // p
    n = g * g
    K = Cc * p * p
    Kp = (K + 63) // 64 * 64
    Wp = torch.zeros((D, Kp), dtype=torch.bfloat16, device=images.device)
    Wp[:, :K] = weight.reshape(D, K).to(torch.bfloat16)
    col = torch.empty((B * n * Kp + 4096,), dtype=torch.bfloat16, device=images.device)
    row_stride = n if row_stride is None else row_stride
    if out is None:
        out = torch.zeros((B * row_stride, D), dtype=torch.bfloat16, device=images.device)
    images = images.contiguous()
    L.check(lib.vdr_op_patch_embed(images.data_ptr(), 1 if images.dtype == torch.bfloat16 else 0, Wp.data_ptr(),
                                   _p(bias), _p(pos), col.data_ptr(), out.data_ptr(), B, Cc, H, p, D, row_stride,
                                   row_offset, _s(images)))
    return out


def attention_relpos(qkv: torch.Tensor, rel_pos_h: torch.Tensor, rel_pos_w: torch.Tensor, batch: int, S: int, heads: int):
    """SAM attention with decomposed relative position bias over `batch` windows/grids of S x S tokens."""
    lib = L.load()
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    assert qkv.shape == (batch * S * S, 3 * heads * 64)
    assert rel_pos_h.shape == (2 * S - 1, 64) and rel_pos_w.shape == (2 * S - 1, 64)
    npad = 2 * ((2 * S - 1 + 31) // 32 * 32)
    rel = torch.empty(batch * S * S * heads * npad + npad * 32, dtype=torch.float32, device=qkv.device)
    out = torch.empty((batch * S * S, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    L.check(lib.vdr_op_attention_relpos(qkv.data_ptr(), rel_pos_h.float().contiguous().data_ptr(),
                                        rel_pos_w.float().contiguous().data_ptr(), rel.data_ptr(), out.data_ptr(), batch, S,
                                        heads, _s(qkv)))
    return out
