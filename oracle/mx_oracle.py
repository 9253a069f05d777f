"""CPU restatement of the MX-fp8 quantiser used by the fp8 path (BASELINE config 5).

TEST INFRASTRUCTURE ONLY (never imported by the product package).

The reference has no reduced-precision path at all (fp32 eager everywhere: src/tfds_dense_descriptor.py:123,
src/models_archs.py:141-147); "fp8 weights (CDNA4 fp8 MFMA)" is a BASELINE.json configuration, so the format
is build-defined and *parity with the reference is measured against its fp32 arithmetic* (cosine gate, SURVEY
§8d).  What this file pins is the quantiser itself:

    block  = 32 consecutive elements along K (OCP Microscaling "MX" block size)
    e      = ceil(log2(amax / 448)) clamped to [-126, 126]        (448 = largest finite e4m3fn)
    scale  = 2^e stored as e8m0 byte e + 127
    q      = RNE_e4m3fn(x * 2^-e)                                 (|x| 2^-e <= 448: never saturates)

e4m3fn decode / round-to-nearest-even come from torch.float8_e4m3fn on the CPU (tests/golden/e4m3fn_table.npz
holds the 256-entry decode table produced with it).
"""
from __future__ import annotations

import numpy as np
import torch

BLOCK = 32
E4M3_MAX = 448.0


def e4m3_decode_table() -> np.ndarray:
    """value of every e4m3fn byte (NaN for 0x7f / 0xff), restated from the OCP definition."""
    out = np.zeros(256, dtype=np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m * 2.0 ** -9
        else:
            v = (1.0 + m / 8.0) * 2.0 ** (e - 7)
        out[b] = -v if s else v
    return out


def scale_exponent(amax: torch.Tensor) -> torch.Tensor:
    """e = ceil(log2(amax / 448)) computed on the fp32 bit pattern exactly as the kernel does."""
    t = (amax.to(torch.float32) * np.float32(1.0 / 448.0)).contiguous()
    bits = t.view(torch.int32)
    e = ((bits >> 23) & 255) - 127 + ((bits & 0x7FFFFF) != 0).to(torch.int32)
    return e.clamp(-126, 126)


def mx_quantize(x: torch.Tensor):
    """x [..., K] (K % 32 == 0) -> (payload float8_e4m3fn [..., K], exponent int32 [..., K/32])."""
    x = x.to(torch.float32)
    K = x.shape[-1]
    assert K % BLOCK == 0
    xb = x.reshape(*x.shape[:-1], K // BLOCK, BLOCK)
    e = scale_exponent(xb.abs().amax(dim=-1))
    inv = torch.ldexp(torch.ones_like(xb[..., 0]), -e)
    q = (xb * inv[..., None]).to(torch.float8_e4m3fn)
    return q.reshape(x.shape), e


def mx_dequantize(q: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    K = q.shape[-1]
    v = q.to(torch.float32).reshape(*q.shape[:-1], K // BLOCK, BLOCK)
    return (v * torch.ldexp(torch.ones_like(v[..., 0]), e)[..., None]).reshape(q.shape)


def mx_round(x: torch.Tensor) -> torch.Tensor:
    """quantise + dequantise: what an MX operand carries into the fp32-accumulating MFMA."""
    q, e = mx_quantize(x)
    return mx_dequantize(q, e)
