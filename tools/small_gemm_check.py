#!/usr/bin/env python3
"""Small-M GEMMs (one MedSAM slice: M = 4096 / 4900): tile variants 26 / 28 / 29 bitwise + interleaved timing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch  # noqa: E402

import vdr  # noqa: E402
from vdr import ops  # noqa: E402

dev = "cuda"
cases = []
ok = True
for M in (4096, 4900, 256):
    for name, (N, K, epi) in {"proj": (768, 768, vdr.EPI_BIAS_RESID), "fc2": (768, 3072, vdr.EPI_BIAS_RESID), "qkv": (2304, 768, vdr.EPI_BIAS),
                              "fc1": (3072, 768, vdr.EPI_BIAS_GELU)}.items():
        x = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wp = ops.pack_linear_weight(W)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).bfloat16() if epi == vdr.EPI_BIAS_RESID else None
        ref = ops.linear(x, Wp, b, resid=r, epilogue=epi, variant=26, packed=True)
        for v in (26, 28, 29):
            out = ops.linear(x, Wp, b, resid=r, epilogue=epi, variant=v, packed=True)
            same = torch.equal(out, ref)
            ok &= same
            if not same:
                print(f"MISMATCH {name} M{M} variant {v}")
            o = torch.empty_like(ref)
            cases.append((f"{name} M{M}", v, lambda x=x, Wp=Wp, b=b, r=r, epi=epi, v=v, o=o: ops.linear(x, Wp, b, resid=r, epilogue=epi, variant=v, out=o, packed=True)))
print("ALL EQUAL" if ok else "FAILED", flush=True)
times = [[] for _ in cases]
for rnd in range(11):
    order = list(range(len(cases)))
    if rnd & 1:
        order.reverse()
    evs = []
    for i in order:
        c = cases[i]
        c[2]()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            c[2]()
        e1.record()
        evs.append((i, e0, e1))
    torch.cuda.synchronize()
    for i, e0, e1 in evs:
        times[i].append(e0.elapsed_time(e1) / 4)
for (name, v, _), ts in zip(cases, times):
    ts = sorted(ts)
    print(f"{name:12s} variant {v}: {ts[len(ts) // 2] * 1e3:7.1f} us (min {ts[0] * 1e3:.1f})", flush=True)
