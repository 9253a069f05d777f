#!/usr/bin/env python3
"""What the erf-GELU costs in each GEMM kernel: one library, the fc1 / qkv shapes of the headline through vdr_op_linear (plain
weight layout, no LayerNorm fold) with epi = bias / bias + GELU on tile variant 0 (ring4) and 31 (8-phase), interleaved.
   python tools/ab_epi.py [lib.so]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "vit-deep-radiomics_amd", "vdr", "libvdr.so"))
    lib.vdr_op_linear.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    M = 50432 + 256  # (rows past M readable: a ragged last tile of variant 31 needs them; M itself a multiple of 256 here)
    M = (M // 256) * 256
    cases = []
    for name, N, K in (("fc1", 3072, 768), ("qkv", 2304, 768)):
        x = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for variant in (0, 31):
            for epi in (0, 1):
                f = (lambda x=x, W=W, b=b, out=out, N=N, K=K, epi=epi, variant=variant:
                     lib.vdr_op_linear(x.data_ptr(), W.data_ptr(), b.data_ptr(), None, None, out.data_ptr(), M, N, K, epi, variant, st))
                rc = f()
                if rc != 0:
                    print(f"{name} variant {variant} epi {epi}: rc {rc}")
                    continue
                cases.append((f"{name} variant {variant:2d} {'bias+gelu' if epi else 'bias     '}", f))
    torch.cuda.synchronize()
    ts = [[] for _ in cases]
    for rnd in range(12):
        order = list(range(len(cases)))
        if rnd & 1:
            order.reverse()
        ev = {}
        for i in order:
            a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(6):
                cases[i][1]()
            b2.record()
            ev[i] = (a, b2)
        torch.cuda.synchronize()
        for i, (a, b2) in ev.items():
            ts[i].append(a.elapsed_time(b2) / 6)
    for (n, _), t in zip(cases, ts):
        t.sort()
        print(f"{n}: median {t[len(t) // 2] * 1e3:7.1f} us  min {t[0] * 1e3:7.1f}", flush=True)


if __name__ == "__main__":
    main()
