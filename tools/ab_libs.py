#!/usr/bin/env python3
"""Kernel-level A/B of two builds of libvdr.so in ONE process, interleaved rounds (guide rule 24): both libraries are
dlopen'ed side by side and vdr_op_attention / vdr_op_linear_packed are called through ctypes on the same tensors.
   python tools/ab_libs.py path/to/libA.so[:variant] path/to/libB.so[:variant] [attention|gemm]
(":variant" = the tile variant handed to vdr_op_linear_packed by that side, e.g. the same library twice as lib.so:0 lib.so:31)"""
import ctypes as C
import sys

import torch


def load(path):
    lib = C.CDLL(path)
    lib.vdr_op_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vdr_op_linear_packed.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vdr_op_pack_linear_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def main():
    specs = [a.split(":") for a in sys.argv[1:3]]
    libs = [load(sp[0]) for sp in specs]
    variants = [int(sp[1]) if len(sp) > 1 else 0 for sp in specs]
    what = sys.argv[3] if len(sys.argv) > 3 else "attention"
    st = torch.cuda.current_stream().cuda_stream
    cases = []
    if what == "attention":
        for (B, N, H) in ((256, 197, 12), (64, 577, 16), (32, 257, 24)):
            qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
            out = torch.empty(B * N, H * 64, device="cuda", dtype=torch.bfloat16)
            for li, lib in enumerate(libs):
                cases.append((f"attention B{B} N{N} H{H} lib{'AB'[li]}",
                              lambda lib=lib, qkv=qkv, out=out, B=B, N=N, H=H: lib.vdr_op_attention(qkv.data_ptr(), out.data_ptr(), B, N, H, 0, st)))
    else:
        M = 50432
        for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1)):
            x = torch.randn(M, K, device="cuda").bfloat16()
            W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
            Wp = torch.empty_like(W)
            libs[0].vdr_op_pack_linear_weight(W.data_ptr(), N, K, Wp.data_ptr(), st)
            b = torch.randn(N, device="cuda")
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            for li, lib in enumerate(libs):
                cases.append((f"gemm {name} lib{'AB'[li]}", lambda lib=lib, x=x, Wp=Wp, b=b, out=out, N=N, K=K, epi=epi, v=variants[li]:
                              lib.vdr_op_linear_packed(x.data_ptr(), Wp.data_ptr(), b.data_ptr(), None, None, out.data_ptr(), M, N, K, epi, v, st)))
    for _, f in cases:
        assert f() == 0
    torch.cuda.synchronize()
    ts = [[] for _ in cases]
    for rnd in range(22):
        # the two libraries of a pair run back to back on the same tensors: alternate who goes first, or the second one
        # inherits warm caches every time (identical kernels differed by 3-7 % with a fixed order)
        order = list(range(len(cases)))
        if rnd & 1:
            order = [i ^ 1 for i in order]
        ev = {}
        for i in order:
            f = cases[i][1]
            a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); f(); b2.record(); ev[i] = (a, b2)
        torch.cuda.synchronize()
        for i, (a, b2) in ev.items():
            ts[i].append(a.elapsed_time(b2) / 2)
    for (n, _), t in zip(cases, ts):
        t = sorted(t)
        print(f"{n:36s}: median {t[len(t) // 2] * 1e3:8.1f} us  min {t[0] * 1e3:8.1f}", flush=True)


if __name__ == "__main__":
    main()
