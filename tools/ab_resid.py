#!/usr/bin/env python3
"""A/B of two builds of libvdr.so on the residual GEMMs (proj / fc2 shapes, packed weights, in place), interleaved in one
process:  python tools/ab_resid.py libA.so libB.so [M]"""
import ctypes as C
import sys

import torch


def load(path):
    lib = C.CDLL(path)
    lib.vdr_op_linear_packed.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vdr_op_pack_linear_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def main():
    libs = [load(sys.argv[1]), load(sys.argv[2])]
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 50432
    st = torch.cuda.current_stream().cuda_stream
    cases = []
    for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
        x = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        Wp = torch.empty_like(W)
        libs[0].vdr_op_pack_linear_weight(W.data_ptr(), N, K, Wp.data_ptr(), st)
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").bfloat16()
        outs = []
        for li, lib in enumerate(libs):
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            outs.append(out)
            cases.append((f"{name} lib{'AB'[li]}", lambda lib=lib, x=x, Wp=Wp, b=b, r=r, out=out, N=N, K=K:
                          lib.vdr_op_linear_packed(x.data_ptr(), Wp.data_ptr(), b.data_ptr(), r.data_ptr(), None, out.data_ptr(), M, N, K, 2, 0, st)))
        for _, f in cases[-2:]:
            assert f() == 0
        torch.cuda.synchronize()
        print(name, "bitwise equal across the two libraries:", bool(torch.equal(outs[0], outs[1])))
    ts = [[] for _ in cases]
    for rnd in range(16):
        order = list(range(len(cases)))
        if rnd & 1:
            order = [i ^ 1 for i in order]
        ev = {}
        for i in order:
            f = cases[i][1]
            a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(6):
                f()
            b2.record()
            ev[i] = (a, b2)
        torch.cuda.synchronize()
        for i, (a, b2) in ev.items():
            ts[i].append(a.elapsed_time(b2) / 6)
    for (name, _), t in zip(cases, ts):
        t.sort()
        print(f"{name:12s}: median {t[len(t) // 2] * 1e3:7.1f} us  min {t[0] * 1e3:7.1f}")


if __name__ == "__main__":
    main()
