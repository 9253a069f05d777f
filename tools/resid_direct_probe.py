#!/usr/bin/env python3
"""Probe of the direct residual epilogue (csrc/gemm_epi.h epilogue_resid_direct) on a TUNING build: proj / fc2 shapes through
vdr_op_linear (plain weight layout, EPI_BIAS_RESID, variant 26).  With VDR_RESID_DIRECT=1 in the environment the script hands W
with its rows permuted (w_perm_src) and the library takes the direct epilogue; without it the LDS-staged one.  Prints a
checksum of the output bits (equal in both modes = bitwise equal) and the time per launch.
   python tools/resid_direct_probe.py tools/ab_base/libvdr_tuning.so"""
import ctypes as C
import os
import sys

import torch


def perm_src(s):
    return 32 * (s >> 5) + 8 * ((s & 15) >> 2) + 4 * ((s >> 4) & 1) + (s & 3)


def main():
    lib = C.CDLL(sys.argv[1])
    lib.vdr_op_linear.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    direct = os.environ.get("VDR_RESID_DIRECT", "0") == "1"
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(5)
    for name, M, N, K in (("proj", 50432, 768, 768), ("fc2", 50432, 768, 3072), ("ragged", 50432 - 77, 768, 768)):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
        b = torch.randn(N, device="cuda", generator=g)
        gam = torch.rand(N, device="cuda", generator=g) + 0.5
        r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        Wk = W
        if direct:
            idx = torch.tensor([64 * (n // 64) + perm_src(n % 64) for n in range(N)], device="cuda")
            Wk = W[idx].contiguous()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: lib.vdr_op_linear(x.data_ptr(), Wk.data_ptr(), b.data_ptr(), r.data_ptr(), gam.data_ptr(), out.data_ptr(), M, N, K, 2, 26, st)
        assert f() == 0
        torch.cuda.synchronize()
        ref = ((x[:512].float() @ W.float().t() + b) * gam + r[:512].float())
        err = (out[:512].float() - ref).abs().max().item() / ref.abs().max().item()
        chk = int(out.view(torch.int16).to(torch.int64).sum().item()) & 0xFFFFFFFF
        ts = []
        for _ in range(10):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(6):
                f()
            e.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) / 6)
        ts.sort()
        print(f"{'direct' if direct else 'staged'} {name:7s}: median {ts[len(ts) // 2] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f}  rel err vs fp32 {err:.2e}  checksum {chk:08x}", flush=True)


if __name__ == "__main__":
    main()
