#!/usr/bin/env python3
"""Shader clock the chip holds under each GEMM / attention kernel of the headline: back-to-back launches of one kernel for ~1.5 s
while `rocm-smi --showclocks` is polled from a thread; prints the median sclk per kernel (the roofline's 2.5 PF is at 2.4 GHz).
   python tools/clock_probe.py"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def poll(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level.*?\((\d+)Mhz\)", txt)
            if m:
                out.append(int(m.group(1)))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.02)


def main():
    lib = C.CDLL(os.path.join(ROOT, "vit-deep-radiomics_amd", "vdr", "libvdr.so"))
    lib.vdr_op_linear.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vdr_op_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    M = 50432
    cases = []
    for name, N, K, epi, variant in (("qkv 8-phase", 2304, 768, 0, 31), ("fc1 8-phase + GELU", 3072, 768, 1, 31), ("fc1 ring4 + GELU", 3072, 768, 1, 26),
                                     ("fc2 ring4p + resid", 768, 3072, 2, 26), ("proj ring4p + resid", 768, 768, 2, 26)):
        x = torch.randn(M + 256, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").bfloat16() if epi == 2 else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        cases.append((name, 2.0 * M * N * K, lambda x=x, W=W, b=b, r=r, out=out, N=N, K=K, epi=epi, variant=variant:
                      lib.vdr_op_linear(x.data_ptr(), W.data_ptr(), b.data_ptr(), r.data_ptr() if r is not None else None, None, out.data_ptr(), M, N, K, epi, variant, st)))
    qkv = torch.randn(M, 2304, device="cuda").bfloat16()
    o = torch.empty(M, 768, device="cuda", dtype=torch.bfloat16)
    cases.append(("attention 197 tokens", 4.0 * 197 * 197 * 64 * 3072, lambda: lib.vdr_op_attention(qkv.data_ptr(), o.data_ptr(), 256, 197, 12, 0, st)))
    for name, flops, f in cases:
        assert f() == 0, name
        torch.cuda.synchronize()
        stop, clk = threading.Event(), []
        th = threading.Thread(target=poll, args=(stop, clk))
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        th.start()
        t0 = time.time()
        a.record()
        while time.time() - t0 < 1.5:
            for _ in range(50):
                f()
            n += 50
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        us = a.elapsed_time(e) * 1e3 / n
        clk.sort()
        med = clk[len(clk) // 2] if clk else 0
        tf = flops / us / 1e6
        print(f"{name:22s}: {us:7.1f} us per launch, {tf:6.0f} TF/s; sclk median {med} MHz over {len(clk)} samples "
              f"(min {clk[0] if clk else 0}, max {clk[-1] if clk else 0}); of the peak at that clock: {tf / (2500.0 * med / 2400.0) if med else 0:.3f}", flush=True)
        time.sleep(0.5)


if __name__ == "__main__":
    main()
