#!/usr/bin/env python3
"""What the residual epilogue costs: proj / fc2 shapes of the headline on a TUNING build, tile variant 26 complete and with its
ablation encodings (126 = no epilogue, 826 = no output stores, 926 = both), interleaved.
   python tools/abl_resid.py tools/ab_base/libvdr_tuning.so"""
import ctypes as C, sys, torch
lib = C.CDLL(sys.argv[1])
lib.vdr_op_linear_packed.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.vdr_op_pack_linear_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
M = 50432
cases = []
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    Wp = torch.empty_like(W)
    lib.vdr_op_pack_linear_weight(W.data_ptr(), N, K, Wp.data_ptr(), st)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for v in (26, 126, 826, 926):
        f = (lambda x=x, Wp=Wp, b=b, r=r, out=out, N=N, K=K, v=v: lib.vdr_op_linear_packed(x.data_ptr(), Wp.data_ptr(), b.data_ptr(), r.data_ptr(), None, out.data_ptr(), M, N, K, 2, v, st))
        rc = f()
        if rc: print(name, v, "rc", rc); continue
        cases.append((f"{name} variant {v}", f))
torch.cuda.synchronize()
ts = [[] for _ in cases]
for rnd in range(12):
    order = list(range(len(cases)))
    if rnd & 1: order.reverse()
    ev = {}
    for i in order:
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(6): cases[i][1]()
        b2.record(); ev[i] = (a, b2)
    torch.cuda.synchronize()
    for i, (a, b2) in ev.items(): ts[i].append(a.elapsed_time(b2) / 6)
for (n, _), t in zip(cases, ts):
    t.sort(); print(f"{n}: median {t[len(t)//2]*1e3:7.1f} us  min {t[0]*1e3:7.1f}", flush=True)
