"""Is the GEMM time quantised in rounds of 512 resident workgroups?  proj / fc2 shapes (N = 768: 3 tile columns) at M chosen
so that the 128 x 256 tile count is 2.0, 2.31 (the headline M = 50432), 2.5 and 3.0 rounds; interleaved rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch, vdr
from vdr import ops
cases = []
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072), ("qkv", 2304, 768)):
    W = ops.pack_linear_weight((torch.randn(N, K, device="cuda") * 0.05).bfloat16()); b = torch.randn(N, device="cuda")
    tn = N // 256
    for tiles in (1024, 1182, 1280, 1536, 2048):
        if name == "qkv":
            tiles *= 3
        M = tiles // tn * 128
        x = torch.randn(M, K, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        cases.append((f"{name} M={M:6d} tiles={M // 128 * tn:5d} rounds={M // 128 * tn / 512:.2f}", M * N * K * 2.0,
                      lambda x=x, W=W, b=b, out=out: ops.linear(x, W, b, epilogue=vdr.EPI_BIAS, variant=26, out=out, packed=True)))
for _, _, f in cases: f()
torch.cuda.synchronize()
ts = [[] for _ in cases]
for rnd in range(15):
    ev = []
    for _, _, f in cases:
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); f(); b2.record(); ev.append((a, b2))
    torch.cuda.synchronize()
    for i, (a, b2) in enumerate(ev): ts[i].append(a.elapsed_time(b2) / 2)
for (n, fl, _), t in zip(cases, ts):
    t = sorted(t); m = t[len(t) // 2]
    print(f"{n:44s}: {m*1e3:7.1f} us  {fl / m / 1e9:7.1f} TF", flush=True)
