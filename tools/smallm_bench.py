import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch, vdr
from vdr import ops
from tools.kbench import timeit
for M in (4096, 4900, 16384):
    for name, N, K, epi in (("qkv", 2304, 768, vdr.EPI_BIAS), ("proj", 768, 768, vdr.EPI_BIAS_RESID), ("fc1", 3072, 768, vdr.EPI_BIAS_GELU), ("fc2", 768, 3072, vdr.EPI_BIAS_RESID)):
        x = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").bfloat16() if epi == vdr.EPI_BIAS_RESID else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = []
        for v in (22, 23, 24, 25):
            t, _ = timeit(lambda: ops.linear(x, W, b, resid=r, epilogue=epi, variant=v, out=out), iters=20)
            res.append(f"v{v} {t*1e3:6.1f} us")
        print(f"M={M:5d} {name:4s}: " + " | ".join(res), flush=True)
