import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch, vdr
from vdr import ops
from tools.kbench import timeit
M, N, K = 50432, 3072, 768
x = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for v in (22, 23, 122, 22, 23, 122):
    for epi, nm in ((vdr.EPI_BIAS, "bias"), (vdr.EPI_BIAS_GELU, "bias+gelu")):
        med, mn = timeit(lambda: ops.linear(x, W, b, epilogue=epi, variant=v, out=out))
        print(f"fc1 shape variant {v} {nm:10s}: {med:.3f} ms", flush=True)
