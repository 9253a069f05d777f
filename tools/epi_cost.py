import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch, vdr
from vdr import ops
from tools.kbench import timeit
M = 50432
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for rep in range(2):
        t0, _ = timeit(lambda: ops.linear(x, W, b, epilogue=vdr.EPI_BIAS, variant=122, out=out))
        t1, _ = timeit(lambda: ops.linear(x, W, b, epilogue=vdr.EPI_BIAS, variant=22, out=out))
        t2, _ = timeit(lambda: ops.linear(x, W, b, resid=r, epilogue=vdr.EPI_BIAS_RESID, variant=22, out=out))
        t3, _ = timeit(lambda: ops.linear(x, W, b, resid=r, epilogue=vdr.EPI_BIAS_RESID, variant=22, out=r))
        print(f"{name}: no-epilogue {t0:.3f}  bias+store {t1:.3f}  +resid (separate buffers) {t2:.3f}  +resid in place {t3:.3f} ms", flush=True)
