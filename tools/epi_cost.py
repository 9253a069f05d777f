"""What the residual epilogue costs: the proj / fc2 shapes with EPI_BIAS (bf16-staged, write-once) against EPI_BIAS_RESID
(fp32-staged, residual read, in place or not), interleaved rounds in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-deep-radiomics_amd"))
import torch, vdr
from vdr import ops
M = 50432
cases = []
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16(); W = ops.pack_linear_weight((torch.randn(N, K, device="cuda") * 0.05).bfloat16()); b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cases.append((f"{name} bias only", lambda x=x, W=W, b=b, out=out: ops.linear(x, W, b, epilogue=vdr.EPI_BIAS, variant=26, out=out, packed=True)))
    cases.append((f"{name} bias+resid out-of-place", lambda x=x, W=W, b=b, r=r, out=out: ops.linear(x, W, b, resid=r, epilogue=vdr.EPI_BIAS_RESID, variant=26, out=out, packed=True)))
    cases.append((f"{name} bias+resid in place", lambda x=x, W=W, b=b, r=r: ops.linear(x, W, b, resid=r, epilogue=vdr.EPI_BIAS_RESID, variant=26, out=r, packed=True)))
for _, f in cases: f()
torch.cuda.synchronize()
ts = [[] for _ in cases]
for rnd in range(21):
    ev = []
    for _, f in cases:
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); f(); b2.record(); ev.append((a, b2))
    torch.cuda.synchronize()
    for i, (a, b2) in enumerate(ev): ts[i].append(a.elapsed_time(b2) / 2)
for (n, _), t in zip(cases, ts):
    t = sorted(t); print(f"{n:32s}: {t[len(t)//2]*1e3:7.1f} us (min {t[0]*1e3:.1f})", flush=True)
