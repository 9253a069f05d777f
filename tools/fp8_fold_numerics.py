# Numeric cost of folding LayerNorm into the MX-fp8 GEMM: the A operand would be the RAW residual row x (one e8m0 scale per
# 32 values, e4m3 payload) instead of the normalised row; y = rstd * (q(x) W'^T - mean * colsum(W')) + b'.
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.mx_oracle import mx_round
torch.manual_seed(0)
D, N, T = 1536, 512, 512
W = torch.randn(N, D) * 0.02
gam = 1 + 0.1 * torch.randn(D); bet = 0.1 * torch.randn(D)
def run(name, x):
    mu = x.mean(-1, keepdim=True); var = x.var(-1, unbiased=False, keepdim=True); rs = (var + 1e-6).rsqrt()
    xh = (x - mu) * rs * gam + bet
    ref = xh @ W.T
    Wq = mx_round(W)
    cur = mx_round(xh) @ Wq.T                      # shipped: ln_mx_kernel quantises the normalised row
    Wf = mx_round(W * gam)                          # fold: gamma into the weight, then quantise
    csum = Wf.sum(-1); bf = bet @ W.T
    fold = rs * (mx_round(x) @ Wf.T - mu * csum) + bf
    def err(a): return ((a - ref).norm() / ref.norm()).item(), torch.nn.functional.cosine_similarity(a, ref, dim=-1).min().item()
    print(f"{name:34s} shipped rel-L2 {err(cur)[0]:.4f} min-cos {err(cur)[1]:.5f} | folded rel-L2 {err(fold)[0]:.4f} min-cos {err(fold)[1]:.5f}")
x0 = torch.randn(T, D)
run("unit rows, zero mean", x0)
run("row mean = 2 sigma", x0 + 2.0)
run("row mean = 10 sigma", x0 + 10.0)
x1 = x0.clone(); x1[:, 7] += 300; x1[:, 900] -= 120
run("two outlier channels (+300, -120)", x1)
x2 = x0 * (1 + 5 * torch.rand(T, 1))
run("row scales 1..6", x2)
