import torch, time
torch.manual_seed(0)
M = 50432
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dt = t(lambda: torch.matmul(x, W.t(), out=out))
    dt2 = t(lambda: torch.nn.functional.linear(x, W, b))
    print(f"hipBLASLt (torch) M{M} N{N} K{K}: matmul {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:7.1f} TF | linear+bias {dt2*1e3:.3f} ms {2*M*N*K/dt2/1e12:7.1f} TF", flush=True)
