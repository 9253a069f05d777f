import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-a)/n
for mb in (32, 77, 155, 310, 1240):
    n = mb*1000*1000//2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.view(torch.int16).max())
    print(f"{mb:5d} MB  fill {mb/1e3/tf/1e3:6.2f} TB/s   copy(R+W) {2*mb/1e3/tc/1e3:6.2f} TB/s   read(max) {mb/1e3/tr/1e3:6.2f} TB/s", flush=True)
