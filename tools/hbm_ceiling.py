import torch, time
n = 1<<30  # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
dt = t(lambda: b.copy_(a)); print("copy  GB/s (r+w)", 2*4*n/dt/1e9)
dt = t(lambda: a.sum()); print("read  GB/s", 4*n/dt/1e9)
dt = t(lambda: b.fill_(1.0)); print("write GB/s", 4*n/dt/1e9)
