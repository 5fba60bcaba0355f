import torch, time
N, D = 1000000, 64
x = torch.randn(N, D, dtype=torch.float64, device="cuda")
perm = torch.randperm(N, device="cuda")
ident = torch.arange(N, device="cuda")
for name, idx in (("random", perm), ("identity", ident)):
    for _ in range(3):
        y = x.index_select(0, idx)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        y = x.index_select(0, idx)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 20
    print(name, "%.1f us" % (dt * 1e6), "read+write %.2f TB/s" % (2 * N * D * 8 / dt / 1e12))
