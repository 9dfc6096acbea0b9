import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from ptdeco_amd import ops
dev = torch.device("cuda")
def cov(n):
    g = torch.Generator(device="cuda").manual_seed(n)
    y = torch.randn(2 * n + 3, n, generator=g, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, device=dev, dtype=torch.float64)
    a = y.T @ y / y.shape[0]
    return a + torch.eye(n, dtype=torch.float64, device=dev) * (0.01 * torch.diag(a).mean())
for n in (128, 192, 256, 384, 512, 768, 1024):
    a = cov(n); line = f"n={n}:"
    for m in ("jacobi", "tridiag"):
        os.environ["PTD_EIGH_METHOD"] = m
        ops.eigh(a, n // 2, all_values=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ops.eigh(a, n // 2, all_values=False)
        torch.cuda.synchronize(); line += f"  {m} {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms"
    print(line)
