"""Developer probe: ops.eigh(A, k, all_values=False) as the dwain driver calls it."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else n // 4
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn(2 * n, n, generator=g, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, device=dev, dtype=torch.float64)
a = y.T @ y / y.shape[0]
a = a + torch.eye(n, dtype=torch.float64, device=dev) * (0.01 * torch.diag(a).mean())
for all_values in (True, False):
    ops.eigh(a, k, all_values=all_values); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): w, v = ops.eigh(a, k, all_values=all_values)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    res = (a @ v - v * w[n - k:]).abs().max().item() / w[-1].item()
    print(f"n={n} k={k} all_values={all_values}: {dt*1e3:.1f} ms  resid {res:.1e}  nan eigenvalues {int(torch.isnan(w).sum())}")
