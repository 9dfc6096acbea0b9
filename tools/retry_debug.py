import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTD_JACOBI_DEBUG"] = "1"
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
n, k = 2048, 512
g = torch.Generator().manual_seed(700 + n)
y = (torch.rand(2 * n + 3, n, generator=g) * 2 - 1).double() * torch.logspace(0, -2, n, dtype=torch.float64)
a = y.T @ y / y.shape[0]
a = (a + torch.eye(n, dtype=torch.float64) * (0.01 * torch.diag(a).mean())).to(dev)
for deg in (None, "7", "5"):
    if deg: os.environ["PTD_EIGH_FILTER_FORCE_DEGREE"] = deg
    print("--- forced degree", deg, file=sys.stderr, flush=True)
    w, v = ops.eigh(a, k, all_values=False)
    torch.cuda.synchronize()
    print("resid", ((a @ v - v * w[n - k:]).norm(dim=0).max() / w[-1]).item(), file=sys.stderr, flush=True)
