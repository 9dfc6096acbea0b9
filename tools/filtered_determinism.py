"""Debug: checksums of the intermediate blocks of the filtered eigensolver over two runs (PTD_FILTER_CHECKSUM=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTD_FILTER_CHECKSUM"] = "1"
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
n, k = 2048, 512
g = torch.Generator(device=dev).manual_seed(0)
y = torch.randn(2 * n, n, generator=g, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, device=dev, dtype=torch.float64)
c = y.T @ y / (2 * n)
c = c + torch.eye(n, device=dev, dtype=torch.float64) * 0.01 * torch.diag(c).mean()
for rep in range(2):
    print(f"--- run {rep}", file=sys.stderr, flush=True)
    w, v = ops.eigh(c, k, all_values=False)
    torch.cuda.synchronize()
