import sys, torch, time
sys.path.insert(0, "/root/repo")
from ptdeco_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
n_o, n_i, k = 14336, 4096, 2048
w = (torch.randn(n_o, n_i, generator=g) / n_i ** 0.5).to(dev)
x = torch.randn(8192, n_i, generator=g).to(dev).double() * torch.logspace(0, -2, n_i, dtype=torch.float64, device=dev)
ex = x.T @ x / x.shape[0]
for _ in range(2): out = ops.eigh_factored(w, ex, k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): out = ops.eigh_factored(w, ex, k)
torch.cuda.synchronize(); print("eigh_factored 14336x4096 k=2048: %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
