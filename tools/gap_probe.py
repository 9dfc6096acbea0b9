import sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ptdeco_amd import ops
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from eigh_probe import make
c = make(4096, 4096)
w = torch.linalg.eigvalsh(c.cpu())
g = (w[1:] - w[:-1]) / w.max()
print("lam min/max", w.min().item(), w.max().item())
for thr in (1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
    print(thr, "all", int((g < thr).sum()), "top half", int((g[2048:] < thr).sum()), "top quarter", int((g[3072:] < thr).sum()))
print("min gap all", g.min().item(), "top half", g[2047:].min().item())
