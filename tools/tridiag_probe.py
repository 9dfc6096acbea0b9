"""Developer probe: tridiagonalisation + bisection vs LAPACK."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ptdeco_amd import ops

def cov(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(2 * n + 3, n, generator=g, dtype=torch.float64) * torch.logspace(0, -2, n, dtype=torch.float64)
    a = y.T @ y / y.shape[0]
    return a + torch.eye(n, dtype=torch.float64) * (0.01 * torch.diag(a).mean())

# f64 gemm check
for (M, N, K) in [(5, 7, 3), (64, 64, 64), (130, 257, 33), (300, 100, 1000)]:
    a = torch.randn(M, K, dtype=torch.float64); b = torch.randn(K, N, dtype=torch.float64)
    for la in "nt":
        for lb in "nt":
            ad = a.cuda() if la == "n" else a.T.contiguous().cuda().T
            bd = b.cuda() if lb == "n" else b.T.contiguous().cuda().T
            c = ops.matmul(ad, bd).cpu()
            err = (c - a @ b).abs().max().item()
            assert err < 1e-11, (M, N, K, la, lb, err)
print("gemm_f64 ok")
for n in [int(x) for x in sys.argv[1:]] or [3, 10, 64, 65, 200, 1024, 4096]:
    a = cov(n)
    ad = a.cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d, e, w = ops.tridiagonalize(ad)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    w_ref = torch.linalg.eigvalsh(a)
    T = torch.diag(d.cpu()) + torch.diag(e.cpu()[: n - 1], 1) + torch.diag(e.cpu()[: n - 1], -1)
    wT = torch.linalg.eigvalsh(T)
    print(f"n={n} {dt*1e3:.1f} ms  |eig(T)-eig(A)| {((wT - w_ref).abs().max()/w_ref.max()).item():.2e}  "
          f"|bisect-eig(T)| {((w.cpu() - wT).abs().max()/w_ref.max()).item():.2e}")

import os
if os.environ.get("PTD_EIGH_METHOD"):
    for n in [int(x) for x in sys.argv[1:]] or [64, 200, 1024, 4096]:
        a = cov(n)
        ad = a.cuda()
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            w, v = ops.eigh(ad)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        w_ref = torch.linalg.eigvalsh(a)
        v = v.cpu(); w = w.cpu()
        orth = (v.T @ v - torch.eye(n, dtype=torch.float64)).abs().max().item()
        res = (a @ v - v * w).abs().max().item() / w_ref.max().item()
        print(f"eigh[{os.environ['PTD_EIGH_METHOD']}] n={n} {dt*1e3:.1f} ms  eval err {((w - w_ref).abs().max()/w_ref.max()).item():.2e} orth {orth:.2e} resid {res:.2e}")
