"""The eigensolver on the covariance a RANDOM-weight projection produces (y = x W^T, x isotropic: a Marchenko-Pastur-like,
flat spectrum -- what the q / o layers of a random-weight Llama hand it).  Usage: python tools/probes/mp_spectrum_eigh.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n = 4096
for case in ("isotropic x", "x with a decaying scale"):
    x = torch.randn(16384, n, generator=g, device=dev)
    if case != "isotropic x":
        x = x * torch.logspace(0, -1.5, n, device=dev)
    w = torch.randn(n, n, generator=g, device=dev) / n ** 0.5
    y = (x @ w.T).double()
    c = y.T @ y / y.shape[0]
    c = c + 0.01 * torch.diag(c).mean() * torch.eye(n, device=dev, dtype=torch.float64)
    ev = torch.linalg.eigvalsh(c)
    print(case, "lambda_max / lambda_k / lambda_min:", float(ev[-1]), float(ev[-1024]), float(ev[0]), flush=True)
    for k in (1024,):
        ops.EIGH_PROFILE = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            wv, v = ops.eigh(c, k=k, all_values=False)
            torch.cuda.synchronize()
            print("  k", k, "method", ops.EIGH_PROFILE[-1]["method"], "ms", round((time.perf_counter() - t0) * 1e3, 1), flush=True)
        ops.EIGH_PROFILE = None
