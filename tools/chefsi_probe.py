"""Timings that size a filtered-subspace top-k eigensolver (f64 GEMM rates at its shapes, the small dense
eigenproblem, eigh_factored's Cholesky as a proxy): python tools/chefsi_probe.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
def tm(fn, it=5, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
g = torch.Generator(device=dev).manual_seed(0)
n = 4096
y = torch.randn(8192, n, generator=g, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, device=dev, dtype=torch.float64)
c = y.T @ y / 8192
c = c + torch.eye(n, device=dev, dtype=torch.float64) * 0.01 * torch.diag(c).mean()
out = {}
for m in (1152, 1280, 1536):
    x = torch.randn(n, m, generator=g, device=dev, dtype=torch.float64)
    t = tm(lambda: ops.matmul(c, x)); out[f"C_X_f64_m{m}_ms"] = t; out[f"C_X_f64_m{m}_tflops"] = 2 * n * n * m / t / 1e9
    t = tm(lambda: ops.matmul(x.T, x)); out[f"gram_f64_m{m}_ms"] = t; out[f"gram_f64_m{m}_tflops"] = 2 * n * m * m / t / 1e9
    r = torch.randn(m, m, generator=g, device=dev, dtype=torch.float64)
    t = tm(lambda: ops.matmul(x, r)); out[f"X_R_f64_m{m}_ms"] = t; out[f"X_R_f64_m{m}_tflops"] = 2 * n * m * m / t / 1e9
    t = tm(lambda: torch.matmul(c, x)); out[f"C_X_f64_torch_m{m}_ms"] = t
    c32, x32 = c.float(), x.float()
    t = tm(lambda: ops.matmul(c32, x32)); out[f"C_X_f32_m{m}_ms"] = t; out[f"C_X_f32_m{m}_tflops"] = 2 * n * n * m / t / 1e9
    h = x.T @ (c @ x); h = (h + h.T) / 2
    for k in (1024,):
        if k <= m:
            t = tm(lambda: ops.eigh(h, k, all_values=False), it=3, warm=1); out[f"eigh_n{m}_k{k}_ms"] = t
    gmat = x.T @ x
    t = tm(lambda: torch.linalg.cholesky(gmat), it=3, warm=1); out[f"torch_cholesky_m{m}_ms"] = t
t = tm(lambda: ops.eigh(c, 1024, all_values=False), it=3, warm=1); out["eigh_n4096_k1024_ms"] = t
t = tm(lambda: torch.linalg.eigh(c), it=1, warm=1); out["torch_linalg_eigh_n4096_ms"] = t
x4 = torch.randn(n, 4, generator=g, device=dev, dtype=torch.float64)
t = tm(lambda: ops.matmul(c, x4), it=20); out["C_x4_f64_ms"] = t
print(json.dumps(out, indent=1))
