"""Developer probe: f32 / bf16 dense GEMM and SYRK rates."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(4096, 4096, device=dev); w = torch.randn(4096, 4096, device=dev) / 64
for name, fn in (("nt", lambda: ops.matmul(x, w.T)), ("nn", lambda: ops.matmul(x, w)), ("tn", lambda: ops.matmul(x.T, w)), ("tt", lambda: ops.matmul(x.T, w.T))):
    ms = t(fn); print(f"f32 {name} 4096^3: {ms:.3f} ms  {2*4096**3/ms/1e9:.0f} TF")
e = torch.zeros(4096, 4096, dtype=torch.float64, device=dev)
ms = t(lambda: ops.syrk_accumulate(e, x, 1 / 4096)); print(f"f32 syrk: {ms:.3f} ms  {4096**3/ms/1e9:.0f} TF")
xd = torch.randn(4096, 4096, device=dev, dtype=torch.float64); wd = torch.randn(4096, 4096, device=dev, dtype=torch.float64)
for name, fn in (("nt", lambda: ops.matmul(xd, wd.T)), ("nn", lambda: ops.matmul(xd, wd)), ("tn", lambda: ops.matmul(xd.T, wd))):
    ms = t(fn, 5); print(f"f64 {name} 4096^3: {ms:.3f} ms  {2*4096**3/ms/1e9:.1f} TF")
big = torch.randn(14336, 4096, device=dev, dtype=torch.float64)
ms = t(lambda: ops.matmul(big.T, big), 3); print(f"f64 W^T W (14336x4096): {ms:.2f} ms  {2*4096*4096*14336/ms/1e9:.1f} TF")
