"""Developer probe: h @ B^T (K = rank) alone."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for T in (16384, 65536):
    for r in (128, 256):
        h = torch.randn(T, r, device=dev, dtype=torch.bfloat16)
        b = torch.randn(4096, r, device=dev, dtype=torch.bfloat16) / r ** 0.5
        ms = t(lambda: ops.matmul(h, b.T))
        print(f"T={T} r={r}: {ms*1e3:.0f} us  write {T*4096*2/ms/1e9:.2f} TB/s  {2*T*4096*r/ms/1e9:.0f} TF", flush=True)
