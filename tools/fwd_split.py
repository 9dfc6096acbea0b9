"""Developer probe: the two products of the bf16 decomposed forward timed separately."""
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
for T in (4096, 16384, 65536):
    x = torch.randn(T, 4096, device=dev, dtype=torch.bfloat16)
    w = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16) / 64
    d = t(lambda: ops.matmul(x, w.T))
    print(f"T={T} dense {d*1e3:.0f} us  {2*T*4096*4096/d/1e9:.0f} TF")
    for r in (256, 512, 1024):
        a = torch.randn(r, 4096, device=dev, dtype=torch.bfloat16) / 64
        b = torch.randn(4096, r, device=dev, dtype=torch.bfloat16) / r ** 0.5
        h = ops.matmul(x, a.T)
        t1 = t(lambda: ops.matmul(x, a.T)); t2 = t(lambda: ops.matmul(h, b.T)); tf = t(lambda: ops.lowrank_forward(x, a, b, None))
        F = torch.nn.functional
        l1 = t(lambda: F.linear(x, a)); l2 = t(lambda: F.linear(h, b)); lf = t(lambda: F.linear(F.linear(x, a), b))
        print(f"  r={r}: torch/hipBLASLt x@A^T {l1*1e3:.0f} us, h@B^T {l2*1e3:.0f} us, pair {lf*1e3:.0f} us")
        mem = 2 * (T * 4096 * 2) / 5.0e12 * 1e6
        print(f"  r={r}: x@A^T {t1*1e3:.0f} us, h@B^T {t2*1e3:.0f} us, pair {tf*1e3:.0f} us (HBM floor ~{mem:.0f} us, MFMA floor {2*T*r*8192/2.5e15*1e6:.0f} us)")
