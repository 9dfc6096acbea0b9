"""Developer probe: the first product of the decomposed forward with its input rotating over buffers larger than the
Infinity Cache (x comes from HBM every launch, as in a real forward pass) against the same launch repeated on one
buffer (x served by the 256-MB cache), for the package's kernel and torch.nn.functional.linear (hipBLASLt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")
F = torch.nn.functional


def t(fn, n=24):
    for i in range(4):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
T, n = 16384, 4096
xs = [torch.randn(T, n, device=dev, generator=g).bfloat16() for _ in range(6)]      # 6 x 134 MB
for r in (256, 512, 1024):
    a = (torch.randn(r, n, device=dev, generator=g) / 64).bfloat16()
    b = (torch.randn(n, r, device=dev, generator=g) / r ** 0.5).bfloat16()
    hs = [ops.matmul(x, a.T) for x in xs]
    rows = {}
    rows["x@A^T ours"] = (min(t(lambda i: ops.matmul(xs[0], a.T)) for _ in range(3)), min(t(lambda i: ops.matmul(xs[i % 6], a.T)) for _ in range(3)))
    rows["x@A^T lib"] = (min(t(lambda i: F.linear(xs[0], a)) for _ in range(3)), min(t(lambda i: F.linear(xs[i % 6], a)) for _ in range(3)))
    rows["h@B^T ours"] = (min(t(lambda i: ops.matmul(hs[0], b.T)) for _ in range(3)), min(t(lambda i: ops.matmul(hs[i % 6], b.T)) for _ in range(3)))
    rows["h@B^T lib"] = (min(t(lambda i: F.linear(hs[0], b)) for _ in range(3)), min(t(lambda i: F.linear(hs[i % 6], b)) for _ in range(3)))
    rows["pair ours"] = (min(t(lambda i: ops.lowrank_forward(xs[0], a, b, None)) for _ in range(3)), min(t(lambda i: ops.lowrank_forward(xs[i % 6], a, b, None)) for _ in range(3)))
    rows["pair lib"] = (min(t(lambda i: F.linear(F.linear(xs[0], a), b)) for _ in range(3)), min(t(lambda i: F.linear(F.linear(xs[i % 6], a), b)) for _ in range(3)))
    print(f"r={r}: " + "; ".join(f"{k} {v[0]:.0f} / {v[1]:.0f}" for k, v in rows.items()) + "   (us: one buffer / rotating)", flush=True)
