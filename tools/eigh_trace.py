"""One eigendecomposition shape, a few calls: meant to run under `rocprofv3 --kernel-trace --stats` for the per-kernel
split of a solver route.  Usage: python tools/eigh_trace.py n k [calls]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
n, k = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator(device=dev).manual_seed(3)
scale = torch.logspace(0, -2, n, device=dev)
e = torch.zeros(n, n, dtype=torch.float64, device=dev)
for _ in range(2):
    y = torch.randn(4096, n, generator=g, device=dev) * scale
    ops.syrk_accumulate(e, y, 1.0 / 4096)
c = ops.cov_finalize(e, 2, 0.01)
for _ in range(calls):
    w, v = ops.eigh(c, k, all_values=False)
torch.cuda.synchronize()
print("ok", float(w[-1]))
