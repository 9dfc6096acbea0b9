"""Does an idle gap in front of a timed loop change what an MFMA-dense launch measures?  The decomposed forward at
r = 1024 (package kernels and the hipBLASLt pair) timed four ways: synchronise + 10 launches (bench.py until round 3),
no synchronisation + 10, + 30, + 100 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
F = torch.nn.functional
T, n = 16384, 4096
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(T, n, device=dev, generator=g).bfloat16()


def timed(fn, iters, sync_first, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if sync_first:
        torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for r in (256, 1024):
    a = (torch.randn(r, n, device=dev, generator=g) / 64).bfloat16()
    b = (torch.randn(n, r, device=dev, generator=g) / r ** 0.5).bfloat16()
    for name, fn in (("package", lambda: ops.lowrank_forward(x, a, b, None)), ("hipBLASLt pair", lambda: F.linear(F.linear(x, a), b))):
        row = []
        for iters, sync in ((10, True), (10, False), (30, False), (100, False), (10, True), (30, False)):
            row.append(f"{'sync+' if sync else ''}{iters}: {timed(fn, iters, sync):.1f}")
        print(f"r={r} {name:15s} us per launch  " + "  ".join(row), flush=True)
