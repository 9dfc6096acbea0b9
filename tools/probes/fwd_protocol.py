import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
F = torch.nn.functional
def t(fn, n, warm):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for gen in ("gpu", "cpu"):
    if gen == "gpu":
        x = torch.randn(16384, 4096, device=dev, dtype=torch.bfloat16); w = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16) / 64
    else:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(16384, 4096, generator=g).bfloat16().to(dev); w = (torch.randn(4096, 4096, generator=g) / 64).bfloat16().to(dev)
    for n in (10, 30, 100):
        print(gen, "iters", n, "ours %.0f us" % t(lambda: ops.matmul(x, w.T), n, 5), "library %.0f us" % t(lambda: F.linear(x, w), n, 5), flush=True)
    out = torch.empty(16384, 4096, device=dev, dtype=torch.bfloat16)
    print(gen, "x stats", float(x.float().std()), float(w.float().std()), float(x.float().abs().max()))
