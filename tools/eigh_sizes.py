"""ptd_eigh_topk at a few orders / k (ms, HIP events over 5 calls).  Usage: python tools/eigh_sizes.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for n, k in ((768, 768), (1024, 512), (1024, 1024), (1280, 1280), (2048, 1024), (4096, 1024), (4096, 2048)):
    x = torch.randn(4 * n, n, generator=g, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, dtype=torch.float64, device=dev)
    c = x.T @ x / x.shape[0]
    for _ in range(2):
        w, v = ops.eigh(c, k=k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        w, v = ops.eigh(c, k=k)
    e1.record()
    torch.cuda.synchronize()
    r = (c @ v - v * w[-v.shape[1]:]).abs().max().item() / w.abs().max().item()
    print(json.dumps({"n": n, "k": k, "ms": e0.elapsed_time(e1) / 5, "residual_rel": r}), flush=True)
