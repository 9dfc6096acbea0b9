"""The C5 cells where the pair is behind the library: each product alone, package against torch (hipBLASLt).
python tools/probes/fwd_cells.py"""
import json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from ptdeco_amd import ops
F = torch.nn.functional
dev = torch.device("cuda", 0)
n = 4096
for T, r in ((4096, 512), (4096, 1024), (16384, 1024), (65536, 1024), (4096, 256)):
    x = torch.randn(T, n, device=dev).bfloat16()
    a = (torch.randn(r, n, device=dev) / 64).bfloat16()
    b = (torch.randn(n, r, device=dev) / r ** 0.5).bfloat16()
    h = F.linear(x, a)
    res = {"xAt_pkg": bench.time_events(lambda: ops.matmul(x, a.T), iters=20), "xAt_lib": bench.time_events(lambda: F.linear(x, a), iters=20),
           "hBt_pkg": bench.time_events(lambda: ops.matmul(h, b.T), iters=20), "hBt_lib": bench.time_events(lambda: F.linear(h, b), iters=20),
           "pair_pkg": bench.time_events(lambda: ops.lowrank_forward(x, a, b, None), iters=20),
           "pair_lib": bench.time_events(lambda: F.linear(F.linear(x, a), b), iters=20)}
    print(f"T{T}_r{r}", json.dumps({k: round(v * 1e6, 1) for k, v in res.items()}), flush=True)
w = (torch.randn(n, n, device=dev) / 64).bfloat16()
for T in (4096, 16384, 65536):
    x = torch.randn(T, n, device=dev).bfloat16()
    print(f"dense_T{T}", json.dumps({"pkg": round(bench.time_events(lambda: ops.matmul(x, w.T), iters=20) * 1e6, 1),
                                     "lib": round(bench.time_events(lambda: F.linear(x, w), iters=20) * 1e6, 1)}), flush=True)
