"""The two products of the pair at the headline's row count (T = 2048, bf16), one by one, package against torch (hipBLASLt).
python tools/probes/t2048_products.py"""
import json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from ptdeco_amd import ops
F = torch.nn.functional
dev = torch.device("cuda", 0)
us = lambda f: round(bench.time_events(f, iters=50) * 1e6, 1)
T = 2048
for name, (M, N, K) in (("o/q x At r1024", (T, 1024, 4096)), ("o hBt r1024", (T, 4096, 1024)), ("down x At r1024", (T, 1024, 14336)),
                        ("q x At r512", (T, 512, 4096)), ("q hBt r512", (T, 4096, 512)), ("k hBt r512", (T, 1024, 512)),
                        ("dense q/o", (T, 4096, 4096)), ("dense k/v", (T, 1024, 4096)), ("dense gate", (T, 14336, 4096)),
                        ("dense down", (T, 4096, 14336))):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    print(name, (M, N, K), json.dumps({"pkg": us(lambda: ops.matmul(x, w.T)), "lib": us(lambda: F.linear(x, w))}), flush=True)
