"""The f32 products of the falor ViT-B/16 run (C3): T = 8 * 197 = 1576 rows (not a multiple of the tile), each tapped
layer's dense product and its rank-r pair, package against torch (hipBLASLt).
python tools/probes/vit_f32_cells.py"""
import json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from ptdeco_amd import ops
F = torch.nn.functional
dev = torch.device("cuda", 0)
T = 1576
us = lambda f: round(bench.time_events(f, iters=30) * 1e6, 1)
for name, n_i, n_o in (("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)):
    x = torch.randn(T, n_i, device=dev)
    w = torch.randn(n_o, n_i, device=dev) / n_i ** 0.5
    print(name, "dense", json.dumps({"pkg": us(lambda: ops.matmul(x, w.T)), "lib": us(lambda: F.linear(x, w))}), flush=True)
    for r in (32, 96, 160, 288, 416):
        a = torch.randn(r, n_i, device=dev) / n_i ** 0.5
        b = torch.randn(n_o, r, device=dev) / r ** 0.5
        bias = torch.randn(n_o, device=dev)
        print(name, f"r{r}", json.dumps({"pkg": us(lambda: ops.lowrank_forward(x, a, b, bias)),
                                        "lib": us(lambda: F.linear(F.linear(x, a), b, bias))}), flush=True)
# bf16 dense layers of the Llama block at T = 2048 (ptd_gemm_ws against ptd_gemm against the library)
from ptdeco_amd import ops as _o
for (M, N, K) in ((2048, 1024, 4096), (2048, 4096, 4096), (2048, 14336, 4096), (2048, 4096, 14336)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    res = {}
    for flag in (True, False):
        _o._GEMM_WS = flag
        res["ws" if flag else "plain"] = us(lambda: ops.matmul(x, w.T))
    _o._GEMM_WS = True
    res["lib"] = us(lambda: F.linear(x, w))
    print("bf16", (M, N, K), json.dumps(res), flush=True)
