"""ptd_lowrank_forward at small ranks (the gate / up modules of a decomposed Llama block end at r = 32):
python tools/probes/lowrank_small_r.py"""
import json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from ptdeco_amd import ops

dev = torch.device("cuda", 0)
out = {}
for T, n_i, n_o in ((2048, 4096, 14336), (2048, 4096, 4096), (2048, 14336, 4096), (4096, 4096, 4096), (16384, 4096, 4096)):
    x = torch.randn(T, n_i, device=dev).bfloat16()
    for r in (16, 32, 48, 64, 96, 128, 256):
        a = (torch.randn(r, n_i, device=dev) / 64).bfloat16()
        b = (torch.randn(n_o, r, device=dev) / r ** 0.5).bfloat16()
        t = bench.time_events(lambda: ops.lowrank_forward(x, a, b, None), iters=20)
        h = ops.matmul(x, a.T)
        t1 = bench.time_events(lambda: ops.matmul(x, a.T), iters=20)
        t2 = bench.time_events(lambda: ops.matmul(h, b.T), iters=20)
        lib = bench.time_events(lambda: torch.nn.functional.linear(torch.nn.functional.linear(x, a), b), iters=20)
        by = 2 * (T * n_i + T * n_o + r * (n_i + n_o))
        out[f"T{T}_{n_i}to{n_o}_r{r}"] = {"pair_us": round(t * 1e6, 1), "xAt_us": round(t1 * 1e6, 1), "hBt_us": round(t2 * 1e6, 1),
                                          "lib_pair_us": round(lib * 1e6, 1), "hbm_bound_us": round(by / 8e12 * 1e6, 1)}
        print(f"T{T}_{n_i}to{n_o}_r{r}", json.dumps(out[f"T{T}_{n_i}to{n_o}_r{r}"]), flush=True)
