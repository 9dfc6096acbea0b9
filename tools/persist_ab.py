"""Developer probe: the persistent form of the 256 x 256 bf16 kernel against one workgroup per tile (PTD_GEMM_8PH_PERSIST=0)
and torch.nn.functional.linear (hipBLASLt), alternating blocks in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(16384, 4096, 512), (16384, 4096, 1024), (16384, 4096, 4096), (16384, 4096, 256), (65536, 4096, 512),
                  (8192, 4096, 512), (16384, 1024, 4096), (16384, 512, 4096), (16384, 768, 4096), (32768, 512, 4096)]:
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    res = {"persist": [], "tile": [], "lib": []}
    for rep in range(3):
        os.environ["PTD_GEMM_8PH_PERSIST"] = "1"; res["persist"].append(t(lambda: ops.matmul(x, w.T)))
        os.environ["PTD_GEMM_8PH_PERSIST"] = "0"; res["tile"].append(t(lambda: ops.matmul(x, w.T)))
        res["lib"].append(t(lambda: torch.nn.functional.linear(x, w)))
    os.environ["PTD_GEMM_8PH_PERSIST"] = "1"
    if N <= 768:   # the 128 x 256 kernel against the 128 x 128 one
        res["tile"] = []
        for rep in range(3):
            os.environ["PTD_GEMM_6PH"] = "0"; res["tile"].append(t(lambda: ops.matmul(x, w.T)))
        os.environ["PTD_GEMM_6PH"] = "1"
    b = {k: min(v) for k, v in res.items()}
    print(f"M={M} N={N} K={K}: persistent {b['persist']:.1f} us ({2*M*N*K/b['persist']/1e6:.0f} TF)  one-tile {b['tile']:.1f} us  "
          f"hipBLASLt {b['lib']:.1f} us ({2*M*N*K/b['lib']/1e6:.0f} TF)", flush=True)
