"""ptd_eigh_factored at the Llama gate / up shape (W 14336 x 4096 bf16, Ex 4096 x 4096, k = 2048), a few calls: meant to
run under rocprofv3 --kernel-trace --stats (tools/prof_kernels.sh) for the per-kernel split of the widening route."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
n_o, n_i, k = 14336, 4096, int(os.environ.get("K", "2048"))
g = torch.Generator(device=dev).manual_seed(1)
w = (torch.randn(n_o, n_i, generator=g, device=dev) / n_i ** 0.5).bfloat16()
scale = torch.logspace(0, -2, n_i, device=dev)
ex = torch.zeros(n_i, n_i, dtype=torch.float64, device=dev)
for _ in range(2):
    x = torch.randn(4096, n_i, generator=g, device=dev) * scale
    ops.syrk_accumulate(ex, x, 1.0 / 4096)
ex = ops.cov_finalize(ex, 2, 0.01)
import time
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ops.eigh_factored(w, ex, k)
    torch.cuda.synchronize()
    print("call", i, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
print("ok", out is not None)
