"""torch's own nn.Linear forward (hipBLASLt) against ops.matmul (the package's NT GEMM) at the shapes a model forward
of the metric phase runs: [T, n_in] x [n_out, n_in]^T.  Usage: python tools/linear_vs_lib.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for dtype in (torch.float32, torch.bfloat16):
    for T, n_in, n_out in ((4096, 4096, 4096), (2048, 4096, 4096), (2048, 4096, 14336), (2048, 14336, 4096),
                           (2048, 4096, 1024)):
        x = torch.randn(T, n_in, generator=g, device=dev).to(dtype)
        w = (torch.randn(n_out, n_in, generator=g, device=dev) / n_in ** 0.5).to(dtype)
        lib = timed(lambda: torch.nn.functional.linear(x, w))
        own = timed(lambda: ops.matmul(x, w.T))
        fl = 2.0 * T * n_in * n_out
        out.append({"dtype": str(dtype), "T": T, "n_in": n_in, "n_out": n_out, "torch_linear_ms": lib, "ops_matmul_ms": own,
                    "torch_tflops": fl / lib / 1e9, "ops_tflops": fl / own / 1e9})
        print(json.dumps(out[-1]), flush=True)
