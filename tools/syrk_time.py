"""Event-timed bf16 / f32 covariance accumulate at the bench shapes (n = 4096 and the Llama down_proj input width)."""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench
from ptdeco_amd import ops

dev = torch.device("cuda")
out = {}
for n, T in ((4096, 4096), (4096, 16384), (14336, 4096), (2048, 4096)):
    y = torch.randn(T, n, device=dev)
    e = torch.zeros(n, n, dtype=torch.float64, device=dev)
    for name, yy in (("bf16", y.bfloat16()), ("f32", y)):
        t = bench.time_events(lambda: ops.syrk_accumulate(e, yy, 1.0 / T), iters=10)
        fl = T * n * (n + 1)
        out[f"{name}_n{n}_T{T}"] = {"ms": round(t * 1e3, 4), "tflops": round(fl / t / 1e12, 1)}
    del y, e
print(json.dumps(out, indent=1))
