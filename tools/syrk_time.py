import sys, torch
sys.path.insert(0, "/root/repo")
from ptdeco_amd import ops
dev = torch.device("cuda")
for n, T in [(4096, 4096), (14336, 2048), (3072, 3152)]:
    y = torch.randn(T, n, device=dev)
    e = torch.zeros(n, n, dtype=torch.float64, device=dev)
    for _ in range(3): ops.syrk_accumulate(e, y, 1.0 / T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.syrk_accumulate(e, y, 1.0 / T)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"n={n} T={T}: {ms:.3f} ms  {T*n*n/ms/1e9:.1f} TFLOP/s algorithmic")
