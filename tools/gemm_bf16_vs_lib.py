"""Developer probe: the bf16 nn.Linear-layout GEMMs of the decomposed forward (BASELINE configs[4]) against
torch.nn.functional.linear (hipBLASLt) on the same operands, randn data as in bench.py."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")


def t(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(16384, 4096, 4096), (16384, 1024, 4096), (16384, 4096, 1024), (16384, 512, 4096), (16384, 4096, 512),
          (4096, 4096, 4096)]
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K**0.5).bfloat16()
    ours = t(lambda: ops.matmul(x, w.T))
    lib = t(lambda: torch.nn.functional.linear(x, w))
    print(f"M={M} N={N} K={K}: ours {ours:.4f} ms ({2*M*N*K/ours/1e9:.0f} TF)  hipBLASLt {lib:.4f} ms ({2*M*N*K/lib/1e9:.0f} TF)  ratio {ours/lib:.3f}", flush=True)
