"""Developer probe: bf16 nn.Linear-layout GEMM rates by shape (random operands).  PTD_GEMM_8PH=0 / 1 / 2
selects the 128^2 kernel, the 256^2 deep-pipelined kernel in lockstep, or with the two wave groups
staggered (default)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 4096, 4096), (16384, 1024, 4096), (16384, 4096, 1024),
          (16384, 512, 4096), (2048, 4096, 4096), (4096, 14336, 4096), (4096, 4096, 14336)]
print("PTD_GEMM_8PH =", os.environ.get("PTD_GEMM_8PH", "(default 2)"))
for (M, N, K) in shapes:
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    ms = t(lambda: ops.matmul(x, w.T))
    ref = x[:256].float() @ w.float().T
    err = (ops.matmul(x, w.T)[:256].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"bf16 nt M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF  rel err {err:.1e}", flush=True)
