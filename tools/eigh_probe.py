"""Developer probe (not part of the product): time / accuracy of ptd_eigh on a C2-like matrix."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops

def make(n, T, seed=0):
    g = torch.Generator().manual_seed(seed)
    scale = torch.logspace(0, -2, n)
    w = (torch.randn(n, n, generator=g) / n**0.5)
    e = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    for _ in range(4):
        x = (torch.randn(T, n, generator=g) * scale).cuda()
        y = ops.matmul(x, w.cuda().T)
        ops.syrk_accumulate(e, y, 1.0 / T)
    return ops.cov_finalize(e, 4, 0.01)

if __name__ == "__main__":
  for n in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
      c = make(n, 4096)
      torch.cuda.synchronize()
      ops.EIGH_PROFILE = []
      t0 = time.perf_counter(); w, v = ops.eigh(c); torch.cuda.synchronize(); dt = time.perf_counter() - t0
      p = ops.EIGH_PROFILE[0]; ops.EIGH_PROFILE = None
      orth = (v.T @ v - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max().item()
      res = (c @ v - v * w).abs().max().item() / w.max().item()
      print(f"n={n} wall {dt*1e3:.1f} ms sweeps {p['sweeps']} phase ms {[round(x,1) for x in p['ms']]} orth {orth:.2e} resid {res:.2e}")
      if n <= 2048:
          wr = torch.linalg.eigvalsh(c.cpu())
          print("   eigenvalue err", ((w.cpu() - wr).abs().max() / wr.max()).item())
