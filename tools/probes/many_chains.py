"""Probe (round 5): throughput of N concurrent (4096, 2048) direct eigensolver chains, N = 1 .. 8, on as many measured
streams as the runtime has distinct hardware queues (GPU_MAX_HW_QUEUES=8 in the environment gives eight)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptdeco_amd import ops, _engine as eng
dev = torch.device("cuda", 0)
n, k = 4096, 2048
def cov(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    scale = torch.logspace(0, -2, n, device=dev)
    e = torch.zeros(n, n, dtype=torch.float64, device=dev)
    for _ in range(2):
        y = torch.randn(4096, n, generator=g, device=dev) * scale
        ops.syrk_accumulate(e, y, 1.0 / 4096)
    return ops.cov_finalize(e, 2, 0.01)
mats = [cov(s) for s in range(8)]
ops.eigh(mats[0], k, all_values=False); torch.cuda.synchronize()
for nch in (1, 2, 3, 4, 6, 8):
    jobs = [lambda m=m: ops.eigh(m, k, all_values=False) for m in mats[:nch]]
    os.environ["PTD_EIGH_STREAMS"] = str(nch)
    eng.run_concurrently(jobs, dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): eng.run_concurrently(jobs, dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    got = len(eng.chain_streams(dev, nch)) if nch > 1 else 1
    print(f"{nch} chains on {got} streams: {dt*1e3:.1f} ms = {dt*1e3/nch:.1f} ms per matrix", flush=True)
