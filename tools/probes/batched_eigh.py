"""Timing of ptd_eigh_topk_batched against single calls: python tools/probes/batched_eigh.py [n k counts...]"""
import json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from ptdeco_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else n // 2
counts = [int(c) for c in sys.argv[3:]] or [1, 2, 3, 4]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def cov(seed):
    gg = torch.Generator(device=dev).manual_seed(seed)
    y = torch.randn(2 * n + 3, n, generator=gg, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, device=dev, dtype=torch.float64)
    a = y.T @ y / y.shape[0]
    return a + torch.eye(n, dtype=torch.float64, device=dev) * (0.01 * torch.diag(a).mean())


mats = [cov(s) for s in range(max(counts))]
out = {"n": n, "k": k}
for c in counts:
    ops.eigh_batched(mats[:c], k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ops.eigh_batched(mats[:c], k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ops.EIGH_PROFILE = []
    ops.eigh_batched(mats[:c], k)
    torch.cuda.synchronize()
    prof, ops.EIGH_PROFILE = ops.EIGH_PROFILE, None
    p = prof[0]
    out[f"count{c}"] = {"ms": round(sorted(ts)[1], 2), "ms_per_matrix": round(sorted(ts)[1] / c, 2),
                        "profiled": {"total_ms": round(p["total_ms"], 2), "symv_ms": round(p["ms"][0], 2),
                                     "other_reduction_ms": round(p["ms"][1], 2), "eigenpairs_ms": round(p["ms"][3], 2),
                                     "symv_gbps_algorithmic": round(p["work"][0] / max(p["ms"][0], 1e-9) / 1e6, 1)}}
    print(json.dumps({f"count{c}": out[f"count{c}"]}), flush=True)
print(json.dumps(out))
