"""The filtered subspace-iteration route of ptd_eigh_topk against the direct route and LAPACK on a covariance of the
headline workload's kind (n = 4096, k = 1024): accuracy and time.  python tools/filtered_probe.py [n] [k]"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else n // 4
g = torch.Generator(device=dev).manual_seed(0)
w = torch.randn(n, n, generator=g, device=dev) / n ** 0.5
e = torch.zeros(n, n, dtype=torch.float64, device=dev)
for _ in range(4):
    x = torch.randn(4096, n, generator=g, device=dev) * torch.logspace(0, -2, n, device=dev)
    ops.syrk_accumulate(e, ops.matmul(x, w.T), 1.0 / 4096)
c = ops.cov_finalize(e, 4, 0.01)

def run(flag):
    os.environ["PTD_EIGH_FILTERED"] = flag
    ops.eigh(c, k, all_values=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        wv, v = ops.eigh(c, k, all_values=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    ops.EIGH_PROFILE = []
    ops.eigh(c, k, all_values=False)
    prof, ops.EIGH_PROFILE = ops.EIGH_PROFILE, None
    return wv, v, dt, prof[0]

out = {"n": n, "k": k}
w0, v0, t0_, p0 = run("0")
w1, v1, t1_, p1 = run("1")
out["direct_ms"], out["filtered_ms"] = t0_, t1_
out["direct_profile"] = {kk: p0[kk] for kk in ("method", "ms", "launches", "total_ms")}
out["filtered_profile"] = {kk: p1[kk] for kk in ("method", "ms", "launches", "total_ms", "work")}
wr, vr = torch.linalg.eigh(c.cpu())
lmax = wr[-1].item()
for name, (ww, vv) in (("direct", (w0, v0)), ("filtered", (w1, v1))):
    ww, vv = ww.cpu(), vv.cpu()
    sgn = torch.sign((vv * vr[:, n - k:]).sum(0))
    cc = c.cpu()
    out[name] = {"eig_err": (ww[n - k:] - wr[n - k:]).abs().max().item() / lmax,
                 "resid": (cc @ vv - vv * ww[n - k:]).norm(dim=0).max().item() / lmax,
                 "orth": (vv.T @ vv - torch.eye(k, dtype=torch.float64)).abs().max().item(),
                 "max_dv": (vv * sgn - vr[:, n - k:]).abs().max().item()}
    for r in (k, k // 2, k // 16):
        d2 = 2.0 * r - 2.0 * (vv[:, k - r:].T @ vr[:, n - r:]).pow(2).sum().item()
        out[name][f"proj_err_r{r}"] = max(d2, 0.0) ** 0.5
# run to run
w2, v2 = ops.eigh(c, k, all_values=False)
out["filtered_run_to_run_max_dv"] = (v2 - v1).abs().max().item()
print(json.dumps(out, indent=1))
