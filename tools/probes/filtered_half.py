"""Probe (round 5): the filtered subspace iteration at k = n / 2 (block 5 n / 8) against the direct route, alone and as
three concurrent chains.  PTD_EIGH_FILTER_BLOCK_EIGHTHS=5 PTD_EIGH_FILTERED=2 python tools/probes/filtered_half.py"""
import os, sys, time, threading, torch
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
mats = [cov(s) for s in range(3)]
ops.EIGH_PROFILE = []
w, v = ops.eigh(mats[0], k, all_values=False)
p, ops.EIGH_PROFILE = ops.EIGH_PROFILE[0], None
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): w, v = ops.eigh(mats[0], k, all_values=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
res = (mats[0] @ v - v * w[n - k:]).abs().max().item() / w[-1].item()
orth = (v.T @ v - torch.eye(k, dtype=torch.float64, device=dev)).abs().max().item()
print(f"single: {dt*1e3:.1f} ms method {p['method']} phases {[round(x,1) for x in p['ms']]} launches {p['launches']} resid {res:.1e} orth {orth:.1e}", flush=True)
jobs = [lambda m=m: ops.eigh(m, k, all_values=False) for m in mats]
eng.run_concurrently(jobs, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): eng.run_concurrently(jobs, dev)
torch.cuda.synchronize()
print(f"three concurrent chains: {(time.perf_counter() - t0) / 2 * 1e3:.1f} ms", flush=True)
