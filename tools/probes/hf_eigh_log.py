"""Which eigensolver route each layer of the HF Llama run takes and what it costs (synchronised per call: the streams
are not overlapped here).  Usage: python tools/probes/hf_eigh_log.py [layers]"""
import os, sys, time, runpy, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PTD_EIGH_STREAMS"] = "1"
from ptdeco_amd import ops
log = []
real_eigh, real_fact = ops.eigh, ops.eigh_factored
ops.EIGH_PROFILE = []


def eigh(A, k=None, all_values=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n0 = len(ops.EIGH_PROFILE)
    out = real_eigh(A, k=k, all_values=all_values)
    torch.cuda.synchronize()
    method = ops.EIGH_PROFILE[-1]["method"] if len(ops.EIGH_PROFILE) > n0 else None
    log.append(("eigh", A.shape[0], k, method, round((time.perf_counter() - t0) * 1e3, 1)))
    return out


def fact(W, Ex, k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = real_fact(W, Ex, k)
    torch.cuda.synchronize()
    log.append(("factored", tuple(W.shape), k, "refused" if out is None else "ok", round((time.perf_counter() - t0) * 1e3, 1)))
    return out


ops.eigh, ops.eigh_factored = eigh, fact
sys.argv = ["c4_hf_llama.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "c4_hf_llama.py"), run_name="__main__")
agg = {}
for kind, n, k, m, ms in log:
    key = f"{kind} n={n} k={k} method={m}"
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += ms
for key, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:4d} x {ms / c:8.1f} ms  {key}", file=sys.stderr)
