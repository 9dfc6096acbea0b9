"""Host time against device time of one LowRankLinear forward at the headline's shapes (T = 2048, bf16): is a suffix of
replaced layers bound by the Python / ctypes front end or by its kernels?   python tools/probes/pair_host_cost.py"""
import json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from ptdeco_amd import lowrank
dev = torch.device("cuda", 0)
F = torch.nn.functional
for name, n_i, n_o, r in (("gate r32", 4096, 14336, 32), ("q r512", 4096, 4096, 512), ("o r1024", 4096, 4096, 1024),
                          ("down r1024", 14336, 4096, 1024), ("k r512", 4096, 1024, 512)):
    seq = torch.nn.Sequential(torch.nn.Linear(n_i, r, bias=False), torch.nn.Linear(r, n_o, bias=False)).to(dev).bfloat16()
    plain = torch.nn.Sequential(*list(seq.children()))
    mod = lowrank.fuse_pair(seq)
    x = torch.randn(1, 2048, n_i, device=dev).bfloat16()
    res = {}
    with torch.no_grad():
        for label, m in (("pkg", mod), ("torch", plain)):
            for _ in range(20):
                m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(300):
                m(x)
            e1.record()
            host = (time.perf_counter() - t0) / 300
            torch.cuda.synchronize()
            res[label] = {"host_us": round(host * 1e6, 1), "device_us": round(e0.elapsed_time(e1) * 1e3 / 300, 1)}
    print(name, json.dumps(res), flush=True)
