import sys, os, torch
sys.path.insert(0, "/root/repo")
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
def ev_time(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
n, t, steps = 4096, 2048, 8
ys = [torch.randn(t, n, device=dev).to(torch.bfloat16) for _ in range(steps)]
e = torch.zeros(n, n, dtype=torch.float64, device=dev)
for dbg in ("0", "1", "2", "6", "10"):
    os.environ["PTD_SYRK_RING_DBG"] = dbg
    ms = ev_time(lambda: ops.syrk_accumulate_multi(e, ys, 1.0 / t)) / steps
    ms1 = ev_time(lambda: ops.syrk_accumulate(e, ys[0], 1.0 / t))
    print(f"dbg {dbg}: multi per step {ms*1e3:.1f} us, single {ms1*1e3:.1f} us", flush=True)
