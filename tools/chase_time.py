import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
from twostage_check import spd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = spd(n, n).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.tridiagonalize(a)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("PTD_CHASE_DBG", os.environ.get("PTD_CHASE_DBG", "0"), "workers", os.environ.get("PTD_CHASE_WORKERS", "-"), f"tridiagonalize total {1e3*(t1-t0):.2f} ms")
