"""Developer probe: two (or more) eigendecompositions issued from separate host threads on separate
streams -- do their launch chains overlap on the GPU?  Usage: python tools/eigh_threads.py [n] [threads]"""
import sys, os, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nthr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
def cov(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    y = torch.randn(2 * n, n, generator=g, device=dev, dtype=torch.float64) * torch.logspace(0, -2, n, device=dev, dtype=torch.float64)
    a = y.T @ y / y.shape[0]
    return a + torch.eye(n, dtype=torch.float64, device=dev) * (0.01 * torch.diag(a).mean())
mats = [cov(s) for s in range(nthr)]
k = n // 2
ops.eigh(mats[0], k); torch.cuda.synchronize()
t0 = time.perf_counter()
ref = [ops.eigh(m, k) for m in mats]
torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
streams = [torch.cuda.Stream(device=dev) for _ in range(nthr)]
out = [None] * nthr
def work(i):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[i]):
        out[i] = ops.eigh(mats[i], k)
    streams[i].synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
err = max((out[i][1] - ref[i][1]).abs().max().item() for i in range(nthr))
print(f"n={n} x{nthr}: sequential {t_seq*1e3:.1f} ms, threaded {t_par*1e3:.1f} ms, speed-up {t_seq/t_par:.2f}x, max |dU| {err:.1e}")
