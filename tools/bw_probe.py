"""Developer probe: plain write / copy bandwidth of this box (torch kernels), to set beside the store rate of the GEMM epilogues."""
import torch
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (134, 537, 2148):
    n = (mb * 1000 * 1000 // 2) // 4096 * 4096
    y = torch.empty(n, dtype=torch.bfloat16, device=dev); x = torch.randn(n, device=dev, dtype=torch.float32).to(torch.bfloat16)
    ms = t(lambda: y.zero_()); print(f"{mb} MB zero_: {ms*1e3:.0f} us  {mb/ms/1e3:.2f} TB/s written")
    ms = t(lambda: y.fill_(1.5)); print(f"{mb} MB fill_: {ms*1e3:.0f} us  {mb/ms/1e3:.2f} TB/s written")
    ms = t(lambda: y.copy_(x)); print(f"{mb} MB copy_: {ms*1e3:.0f} us  {mb/ms/1e3:.2f} TB/s written (+ same read)")
    y2 = y.view(-1, 4096)
    ms = t(lambda: torch.add(y2, 1.0, out=y2)); print(f"{mb} MB add in place: {ms*1e3:.0f} us  {mb/ms/1e3:.2f} TB/s each way", flush=True)
