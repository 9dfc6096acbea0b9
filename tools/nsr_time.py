"""ptd_nsr at the C2 logits shape (and a vocabulary-sized one): device time per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
for shape, dt in (((4, 1024, 4096), torch.float32), ((4, 1024, 4096), torch.bfloat16), ((1, 2048, 128256), torch.bfloat16)):
    y = torch.randn(shape, device=dev).to(dt)
    x = (y.float() + 0.01).to(dt)
    c = shape[-1]
    t = min(bench.time_events(lambda: ops.nsr(x, y, c), iters=30) for _ in range(3))
    by = 2 * y.numel() * y.element_size()
    print(f"{shape} {dt}: {t * 1e6:.1f} us, {by / t / 1e9:.0f} GB/s = {by / t / 8e12:.2f} of 8 TB/s", flush=True)
