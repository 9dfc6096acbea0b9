"""x A^T at the decomposed forward's rank 256 (16384 x 256 x 4096, bf16): five-buffer ring against four (PTD_GEMM_DEEP)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(16384, 4096, device=dev, generator=g).bfloat16() for _ in range(4)]
for r in (128, 256):
    a = (torch.randn(r, 4096, device=dev, generator=g) / 64).bfloat16()
    b = (torch.randn(4096, r, device=dev, generator=g) / r ** 0.5).bfloat16()
    t1 = min(bench.time_events(lambda: ops.matmul(xs[0], a.T), iters=30) for _ in range(3))
    it = iter(range(10 ** 9))
    t2 = min(bench.time_events(lambda: ops.matmul(xs[next(it) % 4], a.T), iters=30) for _ in range(3))
    t3 = min(bench.time_events(lambda: ops.lowrank_forward(xs[0], a, b, None), iters=30) for _ in range(3))
    print(f"PTD_GEMM_DEEP={os.environ.get('PTD_GEMM_DEEP', '5')} r={r}: x A^T {t1 * 1e6:.1f} us (one buffer) {t2 * 1e6:.1f} us (rotating), pair {t3 * 1e6:.1f} us", flush=True)
