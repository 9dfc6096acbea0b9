import sys, torch
sys.path.insert(0, "/root/repo")
from ptdeco_amd import ops
dev = torch.device("cuda")
T, r = 16384, 256
x = torch.randn(T, 4096, device=dev, dtype=torch.bfloat16)
a = torch.randn(r, 4096, device=dev, dtype=torch.bfloat16) / 64
b = torch.randn(4096, r, device=dev, dtype=torch.bfloat16) / r ** 0.5
for _ in range(30): ops.lowrank_forward(x, a, b, None)
torch.cuda.synchronize()
