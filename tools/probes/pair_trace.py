"""One LowRankLinear forward in a loop (T = 2048, bf16; shape from argv: n_i n_o r [torch]) for a kernel trace:
tools/prof_gaps.sh <tag> tools/probes/pair_trace.py 4096 4096 1024"""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from ptdeco_amd import lowrank
dev = torch.device("cuda", 0)
n_i, n_o, r = (int(a) for a in sys.argv[1:4])
seq = torch.nn.Sequential(torch.nn.Linear(n_i, r, bias=False), torch.nn.Linear(r, n_o, bias=False)).to(dev).bfloat16()
mod = seq if "torch" in sys.argv else lowrank.fuse_pair(seq)
x = torch.randn(1, 2048, n_i, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(300):
        mod(x)
torch.cuda.synchronize()
