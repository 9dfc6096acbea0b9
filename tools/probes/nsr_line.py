import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
k = bench.kernel_lines(dev)
for n in ("nsr_f32", "nsr_bf16_vocab"):
    print(n, k[n]["ms"], k[n]["frac_of_hbm_peak"])
