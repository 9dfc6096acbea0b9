import cProfile, pstats, io, sys, itertools, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import ptdeco_amd
dev = torch.device("cuda", 0)
class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = torch.nn.Conv2d(3, 128, 3, stride=2, padding=1)
        self.pw = torch.nn.ModuleList([torch.nn.Conv2d(128, 256, 1), torch.nn.Conv2d(256, 512, 1, bias=False),
                                       torch.nn.Conv2d(512, 512, 1), torch.nn.Conv2d(512, 256, 1)])
        self.head = torch.nn.Linear(256, 100)
    def forward(self, x):
        x = torch.relu(self.stem(x))
        for c in self.pw:
            x = torch.relu(c(x))
        return self.head(x.mean(dim=(2, 3)))
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
xs = [torch.randn(32, 3, 56, 56, generator=g).to(dev) for _ in range(10)]
kw = dict(proportion_threshold=0.9, nsr_final_threshold=0.05, kl_final_threshold=0.01, num_data_steps=4, num_metric_steps=2, use_float64=True, use_mean=True, use_damping=True)
m = Net().to(dev).eval()
ptdeco_amd.falor.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(xs), **kw)
torch.cuda.synchronize()
m = Net().to(dev).eval()
pr = cProfile.Profile(); pr.enable()
ptdeco_amd.falor.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(xs), **kw)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
