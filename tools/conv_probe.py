"""Developer probe: falor and dwain on a pointwise-convolution stack (plain 1x1 convs at realistic sizes)."""
import itertools, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import ptdeco_amd

dev = torch.device("cuda", 0)
class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = torch.nn.Conv2d(3, 128, 3, stride=2, padding=1)
        self.pw = torch.nn.ModuleList([torch.nn.Conv2d(128, 256, 1), torch.nn.Conv2d(256, 512, 1, bias=False),
                                       torch.nn.Conv2d(512, 512, 1), torch.nn.Conv2d(512, 256, 1)])
        self.head = torch.nn.Linear(256, 100)
    def forward(self, d):
        x = d["x"] if isinstance(d, dict) else d
        x = torch.relu(self.stem(x))
        for c in self.pw:
            x = torch.relu(c(x))
        return self.head(x.mean(dim=(2, 3)))
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
xs = [torch.randn(32, 3, 56, 56, generator=g).to(dev) for _ in range(10)]
for method in ("falor", "dwain"):
    model = Net().to(dev).eval()
    trace = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if method == "falor":
        cfg = ptdeco_amd.falor.decompose_in_place(module=model, device=dev, data_iterator=itertools.cycle(xs), trace=trace,
            proportion_threshold=0.9, nsr_final_threshold=0.05, kl_final_threshold=0.01, num_data_steps=4,
            num_metric_steps=2, use_float64=True, use_mean=True, use_damping=True)
    else:
        with torch.no_grad():
            bt = [{"x": x, "targets": model(x).argmax(-1)} for x in xs]
        ce = lambda b, y: torch.nn.functional.cross_entropy(y, b["targets"], reduction="none")
        cfg = ptdeco_amd.dwain.decompose_in_place(module=model, device=dev, data_iterator=itertools.cycle(bt), loss_fn=ce,
            metric_iterator=itertools.cycle(bt[5:]), num_data_steps=4, num_metric_steps=2, nsr_final_threshold=0.1,
            min_rank=8, finetune_fn=lambda m, d, n: m, trace=trace)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    kinds = {k: type(model.get_submodule(k)).__name__ for k in cfg}
    print(json.dumps({"method": method, "seconds": dt, "candidates": len(trace), "decomposed": kinds}))
    with torch.no_grad():
        model(xs[0])  # the decomposed model still runs
