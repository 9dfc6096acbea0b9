"""Developer probe: dwain on the ViT-B/16-shaped clone (dict batches, CE loss), with and without the
precompute pass.  Usage: python tools/vit_dwain.py [depth] [splits]"""
import itertools, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import ptdeco_amd
from toy_models import ViT, init_randn

dev = torch.device("cuda", 0)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
splits = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "0" else None
model = ViT(depth=depth); init_randn(model, 0); model.to(dev).eval()
g = torch.Generator().manual_seed(1)
xs = [torch.randn(16, 3, 224, 224, generator=g).to(dev) for _ in range(12)]
with torch.no_grad():
    bt = [{"x": x, "targets": model({"x": x}).argmax(-1)} for x in xs]
ce = lambda b, y: torch.nn.functional.cross_entropy(y, b["targets"], reduction="none")
trace = []
torch.cuda.synchronize(); t0 = time.perf_counter()
cfg = ptdeco_amd.dwain.decompose_in_place(module=model, device=dev, data_iterator=itertools.cycle(bt), loss_fn=ce,
    metric_iterator=itertools.cycle(bt[6:]), num_data_steps=4, num_metric_steps=2, nsr_final_threshold=0.1,
    finetune_fn=lambda m, d, n: m, precomputing_covariance_num_splits=splits, trace=trace)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n_lin = sum(1 for _, m in ViT(depth=depth).named_modules() if isinstance(m, torch.nn.Linear))
print(json.dumps({"depth": depth, "splits": splits, "layers": n_lin, "seconds": dt, "layers_per_s": n_lin / dt,
                  "candidates": len(trace), "decomposed": len(cfg)}))
