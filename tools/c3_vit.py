"""C3 (BASELINE configs[2]): falor.decompose_in_place on a ViT-B/16-shaped model (timm layer
shapes, random weights; 48 block Linears + the head, the 16x16 patch convolution is not
decomposable), synthetic calibration images [B, 3, 224, 224], one MI355X.
Usage: python tools/c3_vit.py [depth] [batch]"""
import itertools, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import ptdeco_amd
from toy_models import ViT, init_randn

dev = torch.device("cuda", 0)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = ViT(depth=depth)
init_randn(model, 0)
model.to(dev).eval()
g = torch.Generator().manual_seed(1)
pool = [torch.randn(batch, 3, 224, 224, generator=g).to(dev) for _ in range(24)]
kw = dict(proportion_threshold=0.9, nsr_final_threshold=0.05, kl_final_threshold=0.01, num_data_steps=4,
          num_metric_steps=2, use_float64=True, use_mean=True, use_damping=True)
trace = []
from ptdeco_amd import _engine as eng
if os.environ.get("PTD_PHASES"):
    eng.PHASES = eng.PhaseTimer()   # device-time split (adds two event records per span)
torch.cuda.synchronize(); t0 = time.perf_counter()
cfg = ptdeco_amd.falor.decompose_in_place(module=model, device=dev, data_iterator=itertools.cycle(pool), trace=trace, **kw)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
layers = sum(1 for n, m in ViT(depth=depth).named_modules() if isinstance(m, torch.nn.Linear))
phases = {k: round(v, 1) for k, v in eng.PHASES.totals_ms().items()} if eng.PHASES is not None else None
print(json.dumps({"phases_ms": phases, "workload": f"falor.decompose_in_place, ViT-B/16-shaped, depth {depth}, f32 model, f64 covariance + eigh, "
                              f"[{batch},3,224,224] images, D=4, M=2", "layers": layers, "seconds": dt,
                  "layers_per_s": layers / dt, "candidates_evaluated": len(trace), "decomposed": len(cfg),
                  "kept": {k: v["__meta__"]["proportion"] for k, v in list(cfg.items())[:6]}}))
