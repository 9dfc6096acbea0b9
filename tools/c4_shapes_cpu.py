"""C4 (BASELINE configs[3]) per-shape CPU baseline: the CPU oracle (restatement of the reference,
torch-CPU / MKL, f32 model, f64 decomposition) on ONE layer of each Llama-3-8B linear shape with the same synthetic
inputs as tools/c4_shapes.py ([1, 2048, n_in] tokens, D = 8, M = 2), torch threads = min(physical cores, CPUs the cgroup grants).
Prints a JSON object; the 224-layer figure is an extrapolation like the GPU one.  A few minutes of CPU time."""
import itertools, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ptdeco_oracle as orc

try:
    import psutil
    physical = psutil.cpu_count(logical=False) or os.cpu_count()
except Exception:
    physical = os.cpu_count()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from cpu_quota import usable_cpus   # affinity AND cgroup quota: the GPU boxes show 256 CPUs and grant 16
usable = usable_cpus()
cores = max(1, min(physical, usable))
torch.set_num_threads(cores)
shapes = [s for s in sys.argv[1:] if not s.startswith("-")] or ["q_o", "k_v", "gate_up", "down"]


def _heartbeat():
    # one layer takes up to ~7 minutes: a line a minute on stderr shows the run is alive (a GPU box kills a command
    # that writes nothing for 7 minutes)
    import threading
    t0 = time.time()

    def beat():
        while True:
            time.sleep(60)
            print(f"[c4_shapes_cpu] running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

    threading.Thread(target=beat, daemon=True).start()


_heartbeat()


class One(torch.nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.lin = torch.nn.Linear(n_in, n_out, bias=False)

    def forward(self, d):
        return self.lin(d["x"])


def ce(batch, logits):
    return torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]),
                                             batch["targets"].reshape(-1), reduction="none")


out = {"kind": "port (oracle/ptdeco_oracle.py)", "cores": cores, "physical_cores": physical, "logical_cpus": os.cpu_count(),
       "tokens_per_step": 2048, "D": 8, "M": 2}
total = 0.0
for name, n_in, n_out, count in (("q_o", 4096, 4096, 64), ("k_v", 4096, 1024, 64), ("gate_up", 4096, 14336, 64),
                                 ("down", 14336, 4096, 32)):
    if name not in shapes:
        continue
    g = torch.Generator().manual_seed(1)
    m = One(n_in, n_out)
    with torch.no_grad():
        m.lin.weight.copy_(torch.randn(n_out, n_in, generator=g) / n_in**0.5)
    scale = torch.logspace(0, -2, n_in)
    xs = [torch.randn(1, 2048, n_in, generator=g) * scale for _ in range(10)]
    with torch.no_grad():
        bt = [{"x": x, "targets": m({"x": x}).argmax(-1)} for x in xs]
    kw = dict(num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0, decompose_in_float64=True)
    t0 = time.perf_counter()
    orc.dwain_decompose(module=m, data_iterator=itertools.cycle(bt), loss_fn=ce, metric_iterator=itertools.cycle(bt[8:]),
                        finetune_fn=None, **kw)
    dt = time.perf_counter() - t0
    out[name] = {"n_in": n_in, "n_out": n_out, "s_per_layer": dt, "layers_per_s": 1.0 / dt}
    total += dt * count
    print(name, json.dumps(out[name]), file=sys.stderr, flush=True)
if len(shapes) == 4:
    out["extrapolated_224_layers_s"] = total
    out["extrapolated_layers_per_s"] = 224 / total
print(json.dumps(out))
