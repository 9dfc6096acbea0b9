"""C4 (BASELINE configs[3]) per-shape timing: dwain on ONE layer of each Llama-3-8B linear shape,
synthetic calibration ([1, 2048, n_in] tokens, D = 8 data steps, M = 2 metric steps, f64
decomposition), one MI355X.  Prints a JSON object; the 224-layer figure is an extrapolation
(32 blocks x {q, o: 4096->4096; k, v: 4096->1024; gate, up: 4096->14336; down: 14336->4096})."""
import copy, itertools, json, sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ptdeco_amd
from ptdeco_amd import ops

dev = torch.device("cuda", 0)
dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32


class One(torch.nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.lin = torch.nn.Linear(n_in, n_out, bias=False)

    def forward(self, d):
        return self.lin(d["x"])


def ce(batch, logits):
    return torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]),
                                             batch["targets"].reshape(-1), reduction="none")


out = {"dtype": str(dtype), "tokens_per_step": 2048, "D": 8, "M": 2}
total = 0.0
for name, n_in, n_out, count in (("q_o", 4096, 4096, 64), ("k_v", 4096, 1024, 64), ("gate_up", 4096, 14336, 64),
                                 ("down", 14336, 4096, 32)):
    g = torch.Generator().manual_seed(1)
    m0 = One(n_in, n_out)
    with torch.no_grad():
        m0.lin.weight.copy_(torch.randn(n_out, n_in, generator=g) / n_in**0.5)
    m0.to(dev).to(dtype)
    scale = torch.logspace(0, -2, n_in)
    xs = [(torch.randn(1, 2048, n_in, generator=g) * scale).to(dev).to(dtype) for _ in range(10)]
    with torch.no_grad():
        bt = [{"x": x, "targets": m0({"x": x}).argmax(-1)} for x in xs]
    kw = dict(num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0, decompose_in_float64=True)

    def step():
        m = copy.deepcopy(m0)
        return ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(bt), loss_fn=ce,
                                                   metric_iterator=itertools.cycle(bt[8:]),
                                                   finetune_fn=lambda mm, d, n: mm, **kw)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); cfg = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ops.EIGH_PROFILE = []  # one more, untimed, pass with per-launch events for the eigensolver line
    step(); torch.cuda.synchronize()
    prof, ops.EIGH_PROFILE = ops.EIGH_PROFILE, None
    line = {"n_in": n_in, "n_out": n_out, "ms_per_layer": dt * 1e3, "layers_per_s": 1.0 / dt}
    if prof:
        p = prof[0]
        line["eigh"] = {"n": p["n"], "k": p["k"], "method": {0: "jacobi", 1: "tridiagonal", 2: "two-stage", 3: "filtered subspace iteration"}[p["method"]],
                        "ms_in_profiled_pass": p["total_ms"]}
        if p["method"] == 1 and p["ms"][0] > 0.0:   # (orders up to 2048 run in the resident kernels: no SYMV launch)
            line["eigh"]["symv_gbps"] = p["work"][0] / (p["ms"][0] * 1e-3) / 1e9
            line["eigh"]["symv_ms"] = p["ms"][0]
            line["eigh"]["symv_launches"] = p["launches"][0]
    else:
        line["eigh"] = {"route": "factored: W Ex W^T through an n_in-sized problem (ptd_eigh_factored)"}
    out[name] = line
    total += dt * count
    print(name, json.dumps(line), file=sys.stderr)
out["extrapolated_224_layers_s"] = total
out["extrapolated_layers_per_s_1gpu"] = 224 / total
print(json.dumps(out))
