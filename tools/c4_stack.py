"""C4 (BASELINE configs[3]) end to end at reduced depth: dwain.decompose_in_place on a
Llama-3-8B-shaped stack with FULL layer widths (4096 / 1024 / 14336) and `blocks` blocks, synthetic
calibration [1, 2048, 4096], D = 8, M = 2, precomputing_covariance_num_splits = 4, one MI355X.
Usage: python tools/c4_stack.py [blocks] [bf16]"""
import itertools, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ptdeco_amd

# metric batches the iterator cycles over: 16 > the 14 draws of a layer's search (none recurs within a layer, as with a
# streamed DataLoader); METRIC_POOL=4: batches recur and the engine's reuse across candidates engages
METRIC_POOL = int(os.environ.get("METRIC_POOL", "16"))

dev = torch.device("cuda", 0)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32


def opt(name, default):
    return float(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


# thresholds of the rank search (library defaults 0.5 / 0.1: on a random-weight stack nothing passes them; a layer's
# share of the parameters shrinks with the depth, so --trade-off scales with the number of blocks when asked to)
trade_off = opt("--trade-off", 0.5)
max_ppl = opt("--max-ppl", 0.1)
D, KV, FF = 4096, 1024, 14336


class Block(torch.nn.Module):
    def __init__(self):
        super().__init__()
        mk = lambda i, o: torch.nn.Linear(i, o, bias=False)
        self.q, self.k, self.v, self.o = mk(D, D), mk(D, KV), mk(D, KV), mk(D, D)
        self.gate, self.up, self.down = mk(D, FF), mk(D, FF), mk(FF, D)

    @staticmethod
    def norm(x):
        return x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6).to(x.dtype)

    def forward(self, x):
        h = self.norm(x)
        x = x + self.o(self.q(h) + self.k(h).repeat(1, 1, D // KV) + self.v(h).repeat(1, 1, D // KV))
        h = self.norm(x)
        return x + self.down(torch.nn.functional.silu(self.gate(h)) * self.up(h))


class Stack(torch.nn.Module):
    def __init__(self, n):
        super().__init__()
        self.blocks = torch.nn.ModuleList(Block() for _ in range(n))
        self.head = torch.nn.Linear(D, D, bias=False)

    def forward(self, b):
        x = b["x"]
        for blk in self.blocks:
            x = blk(x)
        return self.head(x)


def ce(b, y):
    return torch.nn.functional.cross_entropy(y.float().reshape(-1, y.shape[-1]), b["targets"].reshape(-1), reduction="none")


# (weights and inputs are generated on the device: 7e9 parameters take minutes through the CPU generator)
g = torch.Generator(device=dev).manual_seed(0)
with torch.device(dev):
    model = Stack(blocks).to(dtype)
with torch.no_grad():
    for p in model.parameters():
        p.copy_((torch.randn(p.shape, generator=g, device=dev) / p.shape[1] ** 0.5).to(dtype))
xs = [torch.randn(1, 2048, D, generator=g, device=dev).to(dtype) for _ in range(8 + METRIC_POOL)]
with torch.no_grad():
    bt = [{"x": x, "targets": model({"x": x}).argmax(-1)} for x in xs]
torch.cuda.synchronize()
# sample check (tests/factor_checks.py): a few layers of the first precompute split are armed before the run
import factor_checks as sample_check
all_names = [n for n, m in model.named_modules() if isinstance(m, torch.nn.Linear) and n != "head"]
armed = sample_check.arm(model, all_names[:max(1, len(all_names) // 4)], bt[:8])
trace = []
from ptdeco_amd import _engine as eng
if os.environ.get("PTD_PHASES"):
    eng.PHASES = eng.PhaseTimer()   # device-time split (two event records per span)
import logging, threading
def heartbeat():
    # (a line a minute on stderr: a 32-block run takes minutes and gpurun takes silence for a hang)
    while not done.wait(45.0):
        print(f"[c4_stack] {time.perf_counter() - t0:.0f} s, {len(trace)} candidates evaluated", file=sys.stderr, flush=True)
done = threading.Event()
t0 = time.perf_counter()
threading.Thread(target=heartbeat, daemon=True).start()
cfg = ptdeco_amd.dwain.decompose_in_place(
    module=model, device=dev, data_iterator=itertools.cycle(bt[:12]), loss_fn=ce, metric_iterator=itertools.cycle(bt[8:]),
    num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0, finetune_fn=lambda m, d, n: m,
    trade_off_factor=trade_off, max_accepted_ppl_diff=max_ppl,
    blacklisted_module_names=["head"], precomputing_covariance_num_splits=4, trace=trace)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
done.set()
phases = None
if eng.PHASES is not None:
    phases = {k: round(v, 1) for k, v in eng.PHASES.totals_ms().items()}
    phases["other_host_and_gaps"] = round(dt * 1e3 - sum(phases.values()), 1)
layers = 7 * blocks
checked = sample_check.verify(armed, model, cfg)
print(json.dumps({"sample_check": checked, "metric_pool": METRIC_POOL,
                  "workload": f"dwain.decompose_in_place, Llama-3-8B-shaped stack, {blocks} blocks x (q, k, v, o, gate, up, down) at "
                              "4096 / 1024 / 14336 + blacklisted head, [1, 2048, 4096] calibration batches, D = 8, M = 2, "
                              "precomputing_covariance_num_splits = 4, f64 covariance + eigh, one MI355X",
                  "phases_ms": phases, "blocks": blocks, "dtype": str(dtype), "layers": layers, "seconds": dt, "layers_per_s": layers / dt,
                  "trade_off_factor": trade_off, "max_accepted_ppl_diff": max_ppl,
                  "candidates_evaluated": len(trace), "layers_replaced": len(cfg),
                  "decomposed": {k: v["__meta__"]["proportion"] for k, v in cfg.items()},
                  "max_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
