"""Per-layer eigensolver log of bench.py's full-width Llama block (c4_block), bf16 or f32: which route each layer takes and
what it costs (PTD_JACOBI_DEBUG=1 prints the filtered route's estimates).  Usage: python tools/block_profile.py [f32] [streams]"""
import copy, itertools, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench, ptdeco_amd
from ptdeco_amd import ops, _engine as eng
dev = torch.device("cuda", 0)
dt = torch.float32 if "f32" in sys.argv else torch.bfloat16
for a in sys.argv[1:]:
    if a.isdigit():
        os.environ["PTD_EIGH_STREAMS"] = a
g = torch.Generator(device=dev).manual_seed(0)
with torch.device(dev):
    model0 = bench.LlamaStack(1)
with torch.no_grad():
    for prm in model0.parameters():
        prm.copy_(torch.randn(prm.shape, generator=g, device=dev) / prm.shape[1] ** 0.5)
model0.to(dt)
scale = torch.logspace(0, -2, bench.D_MODEL, device=dev)
xs = [(torch.randn(1, 2048, bench.D_MODEL, generator=g, device=dev) * scale).to(dt) for _ in range(12)]
with torch.no_grad():
    bt = [{"x": x, "targets": model0({"x": x}).argmax(-1)} for x in xs]


def step():
    m = copy.deepcopy(model0)
    return ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(bt), loss_fn=bench.seq_ce,
                                               metric_iterator=itertools.cycle(bt[8:]), finetune_fn=lambda mm, d, n: mm,
                                               **bench.C4_BLOCK_KW)


step(); torch.cuda.synchronize()
ops.EIGH_PROFILE = []
eng.PHASES = eng.PhaseTimer()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(json.dumps({"wall_ms": wall * 1e3, "phases": {k: round(v, 1) for k, v in eng.PHASES.totals_ms().items()}}))
for p in ops.EIGH_PROFILE:
    print(json.dumps({k: p[k] for k in ("n", "k", "method", "launches", "ms", "total_ms")}))
