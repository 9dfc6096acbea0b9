"""Developer probe: every ptdeco_amd.ops call of one bench step with operand shapes and device time."""
import copy, itertools, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ptdeco_amd
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
model0, data, metric = bench.make_workload(1, dev, bench.D_STEPS, 7 * bench.M_STEPS)
model0.to(dev)
data, metric = bench.with_targets(model0, data, dev), bench.with_targets(model0, metric, dev)
def step():
    model = copy.deepcopy(model0)
    return ptdeco_amd.dwain.decompose_in_place(module=model, device=dev, data_iterator=itertools.cycle(data),
        loss_fn=bench.ce_loss, metric_iterator=itertools.cycle(metric), finetune_fn=lambda m, d, n: m, **bench.DWAIN_KW)
step(); torch.cuda.synchronize()
log = []
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = f(*a, **k); e1.record()
        desc = [f"{tuple(t.shape)}:{str(t.dtype)[6:]}:{t.stride()}" if isinstance(t, torch.Tensor) else repr(t) for t in a]
        log.append((name, desc, e0, e1))
        return out
    setattr(ops, name, g)
for nm in ("matmul", "lowrank_forward", "syrk_accumulate", "colsum_accumulate", "cov_finalize", "eigh", "eigh_factored", "nsr", "sym_kl"):
    if hasattr(ops, nm): wrap(nm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); step(); e1.record(); torch.cuda.synchronize()
tot = 0.0
for name, desc, a, b in log:
    ms = a.elapsed_time(b); tot += ms
    print(f"{ms:8.3f} ms  {name:16s} {'  '.join(desc)[:200]}")
print(f"ops total {tot:.2f} ms of step {e0.elapsed_time(e1):.2f} ms")
