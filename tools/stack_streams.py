"""B_eigh of bench.py's fixed 8-layer stack (bf16 model) by PTD_EIGH_STREAMS.  Usage: python tools/stack_streams.py 3 4 5 ..."""
import copy, itertools, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench, ptdeco_amd
from ptdeco_amd import _engine as eng
dev = torch.device("cuda", 0)
model, data, metric = bench.make_workload(bench.STACK_LAYERS, dev, bench.STACK_D_STEPS, 7 * bench.M_STEPS)
model.to(dev)
data, metric = bench.with_targets(model, data, dev), bench.with_targets(model, metric, dev)
model = model.bfloat16()
data = [{"x": b["x"].bfloat16(), "targets": b["targets"]} for b in data]
metric = [{"x": b["x"].bfloat16(), "targets": b["targets"]} for b in metric]
kw = dict(bench.DWAIN_KW, num_data_steps=bench.STACK_D_STEPS)
loss = lambda b, y: bench.ce_loss(b, y.float())


def step():
    m = copy.deepcopy(model)
    return ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(data), loss_fn=loss,
                                               metric_iterator=itertools.cycle(metric), finetune_fn=lambda mm, d, n: mm,
                                               precomputing_covariance_num_splits=1, **kw)


step(); torch.cuda.synchronize()
for s in sys.argv[1:]:
    os.environ["PTD_EIGH_STREAMS"] = s
    eng.PHASES = eng.PhaseTimer()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ph, eng.PHASES = eng.PHASES.totals_ms(), None
    print(json.dumps({"streams": int(s), "wall_ms": round(wall * 1e3, 1), "B_eigh_ms": round(ph["B_eigh"], 1), "D_ms": round(ph["D_metrics"], 1)}), flush=True)
