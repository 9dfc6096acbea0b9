"""Per-layer eigensolver profile of the bench's N-layer chain (precompute pass, one split), sequential streams."""
import copy, itertools, json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench, ptdeco_amd
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ["PTD_EIGH_STREAMS"] = sys.argv[2] if len(sys.argv) > 2 else "1"
model, data, metric = bench.make_workload(n_layers, dev, bench.D_STEPS, 7 * bench.M_STEPS)
model.to(dev)
data, metric = bench.with_targets(model, data, dev), bench.with_targets(model, metric, dev)
def step():
    m = copy.deepcopy(model)
    return ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(data), loss_fn=bench.ce_loss,
                                               metric_iterator=itertools.cycle(metric), finetune_fn=lambda mm, d, n: mm,
                                               precomputing_covariance_num_splits=1, **bench.DWAIN_KW)
step(); torch.cuda.synchronize()
ops.EIGH_PROFILE = []
step(); torch.cuda.synchronize()
for p in ops.EIGH_PROFILE:
    print(json.dumps({k: p[k] for k in ("n", "k", "method", "launches", "ms", "total_ms")}))
